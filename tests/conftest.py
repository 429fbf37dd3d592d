import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fatal signal inside native code (ours, the HIP runtime's) leaves the Python stacks of all threads in gpurun_out/fault.log as well
    # as on stderr, so that a crash on the GPU box can be read afterwards
    try:
        import faulthandler
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        config._np_fault_log = open(os.path.join(d, "fault.log"), "a")
        faulthandler.enable(file=config._np_fault_log, all_threads=True)
    except Exception:
        pass


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the in-tree libraries once per session (no-op when up to date)."""
    import __graft_entry__ as g
    g.build()


def ref_binary():
    """Path of the compiled reference CLI (oracle/_ref/nextpolish1) or None when it was never built."""
    p = os.path.join(ROOT, "oracle", "_ref", "nextpolish1")
    return p if os.path.exists(p) else None


def run_ref(cmd, fasta, bam, timeout=600):
    """Runs the reference CLI; returns {contig_name: sequence}."""
    out = subprocess.run([ref_binary(), cmd, fasta, bam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                         timeout=timeout, check=True).stdout.decode()
    return parse_cli_fasta(out)


def parse_cli_fasta(text):
    res, name = {}, None
    for line in text.splitlines():
        if line.startswith(">"):
            name = line[1:].rsplit("_", 1)[0]
            res[name] = ""
        elif name is not None:
            res[name] += line
    return res
