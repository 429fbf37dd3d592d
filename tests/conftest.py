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
    # a fatal signal inside native code (ours, the HIP runtime's) must leave enough behind to be read afterwards
    # (pytest's own faulthandler plugin prints them on stderr; the libraries add the native frames of the aborting thread and the exception
    # behind a std::terminate when NP_ABORT_TRACE is set: csrc/np_diag.cpp)
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    os.environ.setdefault("NP_ABORT_TRACE", os.path.join(d, "abort_%d.txt" % os.getpid()))


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
