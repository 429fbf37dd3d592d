"""csrc/np_diag.cpp: with NP_ABORT_TRACE set, a process that loads either library and then dies of SIGABRT leaves the native frames of the
aborting thread on stderr (and in the file the variable names) before Python's faulthandler and the default action take over; without the
variable nothing is installed.  CPU only: loading the libraries needs no device."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = ("import ctypes, os, sys\n"
         "for so in sys.argv[1:]:\n"
         "    ctypes.CDLL(so)\n"
         "os.abort()\n")


def run_child(libs, env):
    e = dict(os.environ)
    e.pop("NP_ABORT_TRACE", None)
    e.pop("NP_ABORT_TRACE_ON", None)
    e.update(env)
    return subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD] + [os.path.join(ROOT, "nextpolish_amd", "lib", so) for so in libs],
                          capture_output=True, text=True, env=e)


@pytest.mark.parametrize("libs", [["nextpolish1.so"], ["nextpolish2.so"], ["nextpolish1.so", "nextpolish2.so"]])
def test_abort_trace_prints_native_frames_once_and_hands_over(libs, tmp_path):
    log = str(tmp_path / "abort.txt")
    p = run_child(libs, {"NP_ABORT_TRACE": log})
    assert p.returncode == -6, p.returncode                          # still dies of SIGABRT
    assert p.stderr.count("[np abort] SIGABRT") == 1, p.stderr       # one handler for the pair of libraries
    assert "libc.so.6" in p.stderr and "abort" in p.stderr           # the frames: ... raise / abort ...
    assert "Fatal Python error: Aborted" in p.stderr                 # faulthandler still has its say afterwards
    import glob
    files = glob.glob(log + "*")      # every process writes a file of its own: the name the variable gives + its pid (np_diag.cpp)
    assert len(files) == 1 and files[0] != log, files
    text = open(files[0]).read()
    assert "[np abort] SIGABRT" in text and "libc.so.6" in text


def test_a_child_process_installs_its_own_handlers(tmp_path):
    """the marker one process leaves in the environment (so that the second library of the pair does not install a second
    handler) must not switch the handlers off in the children that inherit that environment -- they are the ~60 CLI and harness processes
    of the suite."""
    log = str(tmp_path / "abort.txt")
    p = run_child(["nextpolish1.so"], {"NP_ABORT_TRACE": log, "NP_ABORT_TRACE_ON": str(os.getpid())})      # as inherited from a parent
    assert p.returncode == -6
    assert p.stderr.count("[np abort] SIGABRT") == 1, p.stderr


def test_without_the_variable_nothing_is_installed(tmp_path):
    p = run_child(["nextpolish1.so", "nextpolish2.so"], {})
    assert p.returncode == -6
    assert "[np abort]" not in p.stderr and "Fatal Python error: Aborted" in p.stderr
