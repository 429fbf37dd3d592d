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
    os.environ.setdefault("NP_ABORT_TRACE_ALL", "1")
    _start_watchdog(d)


# ------------------------------------------------------------------------------------------------- a test that STOPS must say where
#
# Round 5 ended with the one-process GPU suite stopped inside one native call and nothing to read: pytest-timeout prints Python frames only.
# This watchdog (a daemon thread of the test process) ends a test that has been running for NP_TEST_WATCHDOG_S seconds (default 300; the
# longest test of the suite takes about two minutes; 0 switches it off) and leaves, in gpurun_out/hang_<pid>.txt and on stderr:
#   1. the state and kernel wait channel of every thread (/proc/self/task/*), which tells a thread blocked inside a driver ioctl from one
#      spinning or sleeping in user space;
#   2. rocgdb attached to the process for a moment: every host thread's native stack, and the GPU side -- agents, queues, dispatches in
#      flight and the waves still running, each with the kernel it belongs to;
#   3. Python's frames of every thread (faulthandler);
# then sends SIGABRT to the main thread, so csrc/np_diag.cpp prints the native frames of the blocked thread and of every other thread from
# inside the process (the fallback when no debugger can attach), and the process ends with a core-dump status instead of hanging.
_beat = {"t": 0.0, "name": None}


def _thread_states():
    out = []
    base = "/proc/self/task"
    for tid in sorted(os.listdir(base), key=int):
        row = [tid]
        for f in ("comm", "wchan"):
            try:
                row.append(open("%s/%s/%s" % (base, tid, f)).read().strip() or "-")
            except OSError:
                row.append("?")
        try:
            st = open("%s/%s/stat" % (base, tid)).read()
            row.append("state=" + st[st.rindex(")") + 2])
        except (OSError, ValueError):
            row.append("state=?")
        try:
            row.append("syscall=" + open("%s/%s/syscall" % (base, tid)).read().split()[0])
        except (OSError, IndexError):
            row.append("syscall=?")
        out.append(" ".join(row))
    return "\n".join(out)


def _debugger_dump(path):
    gdb = "/opt/rocm/bin/rocgdb"
    if not os.path.exists(gdb):
        return "no rocgdb on this box\n"
    cmds = ["set pagination off", "set confirm off", "set print thread-events off", "info agents", "info queues", "info dispatches",
            "info threads", "thread apply all -q bt 40"]
    argv = [gdb, "-q", "-batch", "-p", str(os.getpid())]
    for c in cmds:
        argv += ["-ex", c]
    try:
        with open(path + ".gdb", "wb") as f:
            subprocess.run(argv, stdout=f, stderr=subprocess.STDOUT, timeout=240, stdin=subprocess.DEVNULL)
    except Exception as e:      # noqa: BLE001 (whatever happens, the rest of the report must still be written)
        return "rocgdb did not finish: %r\n" % (e,)
    txt = open(path + ".gdb", "rb").read().decode("utf-8", "replace")
    return txt if len(txt) < (4 << 20) else txt[: 3 << 20] + "\n[... cut ...]\n" + txt[-(1 << 20):]


def _library_report(path):
    """np1_diag_report / np2_diag_report of whichever library this process has loaded (streams with hipStreamQuery's answer, allocator cache
    counters), asked from a helper thread of its own: the call goes into the HIP runtime, which the stopped thread may hold a lock of."""
    import ctypes
    import threading
    loaded = open("/proc/self/maps").read()

    def ask():
        with open(path + ".streams", "w") as f:
            for lib, fn in (("nextpolish1.so", "np1_diag_report"), ("nextpolish2.so", "np2_diag_report")):
                full = os.path.join(ROOT, "nextpolish_amd", "lib", lib)
                if full not in loaded and os.path.realpath(full) not in loaded:
                    continue
                f.write("[np watchdog] %s:\n" % fn)
                f.flush()
                try:
                    getattr(ctypes.CDLL(full), fn)(f.fileno())
                except Exception as e:      # noqa: BLE001
                    f.write("  failed: %r\n" % (e,))

    t = threading.Thread(target=ask, daemon=True)
    t.start()
    t.join(20.0)
    try:
        txt = open(path + ".streams").read()
    except OSError:
        txt = ""
    return txt + ("[np watchdog] (the report itself did not return within 20 s)\n" if t.is_alive() else "")


def _watchdog(limit, outdir):
    import faulthandler
    import signal
    import threading
    import time
    main_ident = threading.main_thread().ident
    while True:
        time.sleep(2.0)
        name, t0 = _beat["name"], _beat["t"]
        if not name or time.time() - t0 < limit:
            continue
        path = os.path.join(outdir, "hang_%d.txt" % os.getpid())
        with open(path, "w") as f:
            def say(s):
                f.write(s)
                f.flush()
                os.write(2, s.encode("utf-8", "replace"))      # (fd 2 itself: sys.stderr is pytest's capture object)
            say("\n[np watchdog] %s has been running for %.0f s (limit %d s): collecting the state of process %d\n" % (name, time.time() - t0, limit, os.getpid()))
            say("[np watchdog] threads (tid comm wchan state syscall):\n" + _thread_states() + "\n")
            say("[np watchdog] the libraries' streams and allocator caches:\n" + _library_report(path) + "\n")
            say("[np watchdog] debugger:\n" + _debugger_dump(path) + "\n")
            say("[np watchdog] threads after the debugger let go:\n" + _thread_states() + "\n")
            say("[np watchdog] Python frames:\n")
            faulthandler.dump_traceback(file=f, all_threads=True)
            faulthandler.dump_traceback(file=2, all_threads=True)
            say("[np watchdog] SIGABRT to the main thread (native frames follow from csrc/np_diag.cpp)\n")
        signal.pthread_kill(main_ident, signal.SIGABRT)
        time.sleep(90.0)      # (a thread that never leaves the kernel never takes the signal)
        os.write(2, b"[np watchdog] the main thread did not take SIGABRT within 90 s: _exit(70)\n")
        os._exit(70)


def _start_watchdog(outdir):
    limit = int(os.environ.get("NP_TEST_WATCHDOG_S", "300"))
    if limit <= 0 or _beat.get("started"):
        return
    _beat["started"] = True
    try:      # let a debugger that is our child attach (yama ptrace_scope 1 only admits ancestors otherwise)
        import ctypes
        ctypes.CDLL(None).prctl(0x59616D61, ctypes.c_ulong(-1 & 0xFFFFFFFFFFFFFFFF), 0, 0, 0)
    except Exception:      # noqa: BLE001
        pass
    import threading
    threading.Thread(target=_watchdog, args=(limit, outdir), name="np-watchdog", daemon=True).start()


def pytest_runtest_logstart(nodeid, location):
    import time
    _beat["t"], _beat["name"] = time.time(), nodeid


def pytest_runtest_logfinish(nodeid, location):
    _beat["name"] = None


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
