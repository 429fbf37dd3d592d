"""Aggregate long-read throughput of P worker processes sharing one GPU (the reference's -p model): every process
polishes a synthetic contig N times through lib/nextpolish2.so after a warm-up call; the rate is the work of all
processes over the span from the first timed start to the last timed end.
usage: np2_multi.py [contig_len] [depth] [P,P,...] [calls] [host threads per process]"""
import os, sys, time, tempfile, subprocess
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, ".."))
from nextpolish_amd import _native as nat
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 20
procs = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1", "4", "8"])]
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 3
threads = sys.argv[5] if len(sys.argv) > 5 else None
d = tempfile.mkdtemp(prefix="np2m_")
t = time.time()
st = nat.Stream.synth_long([L], depth=float(depth), seed=5)
fa, bam, fofn = os.path.join(d, "g.fa"), os.path.join(d, "r.bam"), os.path.join(d, "bam.fofn")
st.write_files(fa, bam)
st.close()
open(fofn, "w").write(bam + "\n")
print("generated %d bp at %dx in %.1f s" % (L, depth, time.time() - t), flush=True)
code = ("import sys, time; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(%r); rb.polish(L, %r, %r); "
        "import resource; t0 = time.time(); c0 = resource.getrusage(resource.RUSAGE_SELF)\nfor _ in range(%d): rb.polish(L, %r, %r)\nc1 = resource.getrusage(resource.RUSAGE_SELF); print(t0, time.time(), c1.ru_utime + c1.ru_stime - c0.ru_utime - c0.ru_stime)") % (
    os.path.join(here, "..", "tests"), os.path.join(here, "..", "nextpolish_amd", "lib", "nextpolish2.so"), fa, fofn, calls, fa, fofn)
env = dict(os.environ)
if threads:
    env["NP_HOST_THREADS"] = env["NP_IO_THREADS"] = threads
for P in procs:
    ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for _ in range(P)]
    outs = [p.communicate() for p in ps]
    bad = [o[1][-400:] for p, o in zip(ps, outs) if p.returncode != 0]
    if bad:
        print("P=%d: %d processes failed: %s" % (P, len(bad), bad[0]), flush=True)
        continue
    spans = [[float(x) for x in o[0].strip().splitlines()[-1].split()] for o in outs]
    t0, t1 = min(s[0] for s in spans), max(s[1] for s in spans)
    per = [(s[1] - s[0]) / calls for s in spans]
    cpu = sum(s[2] for s in spans) / (P * calls)
    print("P=%d: %.3f..%.3f s per call and process, %.2f CPU-s per call -> aggregate %.2f Mbp/s" % (P, min(per), max(per), cpu, P * calls * L / (t1 - t0) / 1e6), flush=True)
