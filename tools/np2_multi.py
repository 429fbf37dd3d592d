"""Aggregate long-read throughput of P worker processes sharing one GPU (the reference's -p model): every process
polishes its own copy of a synthetic contig through lib/nextpolish2.so; reports total Mbp/s."""
import os, sys, time, tempfile, subprocess
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..")); sys.path.insert(0, os.path.join(here, "..", "tests"))
import np2_cases
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 20
procs = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1", "4", "8"])]
d = tempfile.mkdtemp(prefix="np2m_")
t = time.time()
fa, fofn, contigs = np2_cases.materialise(dict(seed=5, contig_lens=(L,), depth=depth, mean_len=8000, max_indel=4), d)
print("generated %d bp at %dx in %.1f s" % (L, depth, time.time() - t), flush=True)
code = ("import sys, time; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(%r); rb.polish(L, %r, %r); "
        "t = time.time(); rb.polish(L, %r, %r); print(time.time() - t)") % (
    os.path.join(here, "..", "tests"), os.path.join(here, "..", "nextpolish_amd", "lib", "nextpolish2.so"), fa, fofn, fa, fofn)
for P in procs:
    t = time.time()
    ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True) for _ in range(P)]
    secs = [float(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
    print("P=%d: per-process second-run %.2f..%.2f s -> aggregate %.2f Mbp/s" % (P, min(secs), max(secs), P * L / max(secs) / 1e6), flush=True)
