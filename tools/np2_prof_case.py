"""One warm-up + N timed ctg_cns_core calls on a prepared case directory (for rocprofv3): np2_prof_case.py <dir> [read_type] [n]"""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "tests"))
import ref2_binding as rb
case = sys.argv[1]
rt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
G = rb.bind(os.path.join(here, "..", "nextpolish_amd", "lib", "nextpolish2.so"))
fa, fofn = os.path.join(case, "g.fa"), os.path.join(case, "bam.fofn")
rb.polish(G, fa, fofn, read_type=rt)
t = time.time()
for _ in range(n):
    out = rb.polish(G, fa, fofn, read_type=rt)
dt = (time.time() - t) / n
bp = sum(len(p[0]) for v in out.values() for p in v)
print("%.3f s per call, %d bp -> %.2f Mbp/s" % (dt, bp, bp / dt / 1e6))
