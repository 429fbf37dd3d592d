"""Times the long-read path on one synthetic contig: reference CPU library (oracle/_ref) vs the HIP library.
usage: np2_time.py [contig_len] [depth] [sub] [ins] [del] [read_type]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import np2_cases, ref2_binding as rb
a = sys.argv[1:]
L = int(a[0]) if len(a) > 0 else 300000
depth = int(a[1]) if len(a) > 1 else 20
sub = float(a[2]) if len(a) > 2 else 0.03
ins = float(a[3]) if len(a) > 3 else 0.02
dele = float(a[4]) if len(a) > 4 else 0.02
rt = int(a[5]) if len(a) > 5 else 1
d = tempfile.mkdtemp(prefix="np2t_")
t = time.time()
fa, fofn, contigs = np2_cases.materialise(dict(seed=3, contig_lens=(L,), depth=depth, mean_len=8000, max_indel=4, sub=sub, ins=ins, dele=dele), d)
print("generated in %.1f s: %d bp, %dx, sub %.3f ins %.3f del %.3f, read type %d" % (time.time() - t, L, depth, sub, ins, dele, rt))
here = os.path.dirname(os.path.abspath(__file__))
G = rb.bind(os.path.join(here, "..", "nextpolish_amd", "lib", "nextpolish2.so"))
t = time.time(); got = rb.polish(G, fa, fofn, read_type=rt); tg = time.time() - t
t = time.time(); got2 = rb.polish(G, fa, fofn, read_type=rt); tg2 = time.time() - t
print("gpu  %.2f s (second run %.2f s) -> %.3f Mbp/s" % (tg, tg2, L / tg2 / 1e6))
if rb.available():
    R = rb.bind(rb.REF_SO)
    t = time.time(); want = rb.polish(R, fa, fofn, read_type=rt); tr = time.time() - t
    print("ref  %.2f s -> %.3f Mbp/s; identical: %s" % (tr, L / tr / 1e6, want == got))
