"""Times the long-read path on one synthetic contig: reference CPU library (oracle/_ref) vs the HIP library."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import np2_cases, ref2_binding as rb
L = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = tempfile.mkdtemp(prefix="np2t_")
t = time.time()
fa, fofn, contigs = np2_cases.materialise(dict(seed=3, contig_lens=(L,), depth=depth, mean_len=8000, max_indel=4), d)
print("generated in %.1f s" % (time.time() - t))
here = os.path.dirname(os.path.abspath(__file__))
G = rb.bind(os.path.join(here, "..", "nextpolish_amd", "lib", "nextpolish2.so"))
t = time.time(); got = rb.polish(G, fa, fofn); tg = time.time() - t
t = time.time(); got2 = rb.polish(G, fa, fofn); tg2 = time.time() - t
print("gpu  %.2f s (second run %.2f s) -> %.3f Mbp/s" % (tg, tg2, L / tg2 / 1e6))
if rb.available():
    R = rb.bind(rb.REF_SO)
    t = time.time(); want = rb.polish(R, fa, fofn); tr = time.time() - t
    print("ref  %.2f s -> %.3f Mbp/s; identical: %s" % (tr, L / tr / 1e6, want == got))
