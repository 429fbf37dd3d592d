"""Polishes a prepared case directory with the HIP library and compares with want.json (md5 + length of every piece,
written from the compiled reference): np2_check_case.py <dir> [read_type]"""
import hashlib, json, os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "tests"))
import ref2_binding as rb
case = sys.argv[1]
rt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
G = rb.bind(os.path.join(here, "..", "nextpolish_amd", "lib", "nextpolish2.so"))
t = time.time()
got = rb.polish(G, os.path.join(case, "g.fa"), os.path.join(case, "bam.fofn"), read_type=rt)
dt = time.time() - t
got = {k: [[hashlib.md5(p[0].encode()).hexdigest(), p[1]] for p in v] for k, v in got.items()}
want = json.load(open(os.path.join(case, "want.json")))
print("%s: %.2f s, %s" % (case, dt, "IDENTICAL to the reference" if got == want else "DIFFERENT"))
sys.exit(0 if got == want else 1)
