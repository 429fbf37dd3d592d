"""Randomised parity of the long-read host model (tests/model/libnp2_model.so = the product's host pipeline + host executor)
against the compiled reference, CPU only.  usage: np2_fuzz_cpu.py [first_seed=0] [n=60] [procs=8]"""
import os, sys, tempfile, random, json, subprocess
from concurrent.futures import ThreadPoolExecutor
here = os.path.dirname(os.path.abspath(__file__))
T = os.path.join(here, "..")
sys.path.insert(0, T); sys.path.insert(0, os.path.join(T, ".."))
import np2_cases, ref2_binding as rb
MODEL = os.path.join(T, "model", "libnp2_model.so")
CHILD = ("import sys, json; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(sys.argv[1]); "
         "print(json.dumps(rb.polish(L, sys.argv[2], sys.argv[3], read_type=int(sys.argv[4]))))" % T)
def run(so, fa, fofn, rt):
    p = subprocess.run([sys.executable, "-c", CHILD, so, fa, fofn, str(rt)], capture_output=True, text=True)
    return json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 and p.stdout.strip() else ("rc%d" % p.returncode)
def one(seed):
    rng = random.Random(seed)
    rt = rng.choice([1, 1, 2, 3])
    if rt == 3:
        kw = dict(seed=seed, contig_lens=(rng.choice([12000, 20000]),), depth=rng.choice([15, 30]), sub=0.002, ins=rng.choice([0.002, 0.01]), dele=rng.choice([0.002, 0.008]),
                  max_indel=rng.choice([1, 4, 6]), mean_len=9000, clip_rate=0.02)
    else:
        kw = dict(seed=seed, contig_lens=rng.choice([(15000,), (8000, 3000), (25000,)]), depth=rng.choice([10, 25, 40, 60]), max_indel=rng.choice([2, 6, 10]),
                  sub=rng.choice([0.02, 0.06]), ins=rng.choice([0.02, 0.04]), dele=rng.choice([0.02, 0.05]))
    d = tempfile.mkdtemp(prefix="np2fz%d_" % seed)
    fa, fofn, contigs = np2_cases.materialise(kw, d)
    a, b = run(MODEL, fa, fofn, rt), run(os.path.realpath(rb.REF_SO), fa, fofn, rt)
    return seed, rt, a == b, (a if isinstance(a, str) else "ok", b if isinstance(b, str) else "ok")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
with ThreadPoolExecutor(int(sys.argv[3]) if len(sys.argv) > 3 else 8) as ex:
    res = list(ex.map(one, range(first, first + n)))
bad = [r for r in res if not r[2]]
print("%d cases, %d differ: %s" % (len(res), len(bad), bad[:10]))
