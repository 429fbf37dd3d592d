"""Randomised parity of the split-read structural layer (np2_sv.cpp through the long-read host model, tests/model/libnp2_model.so)
against the compiled reference: contigs of 110-160 kb whose true sequence carries blocks the draft lacks (split reads with SA tags),
stretches no read crosses, assembler QV tracks; ONT / CLR / HiFi rules, split modes 0 / 1 / 2.  CPU only.
usage: np2_sv_fuzz.py [first_seed=0] [n=24] [procs=6]"""
import json
import os
import random
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

here = os.path.dirname(os.path.abspath(__file__))
T = os.path.join(here, "..")
sys.path.insert(0, T)
sys.path.insert(0, os.path.join(T, ".."))
import np2_cases  # noqa: E402
import ref2_binding as rb  # noqa: E402

MODEL = os.path.join(T, "model", "libnp2_model.so")
CHILD = ("import sys, json; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(sys.argv[1]); "
         "print(json.dumps(rb.polish(L, sys.argv[2], sys.argv[3], read_type=int(sys.argv[4]), split=int(sys.argv[5]))))" % T)


def run(so, fa, fofn, rt, split):
    p = subprocess.run([sys.executable, "-c", CHILD, so, fa, fofn, str(rt), str(split)], capture_output=True, text=True)
    return json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 and p.stdout.strip() else ("rc%d" % p.returncode)


def one(seed):
    rng = random.Random(seed * 7919 + 13)
    L = rng.choice([110000, 130000, 160000])
    n_sv = rng.choice([1, 2, 3])
    sv = tuple(sorted((rng.randrange(20000, L - 20000), rng.choice([300, 900, 1500, 4000])) for _ in range(n_sv)))
    rt = rng.choice([1, 1, 2, 3])
    kw = dict(seed=seed, L=L, depth=rng.choice([25, 40, 60]), sv=sv, mean_len=rng.choice([6000, 8000, 12000]))
    if rt == 3:
        kw.update(sub=0.005, ins=0.003, dele=0.003)
    if rng.random() < 0.5:
        h = rng.randrange(30000, L - 30000)
        kw["hole"] = (h, h + rng.choice([100, 300, 800]))
    qvs = None
    if rng.random() < 0.5:
        qvs = sorted((rng.randrange(5000, L - 5000), rng.randrange(100, 1000), rng.randrange(100, 1000), rng.randrange(100, 1000)) for _ in range(rng.choice([2, 5, 9])))
        if "hole" in kw and rng.random() < 0.7:
            qvs = sorted(qvs + [(kw["hole"][0] + rng.randrange(-300, 300), rng.randrange(50, 400), rng.randrange(50, 400), rng.randrange(50, 400))])
    split = rng.choice([0, 1, 1, 2])
    d = tempfile.mkdtemp(prefix="np2svfz%d_" % seed)
    fa, fofn, _contigs = np2_cases.materialise_sv(kw, qvs, d)
    a, b = run(MODEL, fa, fofn, rt, split), run(os.path.realpath(rb.REF_SO), fa, fofn, rt, split)
    return seed, rt, split, a == b, (a if isinstance(a, str) else "ok", b if isinstance(b, str) else "ok")


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
with ThreadPoolExecutor(int(sys.argv[3]) if len(sys.argv) > 3 else 6) as ex:
    res = list(ex.map(one, range(first, first + n)))
bad = [r for r in res if not r[3]]
crashed = [r for r in res if isinstance(r[4][1], str) and r[4][1].startswith("rc")]
print("%d cases, %d differ (%d where the reference itself crashed): %s" % (len(res), len(bad), len(crashed), bad[:10]))
