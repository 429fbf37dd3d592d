"""Randomised parity of the HIP long-read library against the compiled reference (oracle/_ref, must have travelled):
np2_fuzz_gpu.py <first seed> <last seed> [sv]"""
import os, shutil, subprocess, sys, tempfile, json
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "..")); sys.path.insert(0, os.path.join(here, ".."))
import np2_cases, np2_gen, ref2_binding as rb
from nextpolish_amd import _native as nat
PRODUCT = os.path.join(here, "..", "..", "nextpolish_amd", "lib", "nextpolish2.so")
sv = len(sys.argv) > 3 and sys.argv[3] == "sv"
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    d = tempfile.mkdtemp(prefix="np2fz_")
    rt = [1, 2, 3, 1][seed % 4]
    split = 0
    if sv:
        hifi = rt == 3
        kw = dict(seed=seed, depth=[30, 40, 55][seed % 3], mean_len=[8000, 12000][seed % 2], hole=(60000 + 500 * (seed % 7), 60300 + 500 * (seed % 7)) if seed % 3 == 0 else None)
        if hifi:
            kw.update(sub=0.005, ins=0.003, dele=0.003)
        split = [1, 2, 0][seed % 3]
        fa, fofn, contigs = np2_cases.materialise_sv(kw, None, d)
    else:
        kw = dict(seed=seed, contig_lens=[(20000, 6000), (9000,), (30000, 1500, 700), (12000, 12000)][seed % 4], depth=[20, 8, 35, 70][(seed // 4) % 4],
                  max_indel=[2, 1, 6, 12][(seed // 5) % 4], mean_len=[4000, 1500, 9000][seed % 3], n_rate=0.001 if seed % 7 == 0 else 0.0,
                  iupac_rate=0.001 if seed % 5 == 0 else 0.0)
        if rt == 3:
            kw.update(sub=[0.002, 0.006][seed % 2], ins=[0.002, 0.01][(seed // 2) % 2], dele=[0.002, 0.008][(seed // 3) % 2], clip_rate=0.02)
        else:
            kw.update(sub=[0.03, 0.08, 0.005][seed % 3], ins=[0.02, 0.04, 0.002][(seed // 3) % 3], dele=[0.02, 0.05, 0.002][(seed // 2) % 3])
        fa, fofn, contigs = np2_cases.materialise(kw, d)
    # the reference runs in a child too: some inputs crash it (e.g. read bases with the IUPAC code M can leave a node
    # without links, and its backtrace then walks off the graph)
    rcode = ("import sys, json; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(rb.REF_SO); "
             "print(json.dumps(rb.polish(L, %r, %r, read_type=%d, split=%d)))" % (os.path.join(here, ".."), fa, fofn, rt, split))
    pr = subprocess.run([sys.executable, "-c", rcode], capture_output=True, text=True)
    code = ("import sys, json; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(%r); "
            "print(json.dumps(rb.polish(L, %r, %r, read_type=%d, split=%d)))" % (os.path.join(here, ".."), PRODUCT, fa, fofn, rt, split))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if pr.returncode != 0:
        print(seed, "rt", rt, "REFERENCE CRASHED (rc %d); this library: rc %d %s" % (pr.returncode, p.returncode, p.stderr.strip()[-100:]), flush=True)
    elif p.returncode != 0:
        print(seed, "rt", rt, "FAILED:", p.stderr.strip()[-160:]); bad += 1
    else:
        got = json.loads(p.stdout.strip().splitlines()[-1])
        ok = json.loads(pr.stdout.strip().splitlines()[-1]) == got
        if not ok:
            bad += 1
        print(seed, "rt", rt, "split", split, "OK" if ok else "DIFF", flush=True)
    shutil.rmtree(d)
print("mismatches:", bad)
