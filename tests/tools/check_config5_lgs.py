#!/usr/bin/env python
"""The long-read leg of BASELINE config 5 at chromosome scale: ONE contig of 210 Mb (52 windows of 5 Mb overlapping by 1 Mb, stitched
by link_consensus), 20x ONT-like reads (synthetic, nat.Stream.synth_long), polished by ctg_cns_core of a nextpolish2 library (default:
this repository's HIP library) and compared with the md5 + length the COMPILED REFERENCE produced for the same files
(tests/golden/config5_lgs_golden.json; --make-golden --library oracle/_ref/nextpolish2.so: about six minutes of one core).

usage: check_config5_lgs.py [--library path] [--make-golden] [--mb 210]
Prints one JSON line; exit code 1 on a mismatch."""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT)
sys.path.insert(0, TESTS)

GOLDEN = os.path.join(TESTS, "golden", "config5_lgs_golden.json")
CHILD = ("import sys, json, hashlib; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(sys.argv[1]); "
         "r = rb.polish(L, sys.argv[2], sys.argv[3], read_type=1); "
         "print(json.dumps({n: [[l, hashlib.md5(s.encode()).hexdigest()] for s, l in p] for n, p in r.items()}))" % TESTS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--library", default=os.path.join(ROOT, "nextpolish_amd", "lib", "nextpolish2.so"))
    ap.add_argument("--make-golden", action="store_true")
    ap.add_argument("--mb", type=float, default=210.0)
    a = ap.parse_args()
    from nextpolish_amd import _native as nat
    w = tempfile.mkdtemp(prefix="np2c5_")
    t0 = time.time()
    st = nat.Stream.synth_long([int(a.mb * 1e6)], depth=20.0, seed=9500, prefix="chr")
    fa, bam, fofn = os.path.join(w, "g.fa"), os.path.join(w, "r.bam"), os.path.join(w, "bam.fofn")
    st.write_files(fa, bam)
    n_reads = st.n_reads
    st.close()
    open(fofn, "w").write(bam + "\n")
    t_gen = time.time() - t0
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", CHILD, os.path.realpath(a.library), fa, fofn], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t_run = time.time() - t0
    bam_mb = os.path.getsize(bam) / 1e6
    shutil.rmtree(w)
    if p.returncode != 0:
        print(json.dumps({"error": "child exit %d: %s" % (p.returncode, p.stderr[-400:])}))
        return 2
    got = json.loads(p.stdout.strip().splitlines()[-1])
    bp = int(a.mb * 1e6)
    info = {"draft_bp": bp, "records": n_reads, "bam_mb": round(bam_mb, 1), "windows": (bp - 1000000 + 3999999) // 4000000, "gen_s": round(t_gen, 1),
            "polish_s": round(t_run, 1), "mbp_s": round(bp / 1e6 / t_run, 2)}
    if a.make_golden:
        json.dump({"mb": a.mb, "expected": got}, open(GOLDEN, "w"), indent=0, sort_keys=True)
        print(json.dumps(dict(info, wrote=GOLDEN)))
        return 0
    gold = json.load(open(GOLDEN))
    assert gold["mb"] == a.mb, "golden was generated for another size"
    bad = sorted(n for n in gold["expected"] if got.get(n) != gold["expected"][n])
    print(json.dumps(dict(info, mismatches=len(bad), missing=len(set(gold["expected"]) - set(got)), pieces={n: len(v) for n, v in got.items()})))
    return 1 if bad or len(got) != len(gold["expected"]) else 0


if __name__ == "__main__":
    sys.exit(main())
