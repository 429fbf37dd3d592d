#!/usr/bin/env python
"""BASELINE config 4 at its stated size: ~100 Mb synthetic draft in many contigs (one of 12.5 Mb = three windows, 9 / 6 /
5.2 Mb = two windows, the rest log-uniform in [50 kb, 5 Mb]), 20x ONT-like reads, polished by ctg_cns_core of a nextpolish2
library (default: this repository's HIP library, one worker process per group sharing the GPU) and compared with the md5s the
COMPILED REFERENCE produced for the same files (tests/golden/config4_golden.json).

usage: check_config4.py [--library path] [--make-golden] [--total-mb 100] [--procs 8]
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
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT)
sys.path.insert(0, TESTS)
import np2_cases  # noqa: E402

GOLDEN = os.path.join(TESTS, "golden", "config4_golden.json")
CHILD = ("import sys, json, hashlib; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(sys.argv[1]); "
         "r = rb.polish(L, sys.argv[2], sys.argv[3], read_type=1); "
         "print(json.dumps({n: [[l, hashlib.md5(s.encode()).hexdigest()] for s, l in p] for n, p in r.items()}))" % TESTS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--library", default=os.path.join(ROOT, "nextpolish_amd", "lib", "nextpolish2.so"))
    ap.add_argument("--make-golden", action="store_true")
    ap.add_argument("--total-mb", type=float, default=100.0)
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    groups = np2_cases.config4_groups(int(a.total_mb * 1e6))
    w = tempfile.mkdtemp(prefix="np2c4_")
    ncpu = max(1, min(16, len(os.sched_getaffinity(0))))
    t0 = time.time()
    with ThreadPoolExecutor(min(ncpu, len(groups))) as ex:
        files = list(ex.map(lambda k: np2_cases.materialise_config4_group(k, groups[k], w), range(len(groups))))
    t_gen = time.time() - t0
    t0 = time.time()
    procs, got = [], {}
    pending = list(range(len(groups)))
    running = []
    while pending or running:
        while pending and len(running) < a.procs:
            k = pending.pop(0)
            running.append((k, subprocess.Popen([sys.executable, "-c", CHILD, os.path.realpath(a.library), files[k][0], files[k][1]],
                                                stdout=subprocess.PIPE, text=True)))
        k, p = running.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            print(json.dumps({"error": "group %d: child exit %d" % (k, p.returncode)}))
            return 2
        got.update(json.loads(out.strip().splitlines()[-1]))
    t_run = time.time() - t0
    shutil.rmtree(w)
    total = sum(sum(g) for g in groups)
    info = {"draft_bp": total, "contigs": sum(len(g) for g in groups), "groups": len(groups), "longest": max(max(g) for g in groups),
            "gen_s": round(t_gen, 1), "polish_s": round(t_run, 1), "mbp_s": round(total / 1e6 / t_run, 2), "procs": a.procs}
    if a.make_golden:
        json.dump({"total_mb": a.total_mb, "expected": got}, open(GOLDEN, "w"), indent=0, sort_keys=True)
        print(json.dumps(dict(info, wrote=GOLDEN)))
        return 0
    gold = json.load(open(GOLDEN))
    assert gold["total_mb"] == a.total_mb, "golden was generated for another size"
    bad = sorted(n for n in gold["expected"] if got.get(n) != gold["expected"][n])
    multi = sum(1 for n, p in got.items() if p[0][0] > 5000000)
    print(json.dumps(dict(info, mismatches=len(bad), first_bad=bad[:5], missing=len(set(gold["expected"]) - set(got)),
                          contigs_over_one_window=multi)))
    return 1 if bad or len(got) != len(gold["expected"]) else 0


if __name__ == "__main__":
    sys.exit(main())
