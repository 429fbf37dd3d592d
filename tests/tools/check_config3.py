#!/usr/bin/env python
"""BASELINE config 3 at its stated size: ~100 Mb synthetic draft in ~100 contigs (lengths log-uniform in [50 kb, 5 Mb],
SURVEY.md 8d), 30x PE150, polished by score_chain in several HBM batches on one MI355X and compared contig by contig with
the CPU oracle (test infrastructure: oracle/libnp1_oracle.so, run in forked children that never touch HIP).

usage: check_config3.py [total_mb=100] [depth=30] [batch_mb=13] [from_files=0]
Prints one JSON line: {"contigs": n, "batches": b, "reads": r, "mismatches": m, "gpu_s": ..., ...}; exit code 1 on a mismatch."""
import hashlib
import json
import math
import multiprocessing as mp
import os
import random
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from nextpolish_amd import _native as nat  # noqa: E402
from nextpolish_amd.nextpolish1 import plan_batches  # noqa: E402
import oracle_binding as ob  # noqa: E402

STREAMS = []


def contig_lengths(total, seed=20250117 + 3):
    rng = random.Random(seed)
    lens, acc = [], 0
    while acc < total:
        L = int(math.exp(rng.uniform(math.log(50e3), math.log(5e6))))
        L = min(L, total - acc) if total - acc > 50000 else total - acc
        lens.append(L)
        acc += L
    return lens


def _oracle_digest(item):
    b, i = item
    s = ob.score_chain(STREAMS[b], i)
    return b, i, len(s), hashlib.md5(s.encode()).hexdigest()


def main():
    total = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 100000000
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    batch_bp = int(float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else 13000000
    lens = contig_lengths(total)
    names = ["c%03d" % i for i in range(len(lens))]
    batches = plan_batches(names, dict(zip(names, lens)), batch_bp)
    blens = [[lens[int(n[1:])] for n in b] for b in batches]
    ncpu = max(1, min(16, len(os.sched_getaffinity(0))))
    t0 = time.time()
    with ThreadPoolExecutor(ncpu) as ex:
        STREAMS.extend(ex.map(lambda k: nat.Stream.synth(blens[k], depth=depth, seed=7000 + k), range(len(blens))))
    t_synth = time.time() - t0
    items = sorted(((b, i) for b in range(len(blens)) for i in range(len(blens[b]))), key=lambda bi: -blens[bi[0]][bi[1]])
    t0 = time.time()
    with mp.get_context("fork").Pool(ncpu) as pool:       # children: CPU oracle only
        want = {(b, i): (n, d) for b, i, n, d in pool.imap_unordered(_oracle_digest, items, chunksize=1)}
    t_oracle = time.time() - t0
    from nextpolish_amd.device import Context
    ctx = Context(0)
    bad, t_gpu, reads = [], 0.0, 0
    for b, st in enumerate(STREAMS):
        t0 = time.time()
        got = ctx.score_chain(st)
        t_gpu += time.time() - t0
        reads += st.n_reads
        for i, s in enumerate(got):
            if (len(s), hashlib.md5(s.encode()).hexdigest()) != want[(b, i)]:
                bad.append([b, i, len(s), want[(b, i)][0]])
    ctx.close()
    print(json.dumps({"draft_bp": sum(lens), "contigs": len(lens), "longest": max(lens), "batches": len(batches), "depth": depth, "reads": reads,
                      "mismatches": len(bad), "first_bad": bad[:5], "synth_s": round(t_synth, 1), "oracle_s": round(t_oracle, 1),
                      "gpu_s_incl_upload_download": round(t_gpu, 2), "host_threads": ncpu}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
