#!/usr/bin/env python
"""Randomised check of intra-contig tiling: the product's tiling driver with the host model in place of the device
(tests/model: np1m_score_chain_tiled_files -- load_stream_region through the BAI, the driver's join arithmetic) against the untiled
oracle, over random coverage, error rates, soft clips, odd CIGAR shapes, lower-case drafts, thin and gappy pileups, tile sizes from a
few bases to more than the contig and halos from one base up.  CPU only.  usage: np1_tiling_fuzz.py FIRST LAST"""
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import model_binding as mb  # noqa: E402
import oracle_binding as ob  # noqa: E402
from nextpolish_amd import _native as nat  # noqa: E402

a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp(prefix="np1tilefz_")
fa, bam = d + "/g.fa", d + "/r.bam"
st = dict(cases=0, tilings=0, wrong=0, recomputed=0, tiles=0)
for seed in range(a, b):
    rng = random.Random(seed)
    lens = [rng.choice([700, 3000, 9000, 20000]) for _ in range(rng.choice([1, 2, 3]))]
    s = nat.Stream.synth(lens, depth=rng.choice([2, 5, 12, 30, 60]), seed=seed, read_indel=rng.choice([0.0001, 0.003, 0.01]), read_sub=rng.choice([0.001, 0.02]),
                         softclip_rate=rng.choice([0.0, 0.05, 0.2]), draft_lower=rng.choice([0.0, 0.02, 0.3]), weird_rate=rng.choice([0.0, 0.02, 0.1]),
                         draft_indel=rng.choice([0.001, 0.01, 0.03]), draft_sub=rng.choice([0.001, 0.01]))
    s.write_files(fa, bam)
    want = [ob.score_chain(s, i) for i in range(s.n_contigs)]
    st["cases"] += 1
    for _ in range(4):
        tile = rng.choice([rng.randrange(5, 60), rng.randrange(60, 700), rng.randrange(700, 6000), rng.randrange(6000, 40000)])
        halo = rng.choice([1, rng.randrange(1, 20), rng.randrange(20, 300), rng.randrange(300, 2000)])
        for i, n in enumerate(s.names):
            got, info = mb.score_chain_tiled_files(fa, bam, n, tile, halo, fused=rng.randrange(3))
            st["tilings"] += 1
            st["tiles"] += info["tiles"]
            st["recomputed"] += info["recomputed"]
            if got != want[i]:
                st["wrong"] += 1
                print("WRONG seed %d contig %s tile %d halo %d" % (seed, n, tile, halo), flush=True)
    s.close()
print(st)
