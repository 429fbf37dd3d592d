#!/usr/bin/env python
"""Generates tests/golden/np1_golden.json from the REAL reference (oracle/_ref/nextpolish1, built by
oracle/Makefile from /root/reference).  Run in the build container only:

    python tests/golden/make_golden.py

Fixture = inputs (generator parameters, or explicit micro-case records) + the reference's outputs
(md5 / length per contig, full sequences for the small ones).  No reference source is stored."""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from nextpolish_amd import _native as nat  # noqa: E402
from conftest import run_ref  # noqa: E402
from fuzzgen import random_case  # noqa: E402

SYNTH = [
    dict(contig_len=[30000, 8000], depth=30.0, seed=3),
    dict(contig_len=[12000, 500, 200], depth=60.0, seed=11, weird_rate=0.03, softclip_rate=0.05, draft_lower=0.01),
    dict(contig_len=[20000], depth=120.0, seed=12, read_indel=0.002, draft_indel=0.02),
    dict(contig_len=[9000, 9000], depth=8.0, seed=13, draft_lower=0.03),
    dict(contig_len=[150000], depth=50.0, seed=14),
]
MICRO_SEEDS = list(range(0, 60))


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


def main():
    out = {"synth": [], "micro": []}
    with tempfile.TemporaryDirectory() as td:
        fa, bam = os.path.join(td, "g.fa"), os.path.join(td, "g.bam")
        for p in SYNTH:
            kw = dict(p)
            lens = kw.pop("contig_len")
            st = nat.Stream.synth(lens, with_qual=1, **kw)
            st.write_files(fa, bam)
            sc = run_ref("scorechain", fa, bam)
            kc = run_ref("kmercount", fa, bam)
            cfg = nat.lib().config_init(fa.encode(), bam.encode(), None)
            tlen, rlen = cfg.contents.read_tlen, cfg.contents.read_len
            nat.lib().config_destory(cfg)
            out["synth"].append({"params": p, "read_tlen": tlen, "read_len": rlen, "n_reads": st.n_reads,
                                 "score_chain": [{"name": n, "len": len(sc[n]), "md5": md5(sc[n])} for n in st.names],
                                 "kmer_count": [{"name": n, "len": len(kc[n]), "md5": md5(kc[n])} for n in st.names]})
        for seed in MICRO_SEEDS:
            contigs, reads = random_case(seed)
            st = nat.Stream.from_reads(contigs, reads)
            st.write_files(fa, bam)
            sc = run_ref("scorechain", fa, bam)
            out["micro"].append({"seed": seed, "contigs": contigs, "reads": reads,
                                 "score_chain": [sc[n] for n, _ in contigs]})
    with open(os.path.join(HERE, "np1_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "np1_golden.json"), os.path.getsize(os.path.join(HERE, "np1_golden.json")), "bytes")


if __name__ == "__main__":
    main()
