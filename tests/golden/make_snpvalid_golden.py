#!/usr/bin/env python
"""Golden vectors of task 4 (snp_valid, source/lib/snpvalid.c) from the COMPILED REFERENCE (oracle/_ref/nextpolish1 snpvalid):
  * the real-mapper fixtures under tests/golden/real/ (the reference's own task-1 output re-mapped = the input task 4 is run on
    in the reference's pipeline, and the plain draft),
  * seeded synthetic workloads of the product's generator (parameters below; the test rebuilds them from the seed).
Runs in the build container only (needs oracle/_ref).  Output: tests/golden/snpvalid_golden.json (md5 + length per contig)."""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import run_ref  # noqa: E402
from nextpolish_amd import _native as nat  # noqa: E402

REAL = os.path.join(HERE, "real")
SYNTH = [dict(lens=[4000 + 911 * k, 700 + 13 * k], depth=[6, 25, 90, 3, 12][k % 5], seed=7700 + k, with_qual=1, weird_rate=0.03, softclip_rate=0.06,
              draft_lower=[0.02, 0.08, 0.2][k % 3], read_indel=[0.001, 0.01][k % 2]) for k in range(8)]


def digest(s):
    return {"len": len(s), "md5": hashlib.md5(s.encode()).hexdigest()}


def main():
    gold = {"real": {}, "synth": []}
    for tag, fa, bam in (("r1.slice", "r1.fa", "r1.slice.bam"), ("sgs.s30", "g.fa", "sgs.s30.bam")):
        got = run_ref("snpvalid", os.path.join(REAL, fa), os.path.join(REAL, bam))
        gold["real"][tag] = {"fasta": fa, "bam": bam, "snp_valid": {n: digest(s) for n, s in got.items()}}
    td = tempfile.mkdtemp()
    fa, bam = os.path.join(td, "s.fa"), os.path.join(td, "s.bam")
    for kw in SYNTH:
        kw2 = dict(kw)
        lens = kw2.pop("lens")
        st = nat.Stream.synth(lens, **kw2)
        st.write_files(fa, bam)
        got = run_ref("snpvalid", fa, bam)
        cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
        gold["synth"].append({"params": kw, "read_tlen": cfgp.contents.read_tlen, "read_len": cfgp.contents.read_len,
                              "snp_valid": [digest(got[n]) for n in st.names]})
        nat.lib().config_destory(cfgp)
    json.dump(gold, open(os.path.join(HERE, "snpvalid_golden.json"), "w"), indent=0, sort_keys=True)
    print("wrote snpvalid_golden.json:", {k: len(v) for k, v in gold.items()})


if __name__ == "__main__":
    main()
