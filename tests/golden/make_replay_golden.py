#!/usr/bin/env python
"""Golden vectors for kmer_count and snp_valid where the reference's region iterator decides the result: md5 + length
of what the COMPILED REFERENCE (oracle/_ref/nextpolish1 kmercount | snpvalid) makes of thinly covered multi-window contigs
(tests/test_oracle.py thin_multiwindow_stream; seeds 44, 100, 102 are files on which "records in file order" differs from the iterator)
and of deep ones (deep_multiwindow_stream: the max_count_kmer break on a re-used iterator).  Runs in the build container only."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import run_ref  # noqa: E402
from test_oracle import deep_multiwindow_stream, thin_multiwindow_stream  # noqa: E402

THIN = [44, 100, 102, 7, 19]
DEEP = [3, 5]


def main():
    td = tempfile.mkdtemp()
    fa, bam = os.path.join(td, "z.fa"), os.path.join(td, "z.bam")
    gold = {"kmercount": {}, "snpvalid": {}}
    for kind, seeds, gen in (("thin", THIN, thin_multiwindow_stream), ("deep", DEEP, deep_multiwindow_stream)):
        for seed in seeds:
            st, level = gen(seed)
            st.write_files(fa, bam, level)
            for cmd in gold:
                try:
                    got = run_ref(cmd, fa, bam)
                except subprocess.CalledProcessError:
                    continue      # the reference crashed (snp_valid's null list): no golden for this file
                gold[cmd]["%s%d" % (kind, seed)] = {n: {"len": len(s), "md5": hashlib.md5(s.encode()).hexdigest()} for n, s in got.items()}
    json.dump(gold, open(os.path.join(HERE, "replay_golden.json"), "w"), indent=0, sort_keys=True)
    print("wrote replay_golden.json", {k: sorted(v) for k, v in gold.items()})


if __name__ == "__main__":
    main()
