#!/usr/bin/env python
"""Golden vectors for kmer_count on thinly covered multi-window contigs (tests/test_oracle.py thin_multiwindow_stream): md5 + length of
what the COMPILED REFERENCE (oracle/_ref/nextpolish1 kmercount) makes of them.  Seeds 44, 100, 102 are the files on which "records
in file order" differs from the reference's region iterator.  Runs in the build container only."""
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
from test_oracle import thin_multiwindow_stream  # noqa: E402

SEEDS = [44, 100, 102, 7, 19]


def main():
    td = tempfile.mkdtemp()
    fa, bam = os.path.join(td, "z.fa"), os.path.join(td, "z.bam")
    gold = {}
    for seed in SEEDS:
        st, level = thin_multiwindow_stream(seed)
        st.write_files(fa, bam, level)
        got = run_ref("kmercount", fa, bam)
        gold[str(seed)] = {n: {"len": len(s), "md5": hashlib.md5(s.encode()).hexdigest()} for n, s in got.items()}
    json.dump(gold, open(os.path.join(HERE, "replay_golden.json"), "w"), indent=0, sort_keys=True)
    print("wrote replay_golden.json", list(gold))


if __name__ == "__main__":
    main()
