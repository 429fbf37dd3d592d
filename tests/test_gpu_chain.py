"""BASELINE config 5 as one run (tests/tools/check_config5_chain.py): lgs, lgs, score_chain, kmer_count, score_chain, kmer_count -- the reference's
`task = best` order -- each step on the FASTA the step before wrote, reads generated on that FASTA, nextpolish2.so and nextpolish1.so's file pipe
alternating in this process; every step's output (length + md5 per contig) against what the compiled reference wrote for the same files
(tests/golden/config5_chain_golden.json)."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "tools"))
import check_config5_chain as chain  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size", ["quick", "full"])
def test_two_round_sgs_lgs_chain_equals_the_reference_at_every_step(size, tmp_path):
    """quick: 1.4 Mb in four contigs, four short-read batches a step; full: 19 Mb, three short-read batches, the 11 Mb contig in three
    long-read windows stitched by link_consensus"""
    r = chain.run(size, workdir=str(tmp_path))
    assert len(r["steps"]) == len(chain.STEPS) and r["identical"], r
