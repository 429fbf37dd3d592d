"""BASELINE configs 3, 4 and (piecewise) 5 at their stated sizes on one MI355X (the checks themselves live in tests/tools/ so they can be run
by hand; each prints one JSON line and exits non-zero on a mismatch)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_tool(name, *args):
    p = subprocess.run([sys.executable, os.path.join(HERE, "tools", name)] + list(args), capture_output=True, text=True)
    assert p.stdout.strip(), p.stderr[-2000:]
    info = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0, (info, p.stderr[-2000:])
    return info


def test_config3_size_short_reads_100mb_in_batches_matches_oracle():
    """~100 Mb in 94 contigs, 30x PE150 (20 M records), eight HBM batches, every contig bit-identical to the CPU oracle."""
    info = run_tool("check_config3.py")
    assert info["mismatches"] == 0 and info["draft_bp"] == 100000000 and info["batches"] >= 8 and info["contigs"] >= 50


def test_config4_size_long_reads_100mb_many_contigs_matches_reference_golden():
    """~100 Mb in 67 contigs, 20x ONT-like reads, four contigs of two or three windows, eight worker processes on the GPU:
    every contig identical (md5 + length) to what the compiled reference produced for the same files."""
    info = run_tool("check_config4.py")
    assert info["mismatches"] == 0 and info["missing"] == 0 and info["contigs"] == 67 and info["contigs_over_one_window"] >= 4


def test_config5_scale_chromosome_contig_and_multibatch_from_files_match_oracle():
    """The metric's 3 Gb shape at the sizes its pieces have: one 250 Mb contig at 30x (50 M records, one HBM batch) bit-identical to
    the CPU oracle, and a 113 Mb slice of four contigs FROM FILES (sorted BAM with Illumina-like binned qualities -> CLI -> device-side
    inflate + record split, one large contig per batch) identical contig by contig."""
    info = run_tool("check_config5.py")
    assert info["mismatches"] == 0 and info["big"]["draft_bp"] >= 249000000 and info["big"]["records"] >= 49000000
    assert info["slice"]["batches"] >= 4 and info["slice"]["draft_bp"] >= 112000000


def test_config5_scale_kmer_count_round_on_a_chromosome_contig_from_files_in_replay_mode():
    """Task 2 at the scale of the metric's draft: a 250 Mb contig at 30x with qualities from a sorted BAM through `nextpolish1 kmercount`
    (device ingest, iterator replay) identical to the oracle run on the same files with the iterator replayed."""
    info = run_tool("check_config5_kmer.py")
    assert info["mismatches"] == 0 and info["draft_bp"] >= 249000000 and info["records"] >= 49000000 and info["lower_case_out"] >= 0


def test_config5_scale_long_read_leg_chromosome_contig_of_52_windows_matches_reference_golden():
    """The long-read leg at chromosome scale: one 210 Mb contig = 52 overlapping windows stitched by link_consensus, 20x ONT-like reads,
    identical (md5 + length of every piece) to what the compiled reference produced for the same files."""
    info = run_tool("check_config5_lgs.py")
    assert info["mismatches"] == 0 and info["missing"] == 0 and info["windows"] >= 50 and info["draft_bp"] >= 200000000
