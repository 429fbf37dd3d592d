#!/usr/bin/env python
"""Writes a diploid snp_phase workload (tests/snpphase_gen.py) as FASTA + short-read BAM + long-read BAM into a directory:
   python tests/tools/np1_phase_case.py OUTDIR LENGTH [seed]   (generation is plain Python: ~1 min per Mb)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import snpphase_gen  # noqa: E402
from nextpolish_amd import _native as nat  # noqa: E402

out, L = sys.argv[1], int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
os.makedirs(out, exist_ok=True)
t = time.time()
ctgs, srs, lrs = snpphase_gen.make_case(seed, lens=(L,), sr_depth=30, lr_depth=20, lr_len=8000, het=0.002, het_indel=0.0003, draft_err=0.002, read_len=150, frag=400,
                                        sr_holes=max(1, L // 200000))
print("generated %d short, %d long reads in %.0f s" % (len(srs), len(lrs), time.time() - t))
nat.Stream.from_reads(ctgs, srs).write_files(os.path.join(out, "g.fa"), os.path.join(out, "sr.bam"))
nat.Stream.from_reads(ctgs, lrs).write_files(os.path.join(out, "l.fa"), os.path.join(out, "lr.bam"))
print("wrote", out)
