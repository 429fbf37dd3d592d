"""Writes a synthetic long-read case directory (g.fa, r.bam, bam.fofn) for tools/np2_stage_time.py.
usage: np2_make_case.py <dir> [contig Mb] [depth]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nextpolish_amd import _native as nat
d = sys.argv[1]
mb = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
depth = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
os.makedirs(d, exist_ok=True)
st = nat.Stream.synth_long([int(mb * 1e6)], depth=depth, seed=9000)
st.write_files(os.path.join(d, "g.fa"), os.path.join(d, "r.bam"))
st.close()
open(os.path.join(d, "bam.fofn"), "w").write(os.path.join(d, "r.bam") + "\n")
