#!/usr/bin/env python
"""GPU check of the tiles of one dominant contig shared by two ranks (nextpolish1.py --world 2 --tile_bp, DESIGN.md sections 8 and 11): two
caller processes on the one GPU, their -o parts concatenated, against one untiled single-rank run.  Written at the end of round 4 with no
GPU time left: run it first (`python tests/tools/np1_tile_ranks_check.py`), then make it a test of tests/test_gpu_tiling.py."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nextpolish_amd import _native as nat  # noqa: E402

d = tempfile.mkdtemp(prefix="np1tileranks_")
st = nat.Stream.synth([200000, 6000000, 90000, 2500000, 30000], depth=30, seed=8128)
fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
st.write_files(fa, bam)
caller = os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py")


def records(path):
    out, name = {}, None
    for line in open(path):
        if line.startswith(">"):
            name = line[1:].split()[0]
            out[name] = ""
        else:
            out[name] += line.strip()
    return out


one = os.path.join(d, "one.fa")
subprocess.run([sys.executable, caller, "-g", fa, "-t", "1", "-s", bam, "-o", one], check=True)
ps = [subprocess.Popen([sys.executable, caller, "-g", fa, "-t", "1", "-s", bam, "-o", os.path.join(d, "part%d.fa" % r), "--world", "2", "--rank", str(r), "--device", "0",
                        "--tile_bp", "1000000", "--tile_dir", os.path.join(d, "tiles"), "--tile_wait", "600"]) for r in range(2)]
rc = [p.wait() for p in ps]
assert rc == [0, 0], rc
want = records(one)
got = {}
for r in range(2):
    part = records(os.path.join(d, "part%d.fa" % r))
    assert not set(part) & set(got), "a contig in both parts"
    got.update(part)
assert got == want, [n for n in want if got.get(n) != want[n]]
assert not os.listdir(os.path.join(d, "tiles")), "pieces left behind"
print("ok: %d contigs, 2 of them shared tile by tile by two ranks, identical to the untiled single-rank run" % len(want))
