#!/usr/bin/env python
"""From-files score_chain in a warm process: how many device lanes / loader threads?  600 Mb draft in 4 batches of 150 Mb, 30x, Illumina-like
binned qualities.  usage: r4_e2e_lanes.py [mb_per_batch] [batches]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nextpolish_amd import _native as nat  # noqa: E402
from nextpolish_amd.device import Pipe  # noqa: E402

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if len(sys.argv) > 3 and sys.argv[3] == "child":
    fa, bam, lanes = sys.argv[4], sys.argv[5], int(sys.argv[6])
    bp = int(mb * 1e6) * nb
    pipe = Pipe(0, lanes=lanes)
    n = [0]
    best = 1e9
    for rep in range(3):
        n[0] = 0
        t0 = time.perf_counter()
        pipe.run_files(fa, bam, batch_bp=int(mb * 1e6) + 1000000, raw_sink=lambda name, p, ln: n.__setitem__(0, n[0] + ln))
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    print("lanes %d loaders %s: %.3f s -> %.1f Mbp/s (%d bases out)" % (lanes, os.environ.get("NP1_LOADERS", "3"), best, bp / 1e6 / best, n[0]), flush=True)
    pipe.close()
    sys.exit(0)
d = tempfile.mkdtemp(prefix="np1lanes_")
fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
with ThreadPoolExecutor(2) as ex:
    sts = list(ex.map(lambda k: nat.Stream.synth([int(mb * 1e6)], depth=30.0, seed=500 + k, prefix="L%dc" % k), range(nb)))
L = nat.lib()
L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
arr = (C.c_void_p * len(sts))(*[s.handle for s in sts])
assert L.np1_streams_write_files_q(arr, len(sts), fa.encode(), bam.encode(), 1, 1) == 0
print("BAM %.0f MB, %.1f B/record" % (os.path.getsize(bam) / 1e6, os.path.getsize(bam) / sum(s.n_reads for s in sts)), flush=True)
for s in sts:
    s.close()
for lanes, loaders in ((2, 3), (3, 3), (4, 4), (2, 6), (3, 6)):
    subprocess.run([sys.executable, os.path.abspath(__file__), str(mb), str(nb), "child", fa, bam, str(lanes)], env=dict(os.environ, NP1_LOADERS=str(loaders)))
