#!/bin/bash
# round 6: where the cold start of the from-files CLI goes (NP1_TIMING lines from the start of a run; AMD_LOG_LEVEL off)
mkdir -p gpurun_out/r6
python3 - <<'PY' > gpurun_out/r6/cold_start.txt 2>&1
import os, subprocess, sys, tempfile, time, ctypes as C
sys.path.insert(0, os.getcwd())
from nextpolish_amd import _native as nat
from concurrent.futures import ThreadPoolExecutor
d = tempfile.mkdtemp(prefix="np1cold_")
with ThreadPoolExecutor(8) as ex:
    sts = list(ex.map(lambda k: nat.Stream.synth([2500000] * 10, depth=30.0, seed=100 + k, with_qual=0, prefix="b%dc" % k), range(16)))
fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
L = nat.lib()
L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
arr = (C.c_void_p * len(sts))(*[s.handle for s in sts])
assert L.np1_streams_write_files_q(arr, len(sts), fa.encode(), bam.encode(), 1, 1) == 0
exe = os.path.join("nextpolish_amd", "bin", "nextpolish1")
for rep in range(2):
    t = time.time()
    p = subprocess.run([exe, "scorechain", fa, bam], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, NP1_TIMING="1", NP_ALLOC_TIMING="1"))
    print("==== run %d: %.3f s for 400 Mb" % (rep, time.time() - t))
    lines = p.stderr.decode().splitlines()
    print("\n".join(lines[:70]))
    print("...")
    print("\n".join(lines[-6:]))
PY
head -150 gpurun_out/r6/cold_start.txt | cut -c1-260
