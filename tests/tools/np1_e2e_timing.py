"""Stage timing of the from-files short-read path (NP1_TIMING lines of the CLI) on a generated workload.
usage: np1_e2e_timing.py [total Mb=100] [depth=30] [with_qual=0]"""
import os, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", ".."))
import ctypes as C
from nextpolish_amd import _native as nat
MB = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
DEPTH = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
WQ = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d = tempfile.mkdtemp(prefix="np1e2e_")
nb = max(1, int(MB / 12.5))
t = time.time()
with ThreadPoolExecutor(8) as ex:
    sts = list(ex.map(lambda k: nat.Stream.synth([2500000] * int(MB / nb / 2.5), depth=DEPTH, seed=100 + k, with_qual=WQ, prefix="b%dc" % k), range(nb)))
fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
L = nat.lib()
L.np1_streams_write_files.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int]
arr = (C.c_void_p * len(sts))(*[s.handle for s in sts])
assert L.np1_streams_write_files(arr, len(sts), fa.encode(), bam.encode(), 1) == 0
print("generated %.0f Mb %.0fx in %.1f s; BAM %.0f MB" % (MB, DEPTH, time.time() - t, os.path.getsize(bam) / 1e6), flush=True)
exe = os.path.join(here, "..", "..", "nextpolish_amd", "bin", "nextpolish1")
for env in (dict(NP1_TIMING="1"), dict(NP1_LANES="3", NP1_LOADERS="4"), dict(NP1_LANES="3", NP1_LOADERS="6"), dict(NP1_LANES="4", NP1_LOADERS="6"), dict(NP1_INGEST="host"), {}):
    best = 1e9
    for k in range(3):
        t = time.time()
        p = subprocess.run([exe, "scorechain", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), check=True)
        best = min(best, time.time() - t)
    print("#### env %s: %.3f s (best of 3) -> %.1f Mbp/s" % (env, best, MB / best), flush=True)
    if "NP1_TIMING" in env:
        print(p.stderr.decode()[-4000:])
