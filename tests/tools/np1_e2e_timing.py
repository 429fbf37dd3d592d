"""Stage timing of the from-files short-read path (NP1_TIMING lines of the CLI) on a generated workload.
usage: np1_e2e_timing.py [total Mb=100] [depth=30] [with_qual=0 | 1 random | 2 binned] [env sets: "all" | "timing"]"""
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
    sts = list(ex.map(lambda k: nat.Stream.synth([2500000] * int(MB / nb / 2.5), depth=DEPTH, seed=100 + k, with_qual=1 if WQ == 1 else 0, prefix="b%dc" % k), range(nb)))
fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
L = nat.lib()
L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
arr = (C.c_void_p * len(sts))(*[s.handle for s in sts])
assert L.np1_streams_write_files_q(arr, len(sts), fa.encode(), bam.encode(), 1, 1 if WQ == 2 else 0) == 0      # (with_qual 2 = Illumina-like binned qualities, as bench.py writes them)
print("generated %.0f Mb %.0fx in %.1f s; BAM %.0f MB" % (MB, DEPTH, time.time() - t, os.path.getsize(bam) / 1e6), flush=True)
exe = os.path.join(here, "..", "..", "nextpolish_amd", "bin", "nextpolish1")
MODE = sys.argv[4] if len(sys.argv) > 4 else "all"
if MODE == "sweep":      # lanes x loaders x batch size of the from-files CLI (round 6: the decoder takes ~14.5 ms per launch whatever the launch holds, up to ~49 k blocks)
    SETS = tuple(dict(NP1_LANES=str(l), NP1_LOADERS=str(ld), NP1_BATCH_BP=str(bp)) for bp in (16000000, 32000000, 48000000) for l, ld in ((3, 4), (4, 6), (6, 8), (8, 8)))
else:
  SETS = (dict(NP1_TIMING="1"), {}) if MODE == "timing" else (dict(NP1_TIMING="1"), dict(NP1_LANES="3", NP1_LOADERS="4"), dict(NP1_LANES="3", NP1_LOADERS="6"),
                                                                                          dict(NP1_LANES="4", NP1_LOADERS="6"), dict(NP1_INGEST="host"), {})
for env in SETS:
    best = 1e9
    for k in range(3):
        t = time.time()
        p = subprocess.run([exe, "scorechain", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), check=True)
        best = min(best, time.time() - t)
    print("#### env %s: %.3f s (best of 3) -> %.1f Mbp/s" % (env, best, MB / best), flush=True)
    if "NP1_TIMING" in env:
        print(p.stderr.decode()[-4000:])
