"""End-to-end time of the short-read path from files (FASTA + sorted BAM -> polished FASTA): this library's CLI vs the
compiled reference binary, on the bench workload.  usage: np1_e2e.py [threads] [total Mb] [depth] [ref: 0|1]"""
import os, subprocess, sys, tempfile, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", ".."))
from nextpolish_amd import _native as nat
d = tempfile.mkdtemp(prefix="np1e2e_")
t = time.time()
MB = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
DEPTH = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
WITH_REF = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
lens = [2500000, 1500000, 1000000] if MB == 5.0 else [2500000] * int(MB / 2.5)
st = nat.Stream.synth(lens, depth=DEPTH, seed=20250119, with_qual=1)
fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
st.write_files(fa, bam)
st.close()
print("generated %.1f Mb, %.0fx PE150 in %.1f s; BAM %.0f MB" % (MB, DEPTH, time.time() - t, os.path.getsize(bam) / 1e6), flush=True)
exe = os.path.join(here, "..", "..", "nextpolish_amd", "bin", "nextpolish1")
ref = os.path.join(here, "..", "..", "oracle", "_ref", "nextpolish1")
for th in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8"]):
    env = dict(os.environ, NP_IO_THREADS=th)
    best = 1e9
    for k in range(4):
        t = time.time()
        out = subprocess.run([exe, "scorechain", fa, bam], stdout=subprocess.PIPE, env=env, check=True).stdout
        best = min(best, time.time() - t)
    dt = best
    print("this library, %s inflate threads: %.2f s (best of 4) -> %.1f Mbp/s, %d bytes out" % (th, dt, MB / dt, len(out)), flush=True)
if WITH_REF and os.path.exists(ref):
    t = time.time()
    rout = subprocess.run([ref, "scorechain", fa, bam], stdout=subprocess.PIPE, check=True).stdout
    dt = time.time() - t
    print("reference binary, 1 core: %.2f s -> %.2f Mbp/s; identical output: %s" % (dt, MB / dt, rout == out))
