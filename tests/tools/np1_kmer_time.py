"""kmer_count (task 2) on the bench-shaped workload with 0.4 % of the draft flagged lowercase (what a score_chain pass
leaves behind): GPU batch time vs the CPU oracle and, when it travelled, the compiled reference CLI from files."""
import os, subprocess, sys, tempfile, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "..")); sys.path.insert(0, os.path.join(here, ".."))
from nextpolish_amd import _native as nat
from nextpolish_amd.device import Context
import oracle_binding as ob
lens = [2500000, 1500000, 1000000]
st = nat.Stream.synth(lens, depth=50.0, seed=20250119, with_qual=1, draft_lower=0.004)
ctx = Context(0)
cfg = nat.default_config()
cfg.read_tlen = 1500
b = ctx.upload(st)
b.kmer_count(cfg)
t = time.time()
for _ in range(5):
    b.kmer_count(cfg)
dt = (time.time() - t) / 5
got = b.results()
print("GPU kmer_count, 5 Mb / 50x, inputs resident: %.2f ms per pass -> %.0f Mbp/s" % (dt * 1e3, 5.0 / dt))
t = time.time()
same = all(ob.kmer_count(st, i, ob.default_config(read_tlen=1500)) == got[i] for i in range(3))
dto = time.time() - t
print("CPU oracle (C restatement), the 5 Mb: %.2f s -> %.2f Mbp/s; identical: %s" % (dto, 5.0 / dto, same))
ref = os.path.join(here, "..", "..", "oracle", "_ref", "nextpolish1")
if os.path.exists(ref):
    d = tempfile.mkdtemp(prefix="np1k_")
    one = nat.Stream.synth([1000000], depth=50.0, seed=77, with_qual=1, draft_lower=0.004)
    fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
    one.write_files(fa, bam)
    t = time.time()
    subprocess.run([ref, "kmercount", fa, bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    dtr = time.time() - t
    print("compiled reference CLI kmercount, 1 Mb / 50x from files, 1 core: %.2f s -> %.2f Mbp/s" % (dtr, 1.0 / dtr))
