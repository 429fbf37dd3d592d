#!/usr/bin/env python
"""snp_phase on one directory of np1_phase_case.py: the product on the GPU (np1_batch_snp_phase, timed over `reps` runs on the
resident batches), checked against the oracle (timed once, one core) and, where oracle/_ref exists, against the compiled reference.
   python tests/tools/np1_phase_bench.py DIR|synth:<Mb> [reps] [--no-gpu] [--no-oracle]"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
from conftest import parse_cli_fasta, ref_binary  # noqa: E402
from nextpolish_amd import _native as nat  # noqa: E402

d = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
fa, sr, lr = os.path.join(d, "g.fa"), os.path.join(d, "sr.bam"), os.path.join(d, "lr.bam")
t = time.time()
if d.startswith("synth:"):      # synth:<Mb>: the diploid generator of the library, contigs of 4 Mb
    mb = float(d[6:])
    n = max(1, int(mb // 4))
    s, l = nat.Stream.synth_diploid([int(mb * 1e6 / n)] * n, seed=9090, sr_holes=2)
    cfg = nat.default_config()
    cfg.read_tlen, cfg.read_len = 2000, 150
else:
    s, l = nat.Stream.load(fa, sr, with_qual=True), nat.Stream.load(fa, lr, with_qual=True)
    cfgp = nat.lib().config_init(fa.encode(), sr.encode(), lr.encode())
    cfg = cfgp.contents
print("loaded %d short + %d long records of %d contigs (%d bp) in %.1f s" % (s.n_reads, l.n_reads, s.n_contigs, int(s.ctg_len.sum()), time.time() - t))
ocfg = ob.default_config(read_tlen=cfg.read_tlen, read_len=cfg.read_len)
bp = int(s.ctg_len.sum())
want = None
if "--no-oracle" not in sys.argv:
    t = time.time()
    want = [ob.snp_phase(s, l, i, ocfg) for i in range(s.n_contigs)]
    t_or = time.time() - t
    print("oracle (1 core): %.2f s = %.2f Mbp/s; stages %s" % (t_or, bp / t_or / 1e6, ob.snp_phase_stats()))
if want is not None and not d.startswith("synth:") and ref_binary() and os.path.exists("/root/reference"):
    t = time.time()
    p = subprocess.run([ref_binary(), "snpphase", fa, sr, lr], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    t_ref = time.time() - t
    ref = parse_cli_fasta(p.stdout.decode())
    print("compiled reference (1 core, from files): %.2f s = %.2f Mbp/s; equal to the oracle: %s" % (t_ref, bp / t_ref / 1e6, [ref[n] for n in s.names] == want))
if "--no-gpu" not in sys.argv:
    from nextpolish_amd import device
    ctx = device.Context(0)
    b, bl = ctx.upload(s), ctx.upload(l)
    times = []
    for _ in range(reps):
        t = time.time()
        b.snp_phase(bl, cfg)
        got = b.results()
        times.append(time.time() - t)
    print("GPU np1_batch_snp_phase (resident batches, incl. result download): %s s; best %.1f Mbp/s; equal to the oracle: %s"
          % (["%.3f" % x for x in times], bp / min(times) / 1e6, (got == want) if want is not None else "not checked"))
    want = want or got
    bl.close(); b.close(); ctx.close()
print("md5", [hashlib.md5(w.encode()).hexdigest() for w in want])
