"""One very large contig (default 250 Mb @30x, ~50 M records) through score_chain on the GPU: per-stage times, and the polished
string against the oracle (CPU, one core, ~2 Mbp/s).  usage: np1_big_contig_check.py [Mb] [depth] [--no-oracle]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nextpolish_amd import _native as nat
from nextpolish_amd.device import Context
mb = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 250.0
depth = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 30.0
t = time.time(); st = nat.Stream.synth([int(mb * 1e6)], depth=depth, seed=77); print("synth %.1f s, %d records" % (time.time() - t, st.n_reads), flush=True)
c = Context(0); b = c.upload(st); cfg = nat.default_config()
b.score_chain(cfg)
ms = b.score_chain(cfg, timed=True)
print("stages (ms):", {k: round(v, 2) for k, v in ms.items()}, flush=True)
got = b.results()[0]
print("GPU: len %d md5 %s" % (len(got), hashlib.md5(got.encode()).hexdigest()), flush=True)
if "--no-oracle" not in sys.argv:
    import oracle_binding as ob
    t = time.time(); want = ob.score_chain(st, 0); dt = time.time() - t
    print("oracle: len %d md5 %s (%.1f s, %.2f Mbp/s on one core)" % (len(want), hashlib.md5(want.encode()).hexdigest(), dt, mb / dt), flush=True)
    print("IDENTICAL" if got == want else "DIFFERENT")
