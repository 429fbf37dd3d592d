"""Device bytes of one uploaded batch before / after a score_chain pass (inputs vs inputs + work buffers).
usage: np1_batch_mem.py <batch Mb> [depth]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nextpolish_amd import _native as nat
from nextpolish_amd.device import Context
mb = float(sys.argv[1]); depth = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
t = time.time(); st = nat.Stream.synth([int(mb * 1e6)], depth=depth, seed=1); tg = time.time() - t
c = Context(0); b = c.upload(st); d0 = b.device_bytes()
cfg = nat.default_config()
t = time.time(); b.score_chain(cfg); t1 = time.time() - t
t = time.time(); b.score_chain(cfg); t2 = time.time() - t
d1 = b.device_bytes()
print("batch %.0f Mb @%gx: synth %.1f s, %d reads, inputs %.2f GB (%.1f B/bp), inputs+work %.2f GB (%.1f B/bp), pass %.3f s then %.3f s" %
      (mb, depth, tg, st.n_reads, d0 / 1e9, d0 / (mb * 1e6), d1 / 1e9, d1 / (mb * 1e6), t1, t2))
