import sys, ctypes as C
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nextpolish_amd import _native as nat
from nextpolish_amd.device import Context
import oracle_binding as ob
from fuzzgen import random_case
L = nat.lib()
ctx = Context(0)
for seed in [int(x) for x in sys.argv[1:]]:
    contigs, reads = random_case(seed)
    st = nat.Stream.from_reads(contigs, reads)
    b = ctx.upload(st); b.score_chain()
    out = (C.c_uint32 * 24)(); L.np1_batch_debug_counters(b.handle, out, 24)
    res = b.results()
    print('seed', seed, 'counters', list(out), [res[i] == ob.score_chain(st, i) for i in range(st.n_contigs)])
    S = L.np1_batch_debug_slots(b.handle, 0, None, 0)
    info = np.zeros(S, np.uint8); sres = np.zeros(S, np.uint16); srec = np.zeros(S, np.uint32)
    L.np1_batch_debug_slots(b.handle, 0, info.ctypes.data, S)
    L.np1_batch_debug_slots(b.handle, 1, sres.ctypes.data, S)
    L.np1_batch_debug_slots(b.handle, 2, srec.ctypes.data, S)
    print(' info', ' '.join('%02x' % x for x in info[:48]))
    print(' res ', ' '.join('%03x' % x for x in sres[:48]))
    print(' rec ', ' '.join('%d' % (x if x != 0xffffffff else -1) for x in srec[:48]))
    b.close()
