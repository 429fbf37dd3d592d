import sys, ctypes as C
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from nextpolish_amd import _native as nat
from nextpolish_amd.device import Context
import oracle_binding as ob
L = nat.lib()

ctx = Context(0)
for lens, depth in [([3000,900,200],30),([200000],50)]:
    st = nat.Stream.synth(lens, depth=depth, seed=1001)
    b = ctx.upload(st); b.score_chain()
    out=(C.c_uint32*24)(); L.np1_batch_debug_counters(b.handle,out,24); print('counters', list(out))
    res=b.results()
    print(lens, [res[i]==ob.score_chain(st,i) for i in range(st.n_contigs)])
    b.close()
