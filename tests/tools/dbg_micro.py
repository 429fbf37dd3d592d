import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nextpolish_amd import _native as nat
from nextpolish_amd.device import Context
import oracle_binding as ob
from fuzzgen import random_case
ctx = Context(0)
bad = 0
for seed in range(400):
    contigs, reads = random_case(seed)
    st = nat.Stream.from_reads(contigs, reads)
    try:
        got = ctx.score_chain(st)
    except Exception as e:
        print('seed', seed, 'EXC', e); bad += 1; continue
    for i in range(st.n_contigs):
        want = ob.score_chain(st, i)
        if got[i] != want:
            print('seed', seed, 'contig', i, 'len', len(got[i]), len(want)); print(' got ', got[i][:80]); print(' want', want[:80]); bad += 1
    if bad >= 4: break
print('bad', bad)
