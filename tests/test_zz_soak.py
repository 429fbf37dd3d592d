"""Stress last (file names sort the suite: every parity test runs before this one).

One process, 200 alternating score_chain / kmer_count drop-in calls -- the call pattern of source/lib/nextpolish1.py:181-189,219-224 in a
long-lived worker -- with a large batch (130 Mb, 26 M records) uploaded, run and freed every 50 calls, so that the allocator hands the small
buffers of the next calls the address ranges the large ones just left.  Round 5's driver run stopped for good inside the release of the large
batch (hipFree never returned: DESIGN.md section 12); since round 6 releases go to the allocator cache of csrc/np_devalloc.h, and this test
also checks what the cache did."""
import ctypes as C
import hashlib
import threading

import pytest

from nextpolish_amd import _native as nat
from nextpolish_amd import device as npdev
import oracle_binding as ob
from test_real_data import digest, sr_files

pytestmark = pytest.mark.gpu


def test_gpu_soak_alternating_dropin_calls_with_large_batches_in_between():
    g, fa, bam = sr_files("r1.slice")
    L = nat.lib()
    cfg = L.config_init(fa.encode(), bam.encode(), None)
    names = sorted(g["score_chain"])
    big = nat.Stream.synth([100_000_000, 30_000_000], depth=30, seed=77)
    # the oracle's answer for the large batch, computed on two host threads while the device works through the first 50 calls
    want = [None, None]

    def oracle(i):
        want[i] = hashlib.md5(ob.score_chain(big, i).encode()).hexdigest()

    helpers = [threading.Thread(target=oracle, args=(i,)) for i in range(2)]
    for t in helpers:
        t.start()
    ctx = npdev.Context()
    stats0 = (C.c_uint64 * 8)()
    L.np1_alloc_stats(stats0)
    seen = []
    for k in range(200):
        n = names[k % len(names)]
        if k % 2 == 0:
            r = L.score_chain(n.encode(), cfg)
            assert digest(C.string_at(r.contents.contig).decode()) == g["score_chain"][n], "call %d score_chain %s" % (k, n)
        else:
            r = L.kmer_count(n.encode(), cfg)
            assert digest(C.string_at(r.contents.contig).decode()) == g["kmer_count"][n], "call %d kmer_count %s" % (k, n)
        L.polishresult_destory(r)
        if k % 50 == 49:
            b = ctx.upload(big)
            b.score_chain()
            seen.append([hashlib.md5(s.encode()).hexdigest() for s in b.results()])
            b.close()
    ctx.close()
    L.config_destory(cfg)
    for t in helpers:
        t.join()
    assert all(row == want for row in seen), "the large batch differs from the oracle: %r vs %r" % (seen, want)
    stats = (C.c_uint64 * 8)()
    L.np1_alloc_stats(stats)
    if stats[7]:      # (NP_DEVCACHE_MB=0 switches the cache off)
        hits, misses = stats[0] - stats0[0], stats[1] - stats0[1]
        assert hits > 10 * misses, "the allocator cache served %d of %d requests" % (hits, hits + misses)
        assert stats[4] <= stats[7], "idle bytes %d above the bound %d" % (stats[4], stats[7])
