"""Both libraries at work in ONE process, alternating: the short-read path from files (device ingest + score_chain on pipe lanes), drop-in calls on
the bundled real alignments, and the long-read consensus (nextpolish2.so) -- the shape of a long-lived worker that takes whatever the driver
hands it, and of the test suite's own process in which rounds 3 and 4 saw the GPU memory fault of DESIGN.md section 12."""
import hashlib
import json
import os

import pytest

from nextpolish_amd import _native as nat
from nextpolish_amd.device import Pipe
import np2_cases
import oracle_binding as ob
import ref2_binding as rb
import test_real_data as trd
from conftest import ROOT

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# NP_STRESS_REPS=<n> repeats the test n times in the process (tests/tools/r6_hang_hunt.sh: the body makes a few thousand device allocations and
# releases per pass with both libraries' streams alive -- the pattern in which the one-process suite stopped inside hipFree)
@pytest.mark.parametrize("rep", range(int(os.environ.get("NP_STRESS_REPS", "1"))))
def test_short_read_and_long_read_legs_back_to_back_in_one_process(rep, tmp_path):
    gold2 = json.load(open(os.path.join(HERE, "golden", "np2_golden.json")))["cases"]
    L2 = rb.bind(os.path.join(ROOT, "nextpolish_amd", "lib", "nextpolish2.so"))
    st = nat.Stream.synth([2500000, 900000, 40000], depth=30, seed=515)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "r.bam")
    st.write_files(fa, bam)
    want = {n: hashlib.md5(ob.score_chain(st, i).encode()).hexdigest() for i, n in enumerate(st.names)}
    cases = [c for c in np2_cases.CASES if c[0] in ("ont_20x_two_contigs", "ont_lq_regions", "hifi_20x")]
    pipe = Pipe(0, lanes=2)
    try:
        for rnd in range(3):
            got = dict(pipe.run_files(fa, bam, batch_bp=1000000))      # three batches over two lanes, results through the in-order sink
            assert {n: hashlib.md5(s.encode()).hexdigest() for n, s in got.items()} == want, "round %d short reads" % rnd
            cid, kw, rt = cases[rnd]
            d = tmp_path / ("lr%d" % rnd)
            d.mkdir()
            fa2, fofn, contigs = np2_cases.materialise(kw, str(d))
            res = rb.polish(L2, fa2, fofn, read_type=rt)
            for n, _ in contigs:
                assert res[n][0][0] == gold2[cid]["expected"][n], "round %d long reads %s %s" % (rnd, cid, n)
            trd.dropin_symbols_body("r1.slice")                      # config_init, score_chain and kmer_count per contig on real bwa alignments
    finally:
        pipe.close()
