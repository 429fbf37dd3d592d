"""The CPU oracle (oracle/np1_oracle.c) against (a) the committed golden vectors that were generated from
the real reference binary (tests/golden/make_golden.py) and (b), when oracle/_ref/nextpolish1 is present,
the reference binary itself on fresh fuzzed inputs.  This is what pins the oracle."""
import hashlib
import json
import os
import subprocess

import pytest

from nextpolish_amd import _native as nat
import oracle_binding as ob
from conftest import ROOT, ref_binary, run_ref
from fuzzgen import random_case

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "np1_golden.json")))


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


@pytest.mark.parametrize("k", range(len(GOLD["synth"])))
def test_golden_synth_score_chain_and_kmer_count(k):
    g = GOLD["synth"][k]
    kw = dict(g["params"])
    lens = kw.pop("contig_len")
    st = nat.Stream.synth(lens, with_qual=1, **kw)
    assert st.n_reads == g["n_reads"]          # the generator itself is part of the fixture
    cfg = ob.default_config(read_tlen=g["read_tlen"], read_len=g["read_len"])
    for i, exp in enumerate(g["score_chain"]):
        got = ob.score_chain(st, i)
        assert (len(got), md5(got)) == (exp["len"], exp["md5"]), "score_chain %s" % exp["name"]
    for i, exp in enumerate(g["kmer_count"]):
        got = ob.kmer_count(st, i, cfg)
        assert (len(got), md5(got)) == (exp["len"], exp["md5"]), "kmer_count %s" % exp["name"]


def test_golden_micro_cases():
    for g in GOLD["micro"]:
        contigs = [tuple(c) for c in g["contigs"]]
        for r in g["reads"]:
            r["cigar"] = [tuple(x) for x in r["cigar"]]
        st = nat.Stream.from_reads(contigs, g["reads"])
        for i, exp in enumerate(g["score_chain"]):
            assert ob.score_chain(st, i) == exp, "micro seed %d contig %d" % (g["seed"], i)


def test_fuzzgen_is_stable():
    """The committed micro fixtures were produced by tests/fuzzgen.py; a silent change of the generator would
    make the GPU fuzz tests wander away from what the goldens pinned."""
    for g in GOLD["micro"][:10]:
        contigs, reads = random_case(g["seed"])
        assert [list(c) for c in contigs] == [list(c) for c in g["contigs"]]
        assert len(reads) == len(g["reads"])


needs_ref = pytest.mark.skipif(ref_binary() is None, reason="oracle/_ref/nextpolish1 not built (needs /root/reference)")


@needs_ref
def test_oracle_vs_reference_micro_fuzz(tmp_path):
    fa, bam = str(tmp_path / "z.fa"), str(tmp_path / "z.bam")
    for seed in range(1000, 1150):
        contigs, reads = random_case(seed)
        st = nat.Stream.from_reads(contigs, reads)
        st.write_files(fa, bam)
        ref = run_ref("scorechain", fa, bam)
        for i, (n, _) in enumerate(contigs):
            assert ob.score_chain(st, i) == ref[n], "seed %d contig %s" % (seed, n)


def lowercase_micro_case(seed):
    """micro case of fuzzgen with a tenth of the draft in lower case and contigs of 40-500 bases with 10-150 reads: the shape
    that makes kmer_count / snp_valid fall back to their level-1 vote on thin coverage"""
    import random
    rng = random.Random(seed)
    contigs, reads = random_case(seed, n_contigs=2, max_len=rng.choice([40, 160, 500]), max_reads=rng.choice([10, 40, 150]), odd_letters=rng.random() < 0.5,
                                 odd_cigars=rng.random() < 0.7)
    contigs = [(n, "".join(c.lower() if rng.random() < 0.1 else c for c in d)) for n, d in contigs]
    for r in reads:
        r["qual"] = bytes(r["qual"])
    return contigs, reads


@needs_ref
def test_oracle_vs_reference_three_tasks_on_lowercase_micro_cases(tmp_path):
    """score_chain, kmer_count and snp_valid of the same files.  Seed 10019: a part at the end of the first contig with no record
    starting behind it -- the record the fallback keeps re-parsing is the contig's last one, not the next contig's first (during
    development: 1 400 cases of this kind, 0 differences after that fix)."""
    import subprocess
    fa, bam = str(tmp_path / "z.fa"), str(tmp_path / "z.bam")
    compared = 0
    for seed in list(range(10000, 10030)) + [10019]:
        contigs, reads = lowercase_micro_case(seed)
        nat.Stream.from_reads(contigs, reads).write_files(fa, bam)
        cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
        cfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
        nat.lib().config_destory(cfgp)
        st = nat.Stream.load(fa, bam, with_qual=True)
        for cmd, fn in (("scorechain", ob.score_chain), ("kmercount", ob.kmer_count), ("snpvalid", ob.snp_valid)):
            try:
                ref = run_ref(cmd, fa, bam)
            except subprocess.CalledProcessError:
                continue          # the reference crashed (snp_valid's null list): nothing to compare
            for i, n in enumerate(st.names):
                got = fn(st, i, cfg)
                if got is not None:
                    assert got == ref[n], "%s seed %d contig %s" % (cmd, seed, n)
                    compared += 1
    assert compared > 150


def thin_multiwindow_stream(seed):
    """contigs of 20-70 kb (several 16 kb index windows) at 1.5-12x with 1-20 % lower case: where kmer_count's level-1 fallback meets
    the chunk lists of the BAM index"""
    import random
    rng = random.Random(seed)
    lens = [rng.choice([20000, 40000, 70000]), rng.choice([3000, 17000, 33000]), rng.choice([500, 16500])]
    st = nat.Stream.synth(lens, depth=rng.choice([1.5, 3, 6, 12]), seed=seed, with_qual=1, weird_rate=0.03, softclip_rate=0.1, draft_lower=rng.choice([0.01, 0.05, 0.2]),
                          read_indel=rng.choice([0.001, 0.01]), lowmapq_rate=rng.choice([0.05, 0.5]), supp_rate=0.02, sec_rate=0.02, unmapped_rate=0.02)
    return st, rng.choice([1, 6])


def deep_multiwindow_stream(seed):
    """contigs of one to three 16 kb windows at 80-250x: more than max_count_kmer spanning mapq-60 records per part, so the first loop
    of kmercount.c:196-207 leaves through its break and a re-used iterator resumes from where the break left it"""
    import random
    rng = random.Random(seed)
    lens = [rng.choice([17000, 24000, 36000]), rng.choice([2500, 16500])]
    st = nat.Stream.synth(lens, depth=rng.choice([80, 120, 250]), seed=seed, with_qual=1, weird_rate=0.02, softclip_rate=0.05, draft_lower=rng.choice([0.01, 0.05, 0.2]),
                          read_indel=rng.choice([0.001, 0.01]), lowmapq_rate=rng.choice([0.02, 0.3]), supp_rate=0.01, sec_rate=0.01, unmapped_rate=0.01)
    return st, rng.choice([1, 6])


@needs_ref
def test_iterator_replay_reproduces_the_reference_where_file_order_does_not(tmp_path):
    """kmer_count / snp_valid with the records' virtual offsets and the BAI handed to the oracle (oracle_binding.Geometry): the
    reference's region iterator is replayed (chunk lists of hts_itr_query, re-use while a part ends before the record behind the
    first chunk, saved offsets, contig.c:982-1043) and the fallback of kmercount.c:212-217 sees the record the reference sees.
    Seeds 44, 100, 102: a part just behind a 16 kb window boundary whose only spanning record is a level-1 read -- "records in
    file order" gets 4 contigs wrong there, the replay none (during development: 150 such files, 0 differences with the replay)."""
    import subprocess
    fa, bam = str(tmp_path / "z.fa"), str(tmp_path / "z.bam")
    plain_wrong = replay_wrong = compared = 0
    for seed in (44, 100, 102, 7, 19, 63):
        st, level = thin_multiwindow_stream(seed)
        st.write_files(fa, bam, level)
        cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
        cfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
        nat.lib().config_destory(cfgp)
        s2 = nat.Stream.load(fa, bam, with_qual=True)
        geom = ob.Geometry(s2, bam)
        for cmd, fn in (("kmercount", ob.kmer_count), ("snpvalid", ob.snp_valid)):
            try:
                ref = run_ref(cmd, fa, bam)
            except subprocess.CalledProcessError:
                continue
            for i, n in enumerate(s2.names):
                plain, replay = fn(s2, i, cfg), fn(s2, i, cfg, geom)
                if plain is not None:
                    plain_wrong += plain != ref[n]
                if replay is not None:
                    replay_wrong += replay != ref[n]
                    compared += 1
    assert compared > 30 and replay_wrong == 0
    assert plain_wrong == 4          # the documented deviation of the record-stream rule


@needs_ref
def test_product_side_replay_feeding_the_vote_body(tmp_path):
    """np1_replay.h (the product's host-side replay of the iterator: RefIndex::query, Scanner) handing kc_part_winner the records of
    the first loop, the buffered record and the passes of the second loop -- driven by the host model the way the device pass will
    be: kmer_count of the thin multi-window files == the compiled reference, incl. the three seeds file order gets wrong"""
    import model_binding as mb
    fa, bam = str(tmp_path / "z.fa"), str(tmp_path / "z.bam")
    n = 0
    for seed in (44, 100, 102, 7, 19):
        st, level = thin_multiwindow_stream(seed)
        st.write_files(fa, bam, level)
        cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
        s2 = nat.Stream.load(fa, bam, with_qual=True)
        got = mb.kmer_count_replay(s2, cfgp.contents, bam)
        nat.lib().config_destory(cfgp)
        ref = run_ref("kmercount", fa, bam)
        for i, name in enumerate(s2.names):
            assert got[i] == ref[name], "seed %d %s" % (seed, name)
            n += 1
    assert n == 15


@needs_ref
@pytest.mark.parametrize("seed", range(6))
def test_oracle_vs_reference_synth(tmp_path, seed):
    fa, bam = str(tmp_path / "s.fa"), str(tmp_path / "s.bam")
    st = nat.Stream.synth([4000 + 911 * seed, 700], depth=[6, 25, 90][seed % 3], seed=500 + seed, with_qual=1,
                          weird_rate=0.03, softclip_rate=0.06, draft_lower=0.02, read_indel=0.001)
    st.write_files(fa, bam)
    sc, kc = run_ref("scorechain", fa, bam), run_ref("kmercount", fa, bam)
    cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
    cfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
    nat.lib().config_destory(cfgp)
    for i, n in enumerate(st.names):
        assert ob.score_chain(st, i) == sc[n]
        assert ob.kmer_count(st, i, cfg) == kc[n]


@needs_ref
def test_oracle_equals_reference_library_for_general_indel_balance_factor():
    """indel_balance_factor_sgs that is not a dyadic fraction: the reference accumulates doubles (contig.c:448); the oracle
    does the same arithmetic in the same order.  Through the reference's shared library (the CLI takes no options)."""
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "nextpolish1.so"))
    L.config_init.restype = C.POINTER(nat.Configure)
    L.config_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.score_chain.restype = C.POINTER(nat.PolishResult)
    L.score_chain.argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for seed, kw in ((1, dict(contig_len=[15000, 4000], depth=40.0)), (2, dict(contig_len=[5000], depth=150.0, read_sub=0.05, read_indel=0.01))):
            kw = dict(kw)
            lens = kw.pop("contig_len")
            st = nat.Stream.synth(lens, seed=seed, **kw)
            fa, bam = os.path.join(td, "g%d.fa" % seed), os.path.join(td, "g%d.bam" % seed)
            st.write_files(fa, bam)
            cfg = L.config_init(fa.encode(), bam.encode(), None)
            for rate in (0.3, 0.55, 1.0 / 3.0, 0.9):
                cfg.contents.indel_balance_factor_sgs = rate
                ocfg = ob.default_config(indel_balance_factor_sgs=rate)
                for i, n in enumerate(st.names):
                    r = L.score_chain(n.encode(), cfg)
                    assert C.string_at(r.contents.contig).decode() == ob.score_chain(st, i, ocfg), "rate %r contig %s" % (rate, n)


@needs_ref
def test_oracle_kmer_count_equals_reference_library_for_general_indel_balance_factor():
    import ctypes as C
    import tempfile
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "nextpolish1.so"))
    L.config_init.restype = C.POINTER(nat.Configure)
    L.config_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.kmer_count.restype = C.POINTER(nat.PolishResult)
    L.kmer_count.argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    with tempfile.TemporaryDirectory() as td:
        for seed in range(3):
            st = nat.Stream.synth([4000 + seed * 97, 700], depth=[4, 6, 10][seed], seed=3100 + seed, with_qual=1, draft_lower=0.03,
                                  read_indel=0.002, softclip_rate=0.05, lowmapq_rate=0.2)
            fa, bam = os.path.join(td, "k%d.fa" % seed), os.path.join(td, "k%d.bam" % seed)
            st.write_files(fa, bam)
            cfg = L.config_init(fa.encode(), bam.encode(), None)
            for rate in (0.3, 0.55):
                cfg.contents.indel_balance_factor_sgs = rate
                ocfg = ob.default_config(read_tlen=cfg.contents.read_tlen, read_len=cfg.contents.read_len, indel_balance_factor_sgs=rate)
                for i, n in enumerate(st.names):
                    r = L.kmer_count(n.encode(), cfg)
                    assert C.string_at(r.contents.contig).decode() == ob.kmer_count(st, i, ocfg), "rate %r contig %s seed %d" % (rate, n, seed)


def test_cigar_in_a_cg_tag_reference_binary_agrees(tmp_path):
    """a record of more than 65 535 CIGAR operations travels as the placeholder '<l_qseq>S<rlen>N' + tag CG:B:I (SAMv1 4.2.2); htslib
    swaps the real CIGAR in while reading, so the reference votes with it -- pinned here on the reference binary itself: writer
    (CG out), loader (CG in), oracle on the loaded stream == reference on the file, for score_chain and kmer_count"""
    import numpy as np
    import struct, gzip
    from fuzzgen import long_record_case
    contigs, reads = long_record_case(7, n_ops=66000, plain_bases=66000)
    st = nat.Stream.from_reads(contigs, reads)
    assert int(st.n_cigar.max()) > 65535
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam)
    raw = gzip.open(bam, "rb").read()
    assert b"CGBI" + struct.pack("<I", int(st.n_cigar.max())) in raw        # the tag is really in the file
    st2 = nat.Stream.load(fa, bam, with_qual=True)
    for f in ["pos", "flag", "n_cigar", "l_qseq", "cigar", "seq"]:
        assert np.array_equal(getattr(st, f), getattr(st2, f)), f
    if ref_binary() is None:
        pytest.skip("reference binary not built")
    assert run_ref("scorechain", fa, bam)["long"] == ob.score_chain(st2, 0)
    cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
    cfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
    nat.lib().config_destory(cfgp)
    assert run_ref("kmercount", fa, bam)["long"] == ob.from_files("kmer_count", fa, bam, cfg)["long"]


def _meets_twice(reads):
    """does a record hold two insertion operations at one reference position (only M and D advance it: contig.c:262-326)?"""
    for r in reads:
        last = False
        for o, _ in r["cigar"]:
            if o == "I":
                if last:
                    return True
                last = True
            elif o in "MD":
                last = False
    return False


@needs_ref
def test_reference_has_no_result_for_two_insertions_at_one_position(tmp_path):
    """'I P I' / 'I N I' (no aligner writes them; SAM allows them).  contig.c:299-320 votes on the insertion columns a second time with a
    context whose predecessor is an insertion column, not the base before them; the chain (contig.c:424-496) then looks that predecessor
    state up in the previous slot, finds none and dereferences NULL.  Pinned on the compiled reference: scorechain dies with SIGSEGV on
    most such inputs -- there is no result to be identical to, which is why the product refuses these records by name (np1_core.h
    ERR_DOUBLE_INS) instead of inventing one.  (kmercount / snpvalid only chain inside their flagged regions and mostly survive; the product's
    region walk takes these records the same way: tests/test_model.py.)"""
    fa, bam = str(tmp_path / "d.fa"), str(tmp_path / "d.bam")
    died = n = 0
    for seed in range(60):
        contigs, reads = random_case(seed, double_ins=True)
        if not _meets_twice(reads):
            continue
        nat.Stream.from_reads(contigs, reads).write_files(fa, bam)
        p = subprocess.run([ref_binary(), "scorechain", fa, bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        n += 1
        died += p.returncode == -11
    assert n > 40 and died > n // 2, (died, n)
