"""Task 3 (snp_phase, reference: source/lib/snpphase.c:87-903): short reads and long reads of the same contig find the heterozygous
sites, settle them where the evidence is one-sided, correct the low-depth stretches with both streams and phase neighbouring sites
through the reads that link them.  CPU: the oracle restatement against goldens the compiled reference produced (real bwa + minimap2
alignments and seeded diploid workloads) and against the compiled reference itself; the device's stage bodies driven on the host
(tests/model) against the oracle.  GPU: the product (np1_batch_snp_phase, the drop-in `snp_phase` symbol, the CLI, the Python
caller) against the oracle and the same goldens.

Inputs for which the reference reads through a null or unset pointer make the oracle return None and the product fail loudly."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import pytest

import oracle_binding as ob
import snpphase_gen
from conftest import parse_cli_fasta, ref_binary
from nextpolish_amd import _native as nat

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REAL = os.path.join(HERE, "golden", "real")
GOLD = json.load(open(os.path.join(HERE, "golden", "snpphase_golden.json")))


def digest(s):
    return {"len": len(s), "md5": hashlib.md5(s.encode()).hexdigest()}


def streams(params):
    ctgs, srs, lrs = snpphase_gen.make_case(**params)
    return nat.Stream.from_reads(ctgs, srs), nat.Stream.from_reads(ctgs, lrs)


def fuzz_params(seed):
    return dict(seed=seed, lens=(600 + 37 * (seed % 13), 2500 + (seed % 7) * 300), sr_depth=[4, 8, 15, 30, 60][seed % 5], lr_depth=[3, 10, 25, 40][seed % 4],
                het=[0.002, 0.01, 0.03][seed % 3], het_indel=[0.0, 0.001, 0.005][(seed // 3) % 3], draft_err=[0.001, 0.01][(seed // 2) % 2],
                lower=[0, 0.1][(seed // 5) % 2], sr_holes=[0, 2, 5][(seed // 7) % 3], lr_err=[0.02, 0.08][(seed // 11) % 2], lr_len=[800, 1500, 3000][(seed // 13) % 3])


def real_streams(g):
    fa = os.path.join(REAL, g["fasta"])
    return nat.Stream.load(fa, os.path.join(REAL, g["sr"]), with_qual=True), nat.Stream.load(fa, os.path.join(REAL, g["lr"]), with_qual=True)


def real_cfg(g):
    """(read_tlen, read_len) as config_init derives them from the short-read BAM"""
    cfgp = nat.lib().config_init(os.path.join(REAL, g["fasta"]).encode(), os.path.join(REAL, g["sr"]).encode(), os.path.join(REAL, g["lr"]).encode())
    v = cfgp.contents.read_tlen, cfgp.contents.read_len
    nat.lib().config_destory(cfgp)
    return v


# ------------------------------------------------------------------------------------------------------------ CPU


@pytest.mark.parametrize("k", range(len(GOLD["synth"])))
def test_oracle_matches_reference_goldens_synth(k):
    g = GOLD["synth"][k]
    s, l = streams(g["params"])
    cfg = ob.default_config(read_tlen=g["read_tlen"], read_len=g["read_len"])
    for i, exp in enumerate(g["snp_phase"]):
        assert digest(ob.snp_phase(s, l, i, cfg)) == exp, "synth %d contig %d" % (k, i)


@pytest.mark.parametrize("tag", sorted(GOLD["real"]))
def test_oracle_matches_reference_goldens_real_alignments(tag):
    g = GOLD["real"][tag]
    s, l = real_streams(g)
    tlen, rlen = real_cfg(g)
    cfg = ob.default_config(read_tlen=tlen, read_len=rlen)
    for i, n in enumerate(s.names):
        assert digest(ob.snp_phase(s, l, i, cfg)) == g["snp_phase"][n], "%s %s" % (tag, n)


def test_the_workloads_reach_every_stage():
    """sites found / kept / with insertion columns / asked of the long reads, long-read votes, settled sites, low-depth regions,
    links, sites re-written by the chain, long-read links: all of it happens in the goldens' inputs (np1_oracle.c: g_sp_stats)"""
    tot = [0] * 10
    for g in GOLD["synth"]:
        s, l = streams(g["params"])
        cfg = ob.default_config(read_tlen=g["read_tlen"], read_len=g["read_len"])
        for i in range(s.n_contigs):
            ob.snp_phase(s, l, i, cfg)
            tot = [a + b for a, b in zip(tot, ob.snp_phase_stats())]
    g = GOLD["real"]["s30+ont"]     # (a site settled by its spanning reads is rare in the small synthetic contigs)
    s, l = real_streams(g)
    tlen, rlen = real_cfg(g)
    for i in range(s.n_contigs):
        ob.snp_phase(s, l, i, ob.default_config(read_tlen=tlen, read_len=rlen))
        tot = [a + b for a, b in zip(tot, ob.snp_phase_stats())]
    assert all(t > 0 for t in tot), tot


needs_ref = pytest.mark.skipif(ref_binary() is None, reason="oracle/_ref/nextpolish1 not built (needs /root/reference)")


def run_ref3(fa, sr, lr):
    p = subprocess.run([ref_binary(), "snpphase", fa, sr, lr], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    return parse_cli_fasta(p.stdout.decode()) if p.returncode == 0 else None


@needs_ref
def test_goldens_still_match_compiled_reference():
    for tag, g in GOLD["real"].items():
        got = run_ref3(os.path.join(REAL, g["fasta"]), os.path.join(REAL, g["sr"]), os.path.join(REAL, g["lr"]))
        assert {n: digest(s) for n, s in got.items()} == g["snp_phase"]


@needs_ref
def test_oracle_vs_reference_fuzz(tmp_path):
    """Diploid contigs over a grid of depths, heterozygosity, indel rates, coverage holes and lower case (during development:
    1 360 contigs, 0 differences, no crash of the reference)."""
    fa, sr, lr = str(tmp_path / "s.fa"), str(tmp_path / "sr.bam"), str(tmp_path / "lr.bam")
    compared = 0
    for seed in range(1000, 1040):
        s, l = streams(fuzz_params(seed))
        s.write_files(fa, sr)
        l.write_files(str(tmp_path / "l.fa"), lr)
        ref = run_ref3(fa, sr, lr)
        if ref is None:
            continue
        cfgp = nat.lib().config_init(fa.encode(), sr.encode(), lr.encode())
        cfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
        nat.lib().config_destory(cfgp)
        for i, n in enumerate(s.names):
            got = ob.snp_phase(s, l, i, cfg)
            if got is None:
                continue
            assert got == ref[n], "seed %d contig %s" % (seed, n)
            compared += 1
    assert compared > 70


def _model_case(s, l, tlen, rlen):
    import model_binding as mb
    cfg = nat.default_config()
    cfg.read_tlen, cfg.read_len = tlen, rlen
    ocfg = ob.default_config(read_tlen=tlen, read_len=rlen)
    want = [ob.snp_phase(s, l, i, ocfg) for i in range(s.n_contigs)]
    if any(w is None for w in want):
        with pytest.raises(ValueError):
            mb.snp_phase(s, l, cfg)
        return False
    assert mb.snp_phase(s, l, cfg) == want
    return True


def test_host_model_of_the_stage_bodies_matches_oracle():
    """np1_phase.h (histogram, slot verdicts, sites and anchors, sparse low-depth walk, re-slotting, site verdicts, two-stream region
    chain with the third-generation rule, link parser) and np1_phase_host.h (link regions, marks, chain) driven on the host the way
    np1_batch_snp_phase drives the kernels (during development: 780 fuzzed contigs, 0 differences)."""
    n = 0
    for g in GOLD["synth"]:
        s, l = streams(g["params"])
        n += _model_case(s, l, g["read_tlen"], g["read_len"])
    for seed in range(2000, 2030):
        s, l = streams(fuzz_params(seed))
        n += _model_case(s, l, 500, 100)
    assert n > 30


def touching(seed):
    ctgs, srs, lrs = snpphase_gen.touching_case(seed)
    return nat.Stream.from_reads(ctgs, srs), nat.Stream.from_reads(ctgs, lrs)


def test_low_depth_regions_that_touch():
    """two merged low-depth regions sharing a base: the three loops of ts_correct_lower_depth over all regions, not region by region"""
    for seed in range(12):
        s, l = touching(seed)
        ob.snp_phase(s, l, 0, ob.default_config(read_tlen=500, read_len=100))
        assert ob.snp_phase_stats()[6] == 4
        assert _model_case(s, l, 500, 100)


@needs_ref
def test_low_depth_regions_that_touch_vs_reference(tmp_path):
    fa, sr, lr = str(tmp_path / "s.fa"), str(tmp_path / "sr.bam"), str(tmp_path / "lr.bam")
    changed = 0
    for seed in range(12):
        s, l = touching(seed)
        s.write_files(fa, sr)
        l.write_files(str(tmp_path / "l.fa"), lr)
        ref = run_ref3(fa, sr, lr)
        cfgp = nat.lib().config_init(fa.encode(), sr.encode(), lr.encode())
        cfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
        nat.lib().config_destory(cfgp)
        got = ob.snp_phase(s, l, 0, cfg)
        assert got == ref["tig0"], seed
        changed += got != s.contig_draft(0).decode()
    assert changed > 6


@needs_ref
def test_cpp_diploid_generator_workload_vs_reference(tmp_path):
    """the workload bench.py's snp_phase leg times (np1_stream_synth_diploid): reference == oracle == host model"""
    s, l = nat.Stream.synth_diploid([150000, 40000], seed=11, sr_holes=2)
    fa, sr, lr = str(tmp_path / "s.fa"), str(tmp_path / "sr.bam"), str(tmp_path / "lr.bam")
    s.write_files(fa, sr)
    l.write_files(str(tmp_path / "l.fa"), lr)
    ref = run_ref3(fa, sr, lr)
    cfgp = nat.lib().config_init(fa.encode(), sr.encode(), lr.encode())
    tlen, rlen = cfgp.contents.read_tlen, cfgp.contents.read_len
    nat.lib().config_destory(cfgp)
    cfg = ob.default_config(read_tlen=tlen, read_len=rlen)
    for i, n in enumerate(s.names):
        assert ob.snp_phase(s, l, i, cfg) == ref[n], n
    assert _model_case(s, l, tlen, rlen)


def adversarial(seed):
    ctgs, srs, lrs = snpphase_gen.adversarial_case(seed)
    return nat.Stream.from_reads(ctgs, srs), nat.Stream.from_reads(ctgs, lrs)


@needs_ref
def test_oracle_vs_reference_on_adversarial_inputs(tmp_path):
    """odd CIGAR shapes and letters, thin coverage: where the reference answers, the oracle answers the same; every crash of the
    reference is an input the oracle calls undefined (during development: 400 cases, 69 crashes, 0 differences)"""
    fa, sr, lr = str(tmp_path / "s.fa"), str(tmp_path / "sr.bam"), str(tmp_path / "lr.bam")
    compared = crashes = 0
    for seed in range(90):
        s, l = adversarial(seed)
        s.write_files(fa, sr)
        l.write_files(str(tmp_path / "l.fa"), lr)
        ref = run_ref3(fa, sr, lr)
        cfgp = nat.lib().config_init(fa.encode(), sr.encode(), lr.encode())
        cfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
        nat.lib().config_destory(cfgp)
        s2, l2 = nat.Stream.load(fa, sr, with_qual=True), nat.Stream.load(fa, lr, with_qual=True)
        got = [ob.snp_phase(s2, l2, i, cfg) for i in range(s2.n_contigs)]
        if ref is None:
            crashes += 1
            assert any(g is None for g in got), "seed %d: the reference crashed on an input the oracle calls defined" % seed
            continue
        for i, n in enumerate(s2.names):
            if got[i] is not None:
                assert got[i] == ref[n], "seed %d contig %s" % (seed, n)
                compared += 1
    assert compared > 100 and crashes > 5


def test_host_model_on_adversarial_inputs():
    """incl. low-depth regions that touch, where the second region is scored on the list the first one merged in place (base.c:123-146)"""
    n = 0
    for seed in range(100, 190):
        s, l = adversarial(seed)
        n += _model_case(s, l, 500, 100)
    assert n > 60


PARAM_VALUES = dict(trim_len_edge=[0, 1, 2, 4], ext_len_edge=[1, 2, 3, 5], min_map_quality=[0, 10, 30], min_depth_snp=[1, 3, 6], min_count_snp=[2, 5, 10],
                    min_count_snp_link=[1, 5, 12], ploidy=[1, 2, 3, 4.0], indel_balance_factor_lgs=[0.1, 0.33, 0.5, 0.77], max_indel_factor_lgs=[0.1, 0.21, 0.4],
                    max_snp_factor_lgs=[0.3, 0.53, 0.8], min_snp_factor_sgs=[0.1, 0.34, 0.6], max_clip_ratio_lgs=[0.05, 0.4], max_clip_ratio_sgs=[0.05, 0.15, 0.3],
                    max_variant_count_lgs=[150000, 3000, 500])


def random_parameters(seed):
    """about half of the algorithm parameters of task 3 away from their defaults (config.c:8-38)"""
    import random
    rng = random.Random(seed)
    return {k: rng.choice(v) for k, v in PARAM_VALUES.items() if rng.random() < 0.5}


def configs_with(ov, read_tlen, read_len):
    cfg = nat.default_config()
    cfg.read_tlen, cfg.read_len = read_tlen, read_len
    ocfg = ob.default_config(read_tlen=read_tlen, read_len=read_len)
    for k, v in ov.items():
        setattr(cfg, k, type(getattr(cfg, k))(v))
        setattr(ocfg, k, type(getattr(ocfg, k))(v))
    return cfg, ocfg


@needs_ref
def test_parameters_vs_reference_library(tmp_path):
    """the CLI takes no options: through the reference's shared library, Configure fields set by hand (during development: 400
    parameter sets, 0 differences)"""
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "nextpolish1.so"))
    L.config_init.restype = C.POINTER(nat.Configure)
    L.config_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.snp_phase.restype = C.POINTER(nat.PolishResult)
    L.snp_phase.argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    import model_binding as mb
    fa, sr, lr = str(tmp_path / "s.fa"), str(tmp_path / "sr.bam"), str(tmp_path / "lr.bam")
    for seed in range(4000, 4016):
        s, l = streams(fuzz_params(seed))
        s.write_files(fa, sr)
        l.write_files(str(tmp_path / "l.fa"), lr)
        rcfg = L.config_init(fa.encode(), sr.encode(), lr.encode())
        ov = random_parameters(seed)
        for k, v in ov.items():
            setattr(rcfg.contents, k, type(getattr(rcfg.contents, k))(v))
        cfg, ocfg = configs_with(ov, rcfg.contents.read_tlen, rcfg.contents.read_len)
        s2, l2 = nat.Stream.load(fa, sr, with_qual=True), nat.Stream.load(fa, lr, with_qual=True)
        got_model = mb.snp_phase(s2, l2, cfg)
        for i, n in enumerate(s2.names):
            r = L.snp_phase(n.encode(), rcfg)
            want = C.string_at(r.contents.contig).decode()
            assert ob.snp_phase(s2, l2, i, ocfg) == want, "seed %d %s %r" % (seed, n, ov)
            assert got_model[i] == want, "model: seed %d %s %r" % (seed, n, ov)


def test_host_model_on_real_alignments():
    g = GOLD["real"]["s30+ont"]
    s, l = real_streams(g)
    assert _model_case(s, l, *real_cfg(g))


# ------------------------------------------------------------------------------------------------------------ GPU


@pytest.fixture(scope="module")
def ctx():
    from nextpolish_amd import device
    c = device.Context(0)
    yield c
    c.close()


def _check(ctx, s, l, read_tlen=500, read_len=100):
    """product == oracle per contig; a batch holding a contig the reference has no result for must fail loudly"""
    cfg = nat.default_config()
    cfg.read_tlen, cfg.read_len = read_tlen, read_len
    ocfg = ob.default_config(read_tlen=read_tlen, read_len=read_len)
    want = [ob.snp_phase(s, l, i, ocfg) for i in range(s.n_contigs)]
    b, bl = ctx.upload(s), ctx.upload(l)
    try:
        if any(w is None for w in want):
            with pytest.raises(RuntimeError, match="no defined result"):
                b.snp_phase(bl, cfg)
            return False
        b.snp_phase(bl, cfg)
        got = b.results()
    finally:
        bl.close()
        b.close()
    for i in range(s.n_contigs):
        assert len(got[i]) == len(want[i]), "contig %d: length %d != %d" % (i, len(got[i]), len(want[i]))
        if got[i] != want[i]:
            k = next(j for j in range(len(want[i])) if got[i][j] != want[i][j])
            raise AssertionError("snp_phase contig %d differs at %d: %r vs %r" % (i, k, got[i][max(0, k - 8):k + 8], want[i][max(0, k - 8):k + 8]))
    return True


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(GOLD["synth"])))
def test_gpu_matches_oracle_and_goldens_synth(ctx, k):
    g = GOLD["synth"][k]
    s, l = streams(g["params"])
    assert _check(ctx, s, l, g["read_tlen"], g["read_len"])
    cfg = nat.default_config()
    cfg.read_tlen, cfg.read_len = g["read_tlen"], g["read_len"]
    b, bl = ctx.upload(s), ctx.upload(l)
    b.snp_phase(bl, cfg)
    assert [digest(x) for x in b.results()] == g["snp_phase"]
    bl.close()
    b.close()


@pytest.mark.gpu
def test_gpu_matches_oracle_on_fuzzed_diploids(ctx):
    n = 0
    for seed in range(3000, 3030):
        s, l = streams(fuzz_params(seed))
        n += 1 if _check(ctx, s, l) else 0
    assert n > 20


def ultra_long(seed=5):
    """a 120 kb diploid contig whose few long reads each carry more than 65 535 CIGAR operations (what travels in a CG tag)"""
    ctgs, srs, lrs = snpphase_gen.make_case(seed, lens=(120000,), sr_depth=12, lr_depth=5, lr_len=115000, lr_err=0.6, het=0.002)
    s, l = nat.Stream.from_reads(ctgs, srs), nat.Stream.from_reads(ctgs, lrs)
    assert int(l.n_cigar.max()) > 65535
    return s, l


@needs_ref
def test_ultra_long_reads_with_cg_tag_cigars_vs_reference(tmp_path):
    """snpphase.c reads the long-read BAM through htslib, which swaps a CG-tag CIGAR in (bam_tag2cigar): reference on the files ==
    oracle == host model on the streams loaded back from them"""
    fa, sr, lr = str(tmp_path / "s.fa"), str(tmp_path / "sr.bam"), str(tmp_path / "lr.bam")
    s, l = ultra_long()
    s.write_files(fa, sr)
    l.write_files(str(tmp_path / "l.fa"), lr)
    ref = run_ref3(fa, sr, lr)
    cfgp = nat.lib().config_init(fa.encode(), sr.encode(), lr.encode())
    tlen, rlen = cfgp.contents.read_tlen, cfgp.contents.read_len
    nat.lib().config_destory(cfgp)
    s2, l2 = nat.Stream.load(fa, sr, with_qual=True), nat.Stream.load(fa, lr, with_qual=True)
    assert int(l2.n_cigar.max()) == int(l.n_cigar.max())
    assert ob.snp_phase(s2, l2, 0, ob.default_config(read_tlen=tlen, read_len=rlen)) == ref["tig0"]
    assert _model_case(s2, l2, tlen, rlen)


@pytest.mark.gpu
def test_gpu_ultra_long_reads_with_cg_tag_cigars(ctx, tmp_path):
    """the same through the device, in memory and from the files (CLI: the long-read BAM goes through the host loader)"""
    s, l = ultra_long()
    assert _check(ctx, s, l)
    fa, sr, lr = str(tmp_path / "s.fa"), str(tmp_path / "sr.bam"), str(tmp_path / "lr.bam")
    s.write_files(fa, sr)
    l.write_files(str(tmp_path / "l.fa"), lr)
    cfgp = nat.lib().config_init(fa.encode(), sr.encode(), lr.encode())
    ocfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
    nat.lib().config_destory(cfgp)
    s2, l2 = nat.Stream.load(fa, sr, with_qual=True), nat.Stream.load(fa, lr, with_qual=True)
    out = subprocess.run([os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1"), "snpphase", fa, sr, lr], stdout=subprocess.PIPE, timeout=600, check=True).stdout.decode()
    assert parse_cli_fasta(out)["tig0"] == ob.snp_phase(s2, l2, 0, ocfg)


@pytest.mark.gpu
def test_gpu_low_depth_regions_that_touch(ctx):
    for seed in range(12):
        assert _check(ctx, *touching(seed))


@pytest.mark.gpu
def test_gpu_adversarial_inputs(ctx):
    """odd CIGAR shapes and letters, thin coverage, touching low-depth regions; undefined inputs fail loudly"""
    n = 0
    for seed in range(100, 190):
        s, l = adversarial(seed)
        n += 1 if _check(ctx, s, l) else 0
    assert n > 60


@pytest.mark.gpu
def test_gpu_parameters(ctx):
    """algorithm parameters away from their defaults: product == oracle"""
    for seed in range(4000, 4024):
        s, l = streams(fuzz_params(seed))
        cfg, ocfg = configs_with(random_parameters(seed), 500, 100)
        want = [ob.snp_phase(s, l, i, ocfg) for i in range(s.n_contigs)]
        if any(w is None for w in want):
            continue
        b, bl = ctx.upload(s), ctx.upload(l)
        try:
            b.snp_phase(bl, cfg)
            assert b.results() == want, seed
        finally:
            bl.close()
            b.close()


@pytest.mark.gpu
def test_gpu_many_contigs_in_one_batch(ctx):
    s, l = streams(dict(seed=77, lens=tuple(900 + 61 * k for k in range(24)), sr_depth=30, lr_depth=15, het=0.01, het_indel=0.002, sr_holes=1))
    assert _check(ctx, s, l)


@pytest.mark.gpu
def test_gpu_megabase_contigs(ctx):
    """2 x 1.5 Mb + a small contig in one batch pair (8 500 sites, 100 000 links): product == oracle"""
    s, l = nat.Stream.synth_diploid([1500000, 1500000, 30000], seed=5, sr_holes=3)
    assert _check(ctx, s, l, 2000, 150)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(GOLD["real"]))
def test_gpu_batches_of_real_alignments(ctx, tag):
    g = GOLD["real"][tag]
    s, l = real_streams(g)
    tlen, rlen = real_cfg(g)
    assert _check(ctx, s, l, tlen, rlen)


@pytest.mark.gpu
@pytest.mark.parametrize("ingest", ["device", "host"])
def test_gpu_from_files_in_batches(tmp_path, ingest, monkeypatch):
    """np1_pipe_run_phase_files: five contigs in three batches, short reads through the device-side ingest (or the host loader),
    long reads through the host loader, the next batch staged while the device works: every contig == oracle"""
    from nextpolish_amd.device import Pipe
    monkeypatch.setenv("NP1_INGEST", ingest)
    s, l = nat.Stream.synth_diploid([300000, 120000, 200000, 90000, 250000], seed=21, sr_holes=2)
    fa, sr, lr = str(tmp_path / "g.fa"), str(tmp_path / "sr.bam"), str(tmp_path / "lr.bam")
    s.write_files(fa, sr)
    l.write_files(str(tmp_path / "l.fa"), lr)
    cfgp = nat.lib().config_init(fa.encode(), sr.encode(), lr.encode())
    ocfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
    want = {n: ob.snp_phase(s, l, i, ocfg) for i, n in enumerate(s.names)}
    pipe = Pipe(0, 1)
    try:
        got = pipe.run_phase_files(fa, sr, lr, batch_bp=450000, cfg=cfgp.contents)
        assert [n for n, _ in got] == list(s.names)
        assert dict(got) == want
        sub = [s.names[3], s.names[1]]                         # a subset, in another order: index seeks
        got = pipe.run_phase_files(fa, sr, lr, names=sub, batch_bp=16000000, cfg=cfgp.contents)
        assert [n for n, _ in got] == sub and dict(got) == {n: want[n] for n in sub}
    finally:
        pipe.close()
        nat.lib().config_destory(cfgp)


@pytest.mark.gpu
def test_gpu_dropin_symbol_cli_and_caller_on_real_alignments(tmp_path):
    """snp_phase(tigname, cfg) like source/lib/nextpolish1.py:95-96,220; `nextpolish1 snpphase fa bam bam3` like main.c:7-8,39; the
    Python caller with -t 3."""
    g = GOLD["real"]["s30+ont"]
    fa, sr, lr = os.path.join(REAL, g["fasta"]), os.path.join(REAL, g["sr"]), os.path.join(REAL, g["lr"])
    L = nat.lib()
    L.snp_phase.restype = C.POINTER(nat.PolishResult)
    L.snp_phase.argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    cfg = L.config_init(fa.encode(), sr.encode(), lr.encode())
    for n in sorted(g["snp_phase"]):
        r = L.snp_phase(n.encode(), cfg)
        assert digest(C.string_at(r.contents.contig).decode()) == g["snp_phase"][n], n
        L.polishresult_destory(r)
    L.config_destory(cfg)
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    for batch_bp in ("16000000", "55000"):      # both contigs in one batch; one batch per contig
        p = subprocess.run([exe, "snpphase", fa, sr, lr], capture_output=True, text=True, env=dict(os.environ, NP1_BATCH_BP=batch_bp))
        assert p.returncode == 0, p.stderr
        assert {n: digest(x) for n, x in parse_cli_fasta(p.stdout).items()} == g["snp_phase"]
    out = str(tmp_path / "o.fa")
    for extra in ([], ["--batch_bp", "55000"], ["-debug"]):      # batched (one / two batches) and contig by contig through the drop-in symbol
        if os.path.exists(out):
            os.remove(out)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py"), "-g", fa, "-t", "3", "-s", sr, "-l", lr, "-o", out] + extra,
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        recs = open(out).read().strip().split("\n")
        got = {recs[k].split()[0][1:]: recs[k + 1] for k in range(0, len(recs), 2)}
        assert {n.rsplit("_np", 1)[0]: digest(x) for n, x in got.items()} == {n.rsplit("_np", 1)[0]: d for n, d in g["snp_phase"].items()}, extra
