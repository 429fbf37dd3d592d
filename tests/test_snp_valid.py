"""Task 4 (snp_valid, reference: source/lib/snpvalid.c:3-66 on top of kmercount.c): oracle restatement against goldens the
compiled reference produced and against the compiled reference itself; on the GPU the product (np1_batch_snp_valid, the drop-in
`snp_valid` symbol, the CLI and the Python caller) against the oracle and the same goldens.

Inputs for which the reference has no defined result (a second-round region reaching outside the insertion columns of its k-mer
region dereferences a null list there, kmercount.c:398,431) make the oracle return None and the product fail loudly."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import pytest

import oracle_binding as ob
from conftest import ref_binary, run_ref
from nextpolish_amd import _native as nat

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REAL = os.path.join(HERE, "golden", "real")
GOLD = json.load(open(os.path.join(HERE, "golden", "snpvalid_golden.json")))


def digest(s):
    return {"len": len(s), "md5": hashlib.md5(s.encode()).hexdigest()}


def synth(params):
    kw = dict(params)
    return nat.Stream.synth(kw.pop("lens"), **kw)


def adversarial(seed):
    """small contigs with a lot of lower case: odd second-round lists, leftover ends, inverted pairs"""
    lens = [300 + 37 * (seed % 11), 150 + seed % 90, 900]
    return nat.Stream.synth(lens, depth=[6, 25, 60, 3, 12, 1.5][seed % 6], seed=9500 + seed, with_qual=1, weird_rate=0.05, softclip_rate=0.1,
                            draft_lower=[0.05, 0.15, 0.4, 0.7][seed % 4], read_indel=[0.001, 0.01, 0.03][seed % 3])


# ------------------------------------------------------------------------------------------------------------ CPU


@pytest.mark.parametrize("k", range(len(GOLD["synth"])))
def test_oracle_matches_reference_goldens_synth(k):
    g = GOLD["synth"][k]
    st = synth(g["params"])
    cfg = ob.default_config(read_tlen=g["read_tlen"], read_len=g["read_len"])
    for i, exp in enumerate(g["snp_valid"]):
        assert digest(ob.snp_valid(st, i, cfg)) == exp, "synth %d contig %d" % (k, i)


@pytest.mark.parametrize("tag", sorted(GOLD["real"]))
def test_oracle_matches_reference_goldens_real_alignments(tag):
    g = GOLD["real"][tag]
    st = nat.Stream.load(os.path.join(REAL, g["fasta"]), os.path.join(REAL, g["bam"]), with_qual=True)
    cfgp = nat.lib().config_init(os.path.join(REAL, g["fasta"]).encode(), os.path.join(REAL, g["bam"]).encode(), None)
    cfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
    nat.lib().config_destory(cfgp)
    for i, n in enumerate(st.names):
        assert digest(ob.snp_valid(st, i, cfg)) == g["snp_valid"][n], "%s %s" % (tag, n)


needs_ref = pytest.mark.skipif(ref_binary() is None, reason="oracle/_ref/nextpolish1 not built (needs /root/reference)")


@needs_ref
def test_goldens_still_match_compiled_reference():
    for tag, g in GOLD["real"].items():
        got = run_ref("snpvalid", os.path.join(REAL, g["fasta"]), os.path.join(REAL, g["bam"]))
        assert {n: digest(s) for n, s in got.items()} == g["snp_valid"]


@needs_ref
def test_oracle_vs_reference_fuzz(tmp_path):
    """Adversarial drafts (up to 70 % lower case): the reference crashes on a few of them; where it answers, the oracle answers
    the same, and the oracle reports "undefined" only where the reference's answer rests on a null dereference that happened to
    survive."""
    fa, bam = str(tmp_path / "s.fa"), str(tmp_path / "s.bam")
    compared = 0
    for seed in range(60):
        st = adversarial(seed)
        st.write_files(fa, bam)
        try:
            ref = run_ref("snpvalid", fa, bam)
        except subprocess.CalledProcessError:
            continue
        cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
        cfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
        nat.lib().config_destory(cfgp)
        for i, n in enumerate(st.names):
            got = ob.snp_valid(st, i, cfg)
            if got is None:
                continue
            assert got == ref[n], "seed %d contig %s" % (seed, n)
            compared += 1
    assert compared > 120


def test_host_model_of_the_kernel_bodies_matches_oracle():
    """np1_kmer.h bodies (kc_fts_split, the votes that leave the marks alone, the undefined-upstream check) driven on the host the
    way the kernels drive them (tests/model): goldens' inputs and adversarial drafts."""
    import model_binding as mb
    cases = [(synth(g["params"]), g["read_tlen"], g["read_len"]) for g in GOLD["synth"]] + [(adversarial(s), 1500, 150) for s in range(40)]
    n_defined = 0
    for st, tlen, rlen in cases:
        cfg = nat.default_config()
        cfg.read_tlen, cfg.read_len = tlen, rlen
        ocfg = ob.default_config(read_tlen=tlen, read_len=rlen)
        want = [ob.snp_valid(st, i, ocfg) for i in range(st.n_contigs)]
        if any(w is None for w in want):
            with pytest.raises(ValueError):
                mb.snp_valid(st, cfg)
            continue
        assert mb.snp_valid(st, cfg) == want
        n_defined += 1
    assert n_defined > 20


# ------------------------------------------------------------------------------------------------------------ GPU


@pytest.fixture(scope="module")
def ctx():
    from nextpolish_amd import device
    c = device.Context(0)
    yield c
    c.close()


def _check(ctx, st, read_tlen=1500, read_len=150):
    """product == oracle per contig; a batch holding a contig the reference has no result for must fail loudly"""
    cfg = nat.default_config()
    cfg.read_tlen, cfg.read_len = read_tlen, read_len
    ocfg = ob.default_config(read_tlen=read_tlen, read_len=read_len)
    want = [ob.snp_valid(st, i, ocfg) for i in range(st.n_contigs)]
    b = ctx.upload(st)
    try:
        if any(w is None for w in want):
            with pytest.raises(RuntimeError, match="no defined result"):
                b.snp_valid(cfg)
            return False
        b.snp_valid(cfg)
        got = b.results()
    finally:
        b.close()
    for i in range(st.n_contigs):
        assert len(got[i]) == len(want[i]), "contig %d: length %d != %d" % (i, len(got[i]), len(want[i]))
        if got[i] != want[i]:
            k = next(j for j in range(len(want[i])) if got[i][j] != want[i][j])
            raise AssertionError("snp_valid contig %d differs at %d: %r vs %r" % (i, k, got[i][max(0, k - 8):k + 8], want[i][max(0, k - 8):k + 8]))
    return True


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(GOLD["synth"])))
def test_gpu_matches_oracle_and_goldens_synth(ctx, k):
    g = GOLD["synth"][k]
    st = synth(g["params"])
    assert _check(ctx, st, g["read_tlen"], g["read_len"])
    cfg = nat.default_config()
    cfg.read_tlen, cfg.read_len = g["read_tlen"], g["read_len"]
    b = ctx.upload(st)
    b.snp_valid(cfg)
    assert [digest(s) for s in b.results()] == g["snp_valid"]
    b.close()


@pytest.mark.gpu
def test_gpu_matches_oracle_on_adversarial_drafts(ctx):
    n_defined = 0
    for seed in range(80):
        n_defined += 1 if _check(ctx, adversarial(seed)) else 0
    assert n_defined > 30


@pytest.mark.gpu
def test_gpu_many_contigs_in_one_batch(ctx):
    st = nat.Stream.synth([2500 + 97 * k for k in range(40)], depth=30, seed=4141, with_qual=1, draft_lower=0.03, read_indel=0.002, softclip_rate=0.05)
    assert _check(ctx, st)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(GOLD["real"]))
def test_gpu_dropin_symbol_cli_and_caller_on_real_alignments(tag, tmp_path):
    """snp_valid(tigname, cfg) like source/lib/nextpolish1.py:97-98,220; `nextpolish1 snpvalid fa bam` like main.c:7-8; the Python
    caller with -t 4."""
    g = GOLD["real"][tag]
    fa, bam = os.path.join(REAL, g["fasta"]), os.path.join(REAL, g["bam"])
    L = nat.lib()
    L.snp_valid.restype = C.POINTER(nat.PolishResult)
    L.snp_valid.argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    cfg = L.config_init(fa.encode(), bam.encode(), None)
    for n in sorted(g["snp_valid"]):
        r = L.snp_valid(n.encode(), cfg)
        assert digest(C.string_at(r.contents.contig).decode()) == g["snp_valid"][n], "%s %s" % (tag, n)
        L.polishresult_destory(r)
    L.config_destory(cfg)
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    p = subprocess.run([exe, "snpvalid", fa, bam], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().split("\n")
    got = {lines[k][1:].rsplit("_", 1)[0]: lines[k + 1] for k in range(0, len(lines), 2)}
    assert {n: digest(s) for n, s in got.items()} == g["snp_valid"]
    out = str(tmp_path / "o.fa")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py"), "-g", fa, "-t", "4", "-s", bam, "-o", out],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    recs = open(out).read().strip().split("\n")
    got = {recs[k].split()[0][1:]: recs[k + 1] for k in range(0, len(recs), 2)}
    assert {n.rsplit("_np", 1)[0]: digest(s) for n, s in got.items()} == {n.rsplit("_np", 1)[0]: d for n, d in g["snp_valid"].items()}
