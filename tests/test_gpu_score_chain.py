"""GPU parity tests proper: the HIP score_chain path (through the C ABI) against the CPU oracle,
bit for bit (sequence, length, lowercase mask), on seeded synthetic workloads and on micro-cases
built to hit every quirk of the reference walk (reference: source/lib/contig.c:202-496)."""
import ctypes as C
import os
import subprocess

import pytest

from nextpolish_amd import _native as nat
import oracle_binding as ob
from conftest import ROOT, parse_cli_fasta, ref_binary, run_ref
from fuzzgen import random_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from nextpolish_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _check(ctx, st, cfg=None, ocfg=None):
    got = ctx.score_chain(st, cfg)
    for i in range(st.n_contigs):
        want = ob.score_chain(st, i, ocfg)
        assert len(got[i]) == len(want), "contig %d: length %d != %d" % (i, len(got[i]), len(want))
        if got[i] != want:
            k = next(j for j in range(len(want)) if got[i][j] != want[j])
            raise AssertionError("contig %d differs at %d: %r vs %r" % (i, k, got[i][max(0, k - 8):k + 8], want[max(0, k - 8):k + 8]))
    return got


@pytest.mark.parametrize("seed", range(12))
def test_synth_matches_oracle(ctx, seed):
    kw = dict(depth=[5, 15, 30, 60, 120][seed % 5], seed=1000 + seed, weird_rate=0.02 if seed % 2 else 0.0,
              draft_lower=0.01 if seed % 3 == 0 else 0.0, read_indel=0.002 if seed % 4 == 0 else 0.0001,
              softclip_rate=0.05, draft_indel=0.02 if seed % 7 == 0 else 0.005)
    st = nat.Stream.synth([3000 + seed * 137, 900 + seed * 11, 200], **kw)
    _check(ctx, st)


def test_micro_cases_match_oracle(ctx):
    """400 random micro-workloads: leading insertions at position 0, insertion columns after base 0, hard
    clips on kept records, N/=/X/P ops, homopolymer reads, odd draft letters, filtered flags."""
    for seed in range(400):
        contigs, reads = random_case(seed)
        st = nat.Stream.from_reads(contigs, reads)
        _check(ctx, st)


def test_empty_and_ragged(ctx):
    # contigs without any record, single-base contig, records only on the last contig
    contigs = [("a", "ACGTNNacgt"), ("b", "A"), ("c", "ACGTACGTACGTAAAACCCCGGGGTTTT")]
    reads = [dict(ctg=2, pos=2, cigar=[("M", 20)], seq="GTACGTACGTAAAACCCCGG"),
             dict(ctg=2, pos=3, cigar=[("M", 10), ("I", 2), ("M", 8)], seq="TACGTACGTATTAAACCCCG")]
    st = nat.Stream.from_reads(contigs, reads)
    got = _check(ctx, st)
    assert got[1] == "a"   # a lone draft base: depth-1 vote -> flagged lowercase


def test_parameters(ctx):
    st = nat.Stream.synth([20000], depth=40, seed=77, softclip_rate=0.05)
    for rate, ratio, trim in [(0.5, 0.8, 2), (0.25, 0.8, 2), (1.0, 0.5, 0), (0.75, 1.2, 5), (0.0, 0.95, 1)]:
        cfg = nat.default_config()
        cfg.indel_balance_factor_sgs, cfg.min_count_ratio_skip, cfg.trim_len_edge = rate, ratio, trim
        ocfg = ob.default_config(indel_balance_factor_sgs=rate, min_count_ratio_skip=ratio, trim_len_edge=trim)
        _check(ctx, st, cfg, ocfg)


def test_general_indel_balance_factor_takes_the_sequential_fp64_path(ctx):
    """-indel_balance_factor_sgs that is not R / 2^K (the reference takes any double, contig.c:448): scores are the
    reference's doubles in its order, one sequential run per contig; checked against the oracle, which is pinned to the compiled
    reference for the same rates (tests/test_oracle.py).  Noisy, deep and multi-contig inputs, incl. the real bwa records."""
    import hashlib, json
    sts = [nat.Stream.synth([20000, 3000], depth=40, seed=77, softclip_rate=0.05),
           nat.Stream.synth([6000], depth=200, seed=5, read_sub=0.06, read_indel=0.01),
           nat.Stream.load(os.path.join(ROOT, "tests", "golden", "real", "g.fa"), os.path.join(ROOT, "tests", "golden", "real", "sgs.s30.bam"))]
    for rate, ratio in [(0.3, 0.8), (0.55, 0.8), (0.33, 1.1), (1.0 / 3.0, 0.6)]:
        for st in sts:
            cfg = nat.default_config()
            cfg.indel_balance_factor_sgs, cfg.min_count_ratio_skip = rate, ratio
            _check(ctx, st, cfg, ob.default_config(indel_balance_factor_sgs=rate, min_count_ratio_skip=ratio))
    # kmer_count's no-depth regions use the same DP (np1_kmer.h:kc_region_solve): low depth makes many of them
    for seed in range(3):
        kst = nat.Stream.synth([4000 + seed * 97, 700], depth=[4, 6, 10][seed], seed=3100 + seed, with_qual=1, draft_lower=0.03,
                               read_indel=0.002, softclip_rate=0.05, lowmapq_rate=0.2)
        for rate in (0.3, 0.55):
            cfg = nat.default_config()
            cfg.read_tlen, cfg.indel_balance_factor_sgs = 1500, rate
            b = ctx.upload(kst)
            b.kmer_count(cfg)
            assert b.results() == [ob.kmer_count(kst, i, ob.default_config(read_tlen=1500, indel_balance_factor_sgs=rate)) for i in range(kst.n_contigs)]
            b.close()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "real", "real_golden.json")))["sr"]["sgs.s30"]["rates"]
    for rate, exp in gold.items():      # the compiled reference's own output for these rates, committed
        cfg = nat.default_config()
        cfg.indel_balance_factor_sgs = float(rate)
        got = ctx.score_chain(sts[2], cfg)
        assert {n: hashlib.md5(s.encode()).hexdigest() for n, s in zip(sts[2].names, got)} == exp, rate


def test_crowded_slots_escalate(ctx):
    """Very noisy reads: more than 16 distinct contexts per slot forces the larger k_vote instantiations."""
    st = nat.Stream.synth([4000], depth=300, seed=5, read_sub=0.08, read_indel=0.01)
    _check(ctx, st)


def test_one_megabase(ctx):
    st = nat.Stream.synth([700000, 300000], depth=50, seed=20250118)
    _check(ctx, st)


def test_repeatable(ctx):
    st = nat.Stream.synth([50000], depth=30, seed=3)
    b = ctx.upload(st)
    b.score_chain()
    a = b.results()
    for _ in range(3):
        b.score_chain()
        assert b.results() == a
    b.close()


def test_dropin_abi_and_cli(tmp_path):
    """config_init / score_chain / polishresult_destory exactly as the reference's ctypes caller uses them
    (reference: source/lib/nextpolish1.py:181-189,219), plus the CLI, against the oracle and -- when the
    compiled reference travelled with the repo -- against the reference binary itself."""
    st = nat.Stream.synth([30000, 8000], depth=30, seed=11, with_qual=1)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam)
    L = nat.lib()
    cfg = L.config_init(fa.encode(), bam.encode(), None)
    assert cfg.contents.read_len == 150 and cfg.contents.read_tlen > 1000
    cfg.contents.trace_polish_open = 1
    want, want_pts = [], []
    ocfg = ob.default_config(trace_polish_open=1)
    for i in range(st.n_contigs):
        want.append(ob.score_chain(st, i, ocfg))
        want_pts.append(ob.last_points())      # the oracle's change list (np1_oracle.c get_contig == the compiled reference: tests/test_points.py)
    for i, name in enumerate(st.names):
        r = L.score_chain(name.encode(), cfg)
        seq = C.string_at(r.contents.contig).decode()
        assert r.contents.length == len(seq)
        assert seq == want[i]
        pts = [[r.contents.data[k].pos, r.contents.data[k].index, r.contents.data[k].curbase.decode(), r.contents.data[k].base.decode()]
               for k in range(r.contents.datalength)]
        assert len(pts) > 0 and pts == want_pts[i], "task-1 change list of %s" % name      # (the exact list, not just its length)
        L.polishresult_destory(r)
    L.config_destory(cfg)
    out = subprocess.run([os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1"), "scorechain", fa, bam],
                         stdout=subprocess.PIPE, check=True).stdout.decode()
    cli = parse_cli_fasta(out)
    assert [cli[n] for n in st.names] == want
    assert out.startswith(">%s_1\n" % st.names[0])
    if ref_binary():
        ref = run_ref("scorechain", fa, bam)
        assert [ref[n] for n in st.names] == want


def test_python_harness_end_to_end(tmp_path):
    """nextpolish_amd/nextpolish1.py (mirror of the reference's lib/nextpolish1.py): batched GPU path, block
    file selection, resume after a truncated output, -u, and the -debug per-contig path with its trace lines."""
    import sys
    st = nat.Stream.synth([12000, 5000, 3000, 800], depth=30, seed=31, with_qual=1)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam)
    want = {n: ob.score_chain(st, i) for i, n in enumerate(st.names)}
    exe = [sys.executable, os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py")]

    def parse(text):
        recs, name = {}, None
        for line in text.splitlines():
            if line.startswith(">"):
                name, ln = line[1:].split()
                recs[name] = [int(ln), ""]
            else:
                recs[name][1] += line
        return recs

    out = subprocess.run(exe + ["-g", fa, "-s", bam, "-t", "1", "--batch_bp", "9000"], stdout=subprocess.PIPE, check=True).stdout.decode()
    recs = parse(out)
    assert list(recs) == [n + "_np1" for n in st.names]
    for n in st.names:
        assert recs[n + "_np1"] == [len(want[n]), want[n]]
    # block file + resume: contig 0 finished, contig 2 cut off mid-record, contig 1 belongs to another block
    blc = tmp_path / "g.blc"
    blc.write_text("".join("%s %d\n" % (n, 1 if k == 1 else 0) for k, n in enumerate(st.names)))
    part = tmp_path / "part0.fa"
    part.write_text(">%s_np1 %d\n%s\n>%s_np1 %d\n%s" % (st.names[0], len(want[st.names[0]]), want[st.names[0]],
                                                       st.names[2], len(want[st.names[2]]), want[st.names[2]][:100]))
    subprocess.run(exe + ["-g", fa, "-s", bam, "-t", "1", "-b", str(blc), "-i", "0", "-u", "-o", str(part)], check=True)
    recs = parse(part.read_text())
    assert list(recs) == [st.names[k] + "_np1" for k in (0, 2, 3)]
    assert recs[st.names[2] + "_np1"][1] == want[st.names[2]].upper()
    # -debug: per-contig drop-in path, trace lines "name pos index curbase base" on stderr
    r = subprocess.run(exe + ["-g", fa, "-s", bam, "-t", "1", "-debug", "-b", str(blc), "-i", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, check=True)
    recs = parse(r.stdout.decode())
    assert recs == {st.names[1] + "_np1": [len(want[st.names[1]]), want[st.names[1]]]}
    trace = [l.split() for l in r.stderr.decode().splitlines() if l.startswith(st.names[1] + " ")]
    assert trace and all(len(t) == 5 for t in trace)
    # replaying the trace over the draft reproduces the polished sequence (uppercased)
    draft = st.contig_draft(1).decode().upper()
    cols = {}
    for _, pos, idx, cur, _ in trace:
        cols.setdefault(int(pos), {})[int(idx)] = cur
    rebuilt = []
    for i, ch in enumerate(draft):
        c = cols.get(i, {})
        b = c.get(0, ch)
        if b != ".":
            rebuilt.append(b)
        for j in sorted(k for k in c if k > 0):
            rebuilt.append(c[j])
    assert "".join(rebuilt) == want[st.names[1]].upper()


# ---------------------------------------------------------------------------------------------------------------
# kmer_count (task 2) on the GPU against the oracle

def _check_kmer(ctx, st, read_tlen=1500):
    cfg = nat.default_config()
    cfg.read_tlen = read_tlen
    ocfg = ob.default_config(read_tlen=read_tlen)
    b = ctx.upload(st)
    b.kmer_count(cfg)
    got = b.results()
    b.close()
    for i in range(st.n_contigs):
        want = ob.kmer_count(st, i, ocfg)
        assert len(got[i]) == len(want), "contig %d: length %d != %d" % (i, len(got[i]), len(want))
        if got[i] != want:
            k = next(j for j in range(len(want)) if got[i][j] != want[j])
            raise AssertionError("kmer_count contig %d differs at %d: %r vs %r" % (i, k, got[i][max(0, k - 8):k + 8], want[max(0, k - 8):k + 8]))
    return got


@pytest.mark.parametrize("seed", range(10))
def test_kmer_count_synth_matches_oracle(ctx, seed):
    kw = dict(depth=[5, 15, 30, 60, 120][seed % 5], seed=2000 + seed, with_qual=1, weird_rate=0.02 if seed % 2 else 0.0,
              draft_lower=[0.01, 0.03, 0.002][seed % 3], read_indel=0.002 if seed % 4 == 0 else 0.0001, softclip_rate=0.05,
              draft_indel=0.02 if seed % 7 == 0 else 0.005)
    st = nat.Stream.synth([3000 + seed * 137, 900 + seed * 11, 200], **kw)
    _check_kmer(ctx, st)


def test_kmer_count_lowercase_micro_cases_match_oracle(ctx):
    """thin coverage, a tenth of the draft in lower case (tests/test_oracle.py lowercase_micro_case): the level-1 fallback of
    kmer_count incl. the part at a contig's end with no record behind it (seed 10019)"""
    from test_oracle import lowercase_micro_case
    for seed in range(10000, 10080):
        contigs, reads = lowercase_micro_case(seed)
        _check_kmer(ctx, nat.Stream.from_reads(contigs, reads), read_tlen=1000)


def test_kmer_count_micro_cases_match_oracle(ctx):
    import random
    for seed in range(300):
        contigs, reads = random_case(seed + 5000, max_len=300, max_reads=80)
        rng = random.Random(seed)
        c2 = []
        for n, d in contigs:
            d = list(d)
            for _ in range(rng.randint(0, 6)):
                i = rng.randrange(len(d))
                for j in range(i, min(len(d), i + rng.randint(1, 6))):
                    d[j] = d[j].lower()
            c2.append((n, "".join(d)))
        st = nat.Stream.from_reads(c2, reads)
        _check_kmer(ctx, st, read_tlen=1000)


def test_one_round_score_chain_then_kmer_count(ctx, tmp_path):
    """Task 1 then task 2 on its output, as the workflow chains them (reads re-placed on the new draft by shifting
    nothing: the second task simply runs on a draft with score_chain's lowercase marks).  Drop-in ABI + CLI."""
    st = nat.Stream.synth([40000, 9000], depth=40, seed=91, with_qual=1, draft_lower=0.004)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam)
    L = nat.lib()
    cfg = L.config_init(fa.encode(), bam.encode(), None)
    ocfg = ob.default_config(read_tlen=cfg.contents.read_tlen, read_len=cfg.contents.read_len)
    w = ob.from_files("kmer_count", fa, bam, ocfg)      # files: the oracle replays the region iterator like the product does
    want = [w[n] for n in st.names]
    for i, name in enumerate(st.names):
        r = L.kmer_count(name.encode(), cfg)
        assert C.string_at(r.contents.contig).decode() == want[i]
        L.polishresult_destory(r)
    L.config_destory(cfg)
    out = subprocess.run([os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1"), "kmercount", fa, bam],
                         stdout=subprocess.PIPE, check=True).stdout.decode()
    cli = parse_cli_fasta(out)
    assert [cli[n] for n in st.names] == want
    if ref_binary():
        ref = run_ref("kmercount", fa, bam)
        assert [ref[n] for n in st.names] == want


def test_full_size_config2_matches_oracle(ctx):
    """BASELINE config 2 at full size (5 Mb draft in 3 contigs, 50x PE150, 1.67 M records: the bench.py workload): every
    contig bit-identical to the CPU oracle, and the pass is repeatable."""
    st = nat.Stream.synth([2500000, 1500000, 1000000], depth=50.0, seed=20250119)
    got = _check(ctx, st)
    assert ctx.score_chain(st) == got
    st.close()


def test_full_size_kmer_count_matches_oracle(ctx):
    """kmer_count at the bench shape (5 Mb draft in 3 contigs, 50x PE150 with qualities, 0.4 % of the draft flagged): ~10 000
    flagged positions per contig, so the run-parallel region discovery, its sequential cursor pass and the scratch
    layout are exercised at scale."""
    st = nat.Stream.synth([2500000, 1500000, 1000000], depth=50.0, seed=20250120, with_qual=1, draft_lower=0.004)
    _check_kmer(ctx, st)
    st.close()


def test_streamed_pipe_matches_oracle_and_direct_path(tmp_path):
    """np1_pipe_*: batches on two device lanes with reused HBM buffers and pinned host arrays (sizes shrink and grow between
    batches), from memory and from files (loader threads + in-order sink), score_chain and kmer_count."""
    from nextpolish_amd.device import Pipe
    sts = [nat.Stream.synth(lens, depth=d, seed=900 + k, with_qual=1, draft_lower=0.01, prefix="b%dc" % k)
           for k, (lens, d) in enumerate([([40000, 9000], 30), ([3000], 60), ([120000, 500, 70000], 25), ([15000], 8), ([60000, 60000], 40)])]
    for st in sts:
        st.pin()
    pipe = Pipe(0, lanes=2)
    for rep in range(2):      # second pass: every buffer is reused
        got = pipe.run(sts)
        for k, st in enumerate(sts):
            assert got[k] == [ob.score_chain(st, i) for i in range(st.n_contigs)], "batch %d pass %d" % (k, rep)
    cfg = nat.default_config()
    cfg.read_tlen = 1500
    got = pipe.run(sts, cfg=cfg, task=2)
    for k, st in enumerate(sts):
        assert got[k] == [ob.kmer_count(st, i, ob.default_config(read_tlen=1500)) for i in range(st.n_contigs)], "kmer_count batch %d" % k
    # from files: five contigs, batches of at most 50 kb -> three batches, two loaders; sink order = FASTA index order
    big = nat.Stream.synth([40000, 9000, 30000, 45000, 2000], depth=30, seed=77, with_qual=1, draft_lower=0.01)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    big.write_files(fa, bam)
    out = pipe.run_files(fa, bam, batch_bp=50000)
    assert [n for n, _ in out] == big.names
    assert [s for _, s in out] == [ob.score_chain(big, i) for i in range(big.n_contigs)]
    out = pipe.run_files(fa, bam, names=[big.names[3], big.names[1]], batch_bp=50000, cfg=cfg, task=2)
    assert [n for n, _ in out] == [big.names[3], big.names[1]]
    w = ob.from_files("kmer_count", fa, bam, ob.default_config(read_tlen=1500))
    assert [s for _, s in out] == [w[big.names[i]] for i in (3, 1)]
    pipe.close()


def test_records_beyond_16_bit_counts_take_the_staged_sequence(ctx):
    """more than 65 535 CIGAR operations (a CG-tag CIGAR) and more than 65 535 bases in one record: the reference walks any record
    (contig.c:247-331); the descriptors index the query with 16 bits, so the batch re-runs on symbol rows -- and a second, ordinary
    batch on the same batch object is back on the descriptors"""
    from fuzzgen import long_record_case
    contigs, reads = long_record_case(3)
    st = nat.Stream.from_reads(contigs, reads)
    assert int(st.n_cigar.max()) > 65535 and int(st.l_qseq.max()) > 65535
    _check(ctx, st)
    _check(ctx, nat.Stream.synth([30000], depth=30, seed=9))


def test_more_than_160_contexts_in_a_slot(ctx):
    """base.c:60-71: the reference's context list of a slot grows without bound; the fourth vote level keeps every possible context
    (3 symbols of 4 bits) in an HBM list"""
    from fuzzgen import crowded_context_case
    contigs, reads = crowded_context_case(11)
    _check(ctx, nat.Stream.from_reads(contigs, reads))
    contigs, reads = crowded_context_case(12, depth=3000, L=2000)
    _check(ctx, nat.Stream.from_reads(contigs, reads))


def test_staged_sequence_matches_oracle():
    """NP1_PIPELINE=staged: symbol rows in HBM + k_vote, the sequence batches with over-long records or over-crowded slots fall back to"""
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from nextpolish_amd import _native as nat\nfrom nextpolish_amd.device import Context\nimport oracle_binding as ob\n"
            "from fuzzgen import random_case\n"
            "c = Context(0)\n"
            "sts = [nat.Stream.synth([20000, 3000, 700], depth=d, seed=40 + d, weird_rate=0.02, read_indel=0.002, softclip_rate=0.05) for d in (15, 60)]\n"
            "sts += [nat.Stream.from_reads(*random_case(s)) for s in range(40)]\n"
            "sts.append(nat.Stream.synth([4000], depth=300, seed=5, read_sub=0.08, read_indel=0.01))\n"
            "for st in sts:\n"
            "    got = c.score_chain(st)\n"
            "    for i in range(st.n_contigs):\n"
            "        assert got[i] == ob.score_chain(st, i)\n"
            "print('staged ok', len(sts))\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, NP1_PIPELINE="staged")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "staged ok" in r.stdout, r.stdout + r.stderr


def test_four_slots_per_lane_kernel_matches_oracle():
    """NP1_TILE=9: k_tile9 (np1_tile9.h) instead of k_tile3 -- agreeing records counted per six-slot window, every other (record, lane)
    pair deferred, evaluated and tallied in record order; waves that run out of list room hand their chunks to k_tile3.  The switch is
    read once per process, hence the subprocess.  Workloads: the synthetic shapes at 8-120x (deep ones overflow the lists and take
    the hand-back path), the micro-case fuzz (odd CIGARs, chained descriptors, contig edges), crowded slots, a megabase."""
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from nextpolish_amd import _native as nat\nfrom nextpolish_amd.device import Context\nimport oracle_binding as ob\n"
            "from fuzzgen import random_case\n"
            "c = Context(0)\n"
            "sts = [nat.Stream.synth([30000 + 977 * d, 3000, 700], depth=d, seed=90 + d, weird_rate=0.02 if d %% 2 else 0.0, read_indel=0.002, softclip_rate=0.05, draft_lower=0.02) for d in (8, 15, 30, 60, 120)]\n"
            "sts += [nat.Stream.from_reads(*random_case(s)) for s in range(120)]\n"
            "sts.append(nat.Stream.synth([4000], depth=300, seed=5, read_sub=0.08, read_indel=0.01))\n"
            "sts.append(nat.Stream.synth([1000000, 250000], depth=30, seed=123))\n"
            "for st in sts:\n"
            "    got = c.score_chain(st)\n"
            "    for i in range(st.n_contigs):\n"
            "        assert got[i] == ob.score_chain(st, i)\n"
            "print('tile9 ok', len(sts))\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, NP1_TILE="9")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "tile9 ok" in r.stdout, r.stdout + r.stderr


def test_kmer_count_with_records_beyond_16_bit_counts(ctx, tmp_path):
    """tasks 2 / 4 walk the records themselves (np1_kmer.h), no descriptors: a CG-tag CIGAR only needs the 32-bit operation count --
    in memory, and from the files (device ingest meets the placeholder CIGAR and hands the file to the host loader)"""
    from fuzzgen import long_record_case
    contigs, reads = long_record_case(7, n_ops=66000, plain_bases=66000)
    st = nat.Stream.from_reads(contigs, reads)
    _check_kmer(ctx, st)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam)
    cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
    ocfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
    nat.lib().config_destory(cfgp)
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    for cmd, task in (("kmercount", "kmer_count"), ("snpvalid", "snp_valid")):
        out = subprocess.run([exe, cmd, fa, bam], stdout=subprocess.PIPE, timeout=600, check=True).stdout.decode()
        assert parse_cli_fasta(out) == ob.from_files(task, fa, bam, ocfg), cmd
    out = subprocess.run([exe, "scorechain", fa, bam], stdout=subprocess.PIPE, timeout=600, check=True).stdout.decode()
    st2 = nat.Stream.load(fa, bam)
    assert parse_cli_fasta(out)["long"] == ob.score_chain(st2, 0)


def test_two_insertions_at_one_position_are_refused_by_name(ctx):
    """the reference dies on these (tests/test_oracle.py::test_reference_has_no_result_for_two_insertions_at_one_position): no result to match"""
    contigs = [("t", "ACGTTGCAAGGCTTAACCGGTTACGATCGATTGCA" * 3)]
    d = contigs[0][1]
    reads = [dict(ctg=0, pos=5, cigar=[("M", 20), ("I", 2), ("P", 1), ("I", 1), ("M", 30)], seq=d[5:25] + "GG" + "T" + d[25:55])]
    reads += [dict(ctg=0, pos=p, cigar=[("M", 60)], seq=d[p:p + 60]) for p in (0, 3, 8, 20)]
    reads.sort(key=lambda r: r["pos"])
    with pytest.raises(RuntimeError, match="two insertion"):
        ctx.score_chain(nat.Stream.from_reads(contigs, reads))
