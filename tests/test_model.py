"""Host lockstep model of the two HIP launch sequences (tests/model) against the oracle: checks the staged
algorithm itself -- slot space, symbol rows / record descriptors incl. chained parts, single-state shortcut,
run-wise exact integer DP, emission -- on the CPU, with the same per-lane bodies the kernels compile."""
import pytest

from nextpolish_amd import _native as nat
import oracle_binding as ob
import model_binding as mb
from fuzzgen import random_case


@pytest.mark.parametrize("fused", [0, 1, 2])   # staged rows, descriptors (k_tile3), four slots per lane with deferred entries (k_tile9)
def test_model_micro_cases(fused):
    parts = 0
    for seed in range(250):
        contigs, reads = random_case(seed)
        st = nat.Stream.from_reads(contigs, reads)
        got, stats = mb.score_chain(st, fused=fused, want_stats=True)
        parts += stats["escalations"] if fused == 1 else 0
        for i in range(st.n_contigs):
            assert got[i] == ob.score_chain(st, i), "seed %d contig %d fused=%s" % (seed, i, fused)
    if fused == 1:
        assert parts > 100   # the chained-descriptor path is really exercised


@pytest.mark.parametrize("fused", [0, 1, 2])
@pytest.mark.parametrize("seed", range(5))
def test_model_synth(fused, seed):
    st = nat.Stream.synth([3000 + seed * 137, 900 + seed * 11, 200], depth=[5, 15, 30, 60, 120][seed % 5], seed=1000 + seed,
                          weird_rate=0.02 if seed % 2 else 0.0, draft_lower=0.01 if seed % 3 == 0 else 0.0,
                          read_indel=0.002 if seed % 4 == 0 else 0.0001, softclip_rate=0.05)
    got = mb.score_chain(st, fused=fused)
    for i in range(st.n_contigs):
        assert got[i] == ob.score_chain(st, i)


def test_model_parameters():
    st = nat.Stream.synth([6000], depth=40, seed=77, softclip_rate=0.05)
    for rate, ratio, trim in [(0.25, 0.8, 2), (1.0, 0.5, 0), (0.75, 1.2, 5), (0.0, 0.95, 1)]:
        cfg = nat.default_config()
        cfg.indel_balance_factor_sgs, cfg.min_count_ratio_skip, cfg.trim_len_edge = rate, ratio, trim
        ocfg = ob.default_config(indel_balance_factor_sgs=rate, min_count_ratio_skip=ratio, trim_len_edge=trim)
        for fused in (0, 1, 2):
            assert mb.score_chain(st, cfg, fused=fused)[0] == ob.score_chain(st, 0, ocfg)


def test_model_crowded_slots_escalate():
    st = nat.Stream.synth([1500], depth=250, seed=5, read_sub=0.08, read_indel=0.01)
    got, stats = mb.score_chain(st, want_stats=True)
    assert stats["escalations"] > 0
    assert got[0] == ob.score_chain(st, 0)


@pytest.mark.parametrize("fused", [0, 1])
def test_model_records_beyond_16_bit_counts(fused):
    """a CIGAR of more than 65 535 operations and a match of more than 65 535 bases (contig.c:247-331 walks any record): the
    descriptor sequence hands the batch to the staged one"""
    from fuzzgen import long_record_case
    contigs, reads = long_record_case(3)
    st = nat.Stream.from_reads(contigs, reads)
    assert int(st.n_cigar.max()) > 65535 and int(st.l_qseq.max()) > 65535
    got, stats = mb.score_chain(st, fused=fused, want_stats=True)
    assert stats["restarts"] == (1 if fused else 0)
    assert got[0] == ob.score_chain(st, 0)


@pytest.mark.parametrize("fused", [0, 1])
def test_model_more_than_160_contexts_in_a_slot(fused):
    """base.c:60-71 grows a slot's context list without bound; here the fourth level keeps one entry per possible context"""
    from fuzzgen import crowded_context_case
    contigs, reads = crowded_context_case(11)
    st = nat.Stream.from_reads(contigs, reads)
    got, stats = mb.score_chain(st, fused=fused, want_stats=True)
    assert stats["deep_chunks"] > 0 and stats["restarts"] == (1 if fused else 0)
    assert got[0] == ob.score_chain(st, 0)


@pytest.mark.parametrize("seed", range(6))
def test_model_intra_contig_tiling_joins_to_the_untiled_result(seed):
    """DESIGN.md 8 / SURVEY.md 8e: one contig polished as independent tiles (one per GPU when a single contig is all there is) ==
    the untiled oracle.  No state travels between tiles: a tile computes its stretch plus a halo from the records overlapping it and
    the chain restarts behind any single-state slot, which every halo must hold (else the tile is recomputed with a wider halo --
    the tiny halos here force that)."""
    st = nat.Stream.synth([6000 + 700 * seed, 1500], depth=[8, 30, 90][seed % 3], seed=300 + seed, weird_rate=0.02 if seed % 2 else 0.0,
                          read_indel=0.004, read_sub=0.02 if seed % 4 == 0 else 0.004, softclip_rate=0.05, draft_lower=0.01)
    want = [ob.score_chain(st, i) for i in range(st.n_contigs)]
    redo = 0
    for tile, halo in [(997, 64), (300, 8), (2048, 200), (150, 1), (5000, 150)]:
        got, ts = mb.score_chain_tiled(st, tile, halo, fused=seed % 2)
        assert got == want, (tile, halo)
        assert ts["tiles"] >= sum((int(n) + tile - 1) // tile for n in st.ctg_len)
        redo += ts["recomputed"]
    assert redo > 0


def test_model_intra_contig_tiling_micro_cases_and_crowded_runs():
    """odd CIGAR shapes at tile edges (tiles of a few bases), and a pileup noisy enough for multi-state runs longer than a read"""
    for seed in range(120):
        contigs, reads = random_case(seed)
        st = nat.Stream.from_reads(contigs, reads)
        want = [ob.score_chain(st, i) for i in range(st.n_contigs)]
        for tile, halo in [(7, 1), (23, 5), (64, 40)]:
            assert mb.score_chain_tiled(st, tile, halo)[0] == want, (seed, tile, halo)
    st = nat.Stream.synth([3000], depth=250, seed=5, read_sub=0.08, read_indel=0.01)
    got, ts = mb.score_chain_tiled(st, 500, 150)
    assert got[0] == ob.score_chain(st, 0)


def test_model_kmer_count_and_snp_valid_take_two_insertions_at_one_position():
    """'I P I' / 'I N I' records (tests/fuzzgen.py double_ins): score_chain has no result upstream (test_oracle.py pins the reference's
    crash) and refuses them by name, but kmercount.c / snpvalid.c only chain inside their regions and do have one -- the region walk
    of np1_kmer.h gives exactly that (oracle == compiled reference on such files: 198 of 200 live, the other two crash upstream)."""
    n = 0
    for seed in range(60):
        contigs, reads = random_case(seed, double_ins=True)
        st = nat.Stream.from_reads(contigs, reads)
        cfg = nat.default_config()
        cfg.read_tlen = 1500
        ocfg = ob.default_config(read_tlen=1500)
        assert mb.kmer_count(st, cfg) == [ob.kmer_count(st, i, ocfg) for i in range(st.n_contigs)], seed
        assert mb.snp_valid(st, cfg) == [ob.snp_valid(st, i, ocfg) for i in range(st.n_contigs)], seed
        n += 1
    assert n == 60


def test_upload_forms_round_trip():
    """2-bit bases + exception bytes, 4-bit draft, plain-record bits + one-byte position steps -- the product's builders
    (np1_upload.h) against host restatements of the kernels that undo them on the device: every array comes back as it was, on
    PE150-like streams (where the forms pay), on micro-cases full of odd letters, odd lengths, clips and empty contigs, and on
    records beyond the 16-bit counts"""
    from fuzzgen import long_record_case, crowded_context_case
    st = nat.Stream.synth([60000, 9000, 300], depth=30, seed=4)
    z = mb.upload_roundtrip(st)
    assert z["seq2_bytes"] * 100 < z["seq_bytes"] * 53 and z["compact_bytes"] * 3 < z["fields_bytes"] * 2 and z["plain"] * 10 > st.n_reads * 4 and z["full_positions"] == 3
    z = mb.upload_roundtrip(nat.Stream.synth([20000, 700], depth=40, seed=5, weird_rate=0.05, softclip_rate=0.2, read_indel=0.004, draft_lower=0.05))
    assert z["plain"] > 0
    for seed in range(150):
        mb.upload_roundtrip(nat.Stream.from_reads(*random_case(seed)))
    mb.upload_roundtrip(nat.Stream.from_reads(*long_record_case(3)))
    mb.upload_roundtrip(nat.Stream.from_reads(*crowded_context_case(11)))      # every nt16 code in the reads: all bytes are exceptions
    sr, lr = nat.Stream.synth_diploid([30000, 8000], seed=3, sr_holes=1)
    mb.upload_roundtrip(sr)
    mb.upload_roundtrip(lr)


# ---- kmer_count bodies (np1_kmer.h) against the oracle ----------------------------------------------------------
def _lowercase_some(contigs, seed):
    import random
    rng = random.Random(seed)
    out = []
    for n, d in contigs:
        d = list(d)
        for _ in range(rng.randint(0, 6)):
            i = rng.randrange(len(d))
            for j in range(i, min(len(d), i + rng.randint(1, 6))):
                d[j] = d[j].lower()
        out.append((n, "".join(d)))
    return out


def test_model_kmer_count_micro_cases():
    for seed in range(200):
        contigs, reads = random_case(seed + 5000, max_len=300, max_reads=80)
        st = nat.Stream.from_reads(_lowercase_some(contigs, seed), reads)
        cfg = nat.default_config()
        cfg.read_tlen = 1000
        got = mb.kmer_count(st, cfg)
        for i in range(st.n_contigs):
            assert got[i] == ob.kmer_count(st, i, ob.default_config(read_tlen=1000)), "seed %d contig %d" % (seed, i)


@pytest.mark.parametrize("seed", range(6))
def test_model_kmer_count_synth(seed):
    st = nat.Stream.synth([3000 + seed * 137, 900 + seed * 11, 200], depth=[5, 15, 30, 60, 120][seed % 5], seed=2000 + seed,
                          with_qual=1, weird_rate=0.02 if seed % 2 else 0.0, draft_lower=[0.01, 0.03, 0.002][seed % 3],
                          read_indel=0.002 if seed % 4 == 0 else 0.0001, softclip_rate=0.05)
    cfg = nat.default_config()
    cfg.read_tlen = 1500
    got = mb.kmer_count(st, cfg)
    for i in range(st.n_contigs):
        assert got[i] == ob.kmer_count(st, i, ob.default_config(read_tlen=1500))


def test_region_walk_run_parallel_form_equals_the_literal_walk():
    """k_kc_regions does not walk a contig base by base: it cuts the flagged positions into runs, takes each run's region
    on its own, replays the walk only where an extended region swallows the head of the next runs, and merges with the
    last region in registers (np1_kmer.h).  Its steps, executed one after the other on the CPU, must give what the literal
    walk gives -- on draft shapes chosen to make regions collide: dense flags, homopolymers, flagged runs next to each
    other, every gap / minimum-run / extension setting."""
    import ctypes as C
    import random
    import numpy as np
    L = mb.lib()
    L.np1m_regions.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int, C.c_int,
                               C.c_void_p, C.c_int32]
    L.np1m_regions.restype = C.c_int
    rng = random.Random(11)
    n_cases = 0
    for case in range(12000):
        n = rng.choice([1, 2, 5, 30, 200, 1500])
        hp = rng.choice([0.0, 0.3, 0.7, 0.95])            # chance that a base repeats its neighbour
        dens = rng.choice([0.002, 0.02, 0.1, 0.4, 0.9])   # flagged fraction
        clump = rng.choice([0.0, 0.5, 0.9])               # chance that a flagged base is followed by another one
        code = np.zeros(n, dtype=np.uint8)
        flag = np.zeros(n, dtype=np.uint8)
        for i in range(n):
            code[i] = code[i - 1] if i and rng.random() < hp else rng.choice([1, 2, 4, 8])
            f = rng.random() < (clump if i and flag[i - 1] & 1 else dens)
            flag[i] = 1 if f else 0                        # KC_FLAG_ZERO is bit 0 of the flag byte
        F = np.nonzero(flag)[0].astype(np.uint32)
        if F.size == 0:
            continue
        gap, con = rng.choice([(0, 0), (0, 2), (0, 5), (3, 0), (5, 0), (10, 1), (1, 1)])
        ext = rng.choice([0, 1, 2, 5])
        with_ext = rng.choice([0, 1])
        cap = 2 * int(F.size) + 8
        outs = []
        for mode in (0, 1):
            out = np.zeros(cap, dtype=np.int32)
            k = L.np1m_regions(code.ctypes.data, flag.ctypes.data, n, F.ctypes.data, int(F.size), gap, con, ext, with_ext, mode, out.ctypes.data, cap)
            outs.append((k, out[:max(k, 0)].tolist()))
        assert outs[0] == outs[1], (case, n, gap, con, ext, with_ext, F.tolist()[:40], outs)
        n_cases += 1
    assert n_cases > 6000


@pytest.mark.parametrize("rate", [0.3, 0.55])
def test_model_kmer_count_general_indel_balance_factor(rate):
    """kmer_count's no-depth fallback scores regions with the chain DP: for a rate that is no dyadic fraction the region DP
    keeps the reference's doubles (np1_kmer.h:kc_region_solve); low depth makes many such regions."""
    for seed in range(4):
        st = nat.Stream.synth([4000 + seed * 97, 700], depth=[4, 6, 10, 20][seed], seed=3100 + seed, with_qual=1, draft_lower=0.03,
                              read_indel=0.002, softclip_rate=0.05, lowmapq_rate=0.2)
        cfg = nat.default_config()
        cfg.read_tlen = 1500
        cfg.indel_balance_factor_sgs = rate
        got = mb.kmer_count(st, cfg)
        for i in range(st.n_contigs):
            assert got[i] == ob.kmer_count(st, i, ob.default_config(read_tlen=1500, indel_balance_factor_sgs=rate)), "seed %d" % seed


def test_model_intra_contig_tiling_from_files_through_the_index(tmp_path):
    """np1_tile.cpp's driver with the model as the device: every tile reads its own region of the BAM through the index
    (load_stream_region), the join uses the driver's index arithmetic -- equal to the untiled oracle for tiles of 300 bases to a whole
    contig and halos from one base up, on indel-rich reads with soft clips and odd CIGAR shapes"""
    st = nat.Stream.synth([30000, 4000, 700], depth=20, seed=991, read_indel=0.005, softclip_rate=0.06, draft_lower=0.02, weird_rate=0.02)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "r.bam")
    st.write_files(fa, bam)
    want = [ob.score_chain(st, i) for i in range(st.n_contigs)]
    redo = 0
    for tile, halo in ((300, 1), (1700, 25), (9000, 200), (29999, 100), (50000, 10)):
        for i, n in enumerate(st.names):
            got, info = mb.score_chain_tiled_files(fa, bam, n, tile, halo, fused=(tile // 100) % 3)
            assert got == want[i], (tile, halo, n, info)
            redo += info["recomputed"]
    assert redo > 0
