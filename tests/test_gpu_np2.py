"""Long-read path on the GPU: nextpolish_amd/lib/nextpolish2.so (HIP window executor) through the reference's own C
ABI against the golden vectors of the compiled reference (and against oracle/_ref directly when it travelled)."""
import json
import os
import subprocess
import sys

import pytest

import np2_cases
import ref2_binding as rb

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "np2_golden.json")))
PRODUCT_SO = os.path.join(HERE, "..", "nextpolish_amd", "lib", "nextpolish2.so")
LQ_CASES = {"ont_lq_regions", "clr_lq_regions"}   # windows with low-quality regions: POA pseudo-seeds + graph re-consensus


def run_polish(so_path, fa, fofn, read_type, split=0, env=None):
    code = ("import sys, json; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(%r); "
            "print(json.dumps(rb.polish(L, %r, %r, read_type=%d, split=%d)))" % (HERE, so_path, fa, fofn, read_type, split))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    if p.returncode != 0:
        return None, p.stderr
    return json.loads(p.stdout.strip().splitlines()[-1]), p.stderr


@pytest.mark.parametrize("cid", [c[0] for c in np2_cases.CASES])
def test_gpu_matches_reference_goldens(cid, tmp_path):
    kw, rt = next((k, r) for c, k, r in np2_cases.CASES if c == cid)
    fa, fofn, contigs = np2_cases.materialise(kw, str(tmp_path))
    got, err = run_polish(PRODUCT_SO, fa, fofn, rt)
    assert got is not None, err
    for n, _ in contigs:
        assert got[n][0][0] == GOLD["cases"][cid]["expected"][n], "%s %s" % (cid, n)


def test_gpu_cli_prints_reference_format(tmp_path):
    kw, rt = np2_cases.CASES[0][1], np2_cases.CASES[0][2]
    fa, fofn, contigs = np2_cases.materialise(kw, str(tmp_path))
    exe = os.path.join(HERE, "..", "nextpolish_amd", "bin", "nextpolish2")
    p = subprocess.run([exe, fa, fofn], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().split("\n")
    want = GOLD["cases"][np2_cases.CASES[0][0]]["expected"]
    assert lines[0] == ">ctg0_lgs %d %f" % (len(want["ctg0"]), 0.0) and lines[1] == want["ctg0"]
    assert lines[3] == want["ctg1"]


@pytest.mark.skipif(not rb.available(), reason="oracle/_ref did not travel")
def test_gpu_matches_compiled_reference_on_fresh_inputs(tmp_path):
    L = rb.bind(rb.REF_SO)
    n_checked = 0
    for seed in range(100, 112):
        kw = dict(seed=seed, contig_lens=[(15000,), (8000, 3000)][seed % 2], depth=[10, 25, 40][seed % 3], max_indel=[1, 2, 6, 10][seed % 4],
                  sub=[0.02, 0.06][(seed // 2) % 2])
        d = tmp_path / ("s%d" % seed)
        d.mkdir()
        fa, fofn, contigs = np2_cases.materialise(kw, str(d))
        got, err = run_polish(PRODUCT_SO, fa, fofn, 1)
        assert got is not None, err
        want = rb.polish(L, fa, fofn, read_type=1)
        for n, _ in contigs:
            assert got[n][0][0] == want[n][0][0], "seed %d %s" % (seed, n)
        n_checked += 1
    assert n_checked == 12


def test_gpu_harness_with_forked_workers(tmp_path):
    """The reference forks its worker pool after read_ref/ctg_cns_init: every worker must bring up its own HIP context."""
    cid, kw, rt = np2_cases.CASES[0]
    fa, fofn, contigs = np2_cases.materialise(kw, str(tmp_path))
    want = GOLD["cases"][cid]["expected"]
    exe = os.path.join(HERE, "..", "nextpolish_amd", "nextpolish2.py")
    p = subprocess.run([sys.executable, exe, "-g", fa, "-l", fofn, "-r", "ont", "-p", "2"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    recs = p.stdout.strip().split("\n")
    got = {recs[i][1:].split()[0]: recs[i + 1] for i in range(0, len(recs), 2)}
    assert got == want


def test_gpu_two_windows_are_stitched_like_the_reference(tmp_path):
    """4.3 Mb contig, 4.1 Mb window: two windows overlapping by 1 Mb, joined at 50 agreeing bases (link_consensus)."""
    import hashlib
    fa, fofn, contigs = np2_cases.materialise(np2_cases.TWO_WINDOW_CASE, str(tmp_path))
    code = ("import sys, json; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(%r); "
            "print(json.dumps(rb.polish(L, %r, %r, read_type=1, window=%d)))" % (HERE, PRODUCT_SO, fa, fofn, np2_cases.TWO_WINDOW_W))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    s = json.loads(p.stdout.strip().splitlines()[-1])["ctg0"][0][0]
    assert len(s) == GOLD["two_windows"]["len"] and hashlib.md5(s.encode()).hexdigest() == GOLD["two_windows"]["md5"]


@pytest.mark.parametrize("cid", [c[0] for c in np2_cases.SV_CASES])
def test_gpu_structural_layer_matches_reference_goldens(cid, tmp_path):
    """Contigs > 100 kb with split (SA) reads: supplementary tag streams, cluster candidates, split points, QV track."""
    import hashlib
    kw, rt, split, qvs = next((k, r, s, q) for c, k, r, s, q in np2_cases.SV_CASES if c == cid)
    fa, fofn, contigs = np2_cases.materialise_sv(kw, qvs, str(tmp_path))
    got, err = run_polish(PRODUCT_SO, fa, fofn, rt, split=split)
    assert got is not None, err
    want = GOLD["sv"][cid]
    assert [p[1] for p in got["ctg0"]] == want["lens"]
    assert [hashlib.md5(p[0].encode()).hexdigest() for p in got["ctg0"]] == want["md5"]


def test_gpu_megabase_window_matches_reference_golden(tmp_path):
    """1.2 Mb window: several DP runs per wave, two-level (max, +) scan, chunked tag/link kernels at scale."""
    import hashlib
    from nextpolish_amd import _native as nat
    st = nat.Stream.synth_long([1200000], depth=20.0, seed=31)
    fa, bam, fofn = str(tmp_path / "g.fa"), str(tmp_path / "r.bam"), str(tmp_path / "bam.fofn")
    st.write_files(fa, bam)
    st.close()
    open(fofn, "w").write(bam + "\n")
    got, err = run_polish(PRODUCT_SO, fa, fofn, 1)
    assert got is not None, err
    want = GOLD["mb_window"]
    assert [p[1] for p in got["ctg0"]] == want["lens"]
    assert [hashlib.md5(p[0].encode()).hexdigest() for p in got["ctg0"]] == want["md5"]


def test_gpu_reader_windows_inflated_on_the_device_give_the_same_consensus(tmp_path):
    """NP2_INFLATE=device (opt-in): the BAM reader's windows of BGZF blocks go to the wave-per-block decoder on the device (np_bgzf_dev.hip), CRC
    checked there; the 1.2 Mb golden window again, and the reader's own statistics say that windows really went to the device.  (Round 6 also
    tried the lane-per-block decoder of the short-read ingest here: correct, but a long-read block is ~25 k tokens of nearly incompressible bases
    and takes a lane ~35 ms -- 147 ms of inflate per 5 Mb window against 69 with a wave per block and 120 on two host threads.)"""
    import hashlib
    from nextpolish_amd import _native as nat
    st = nat.Stream.synth_long([1200000], depth=20.0, seed=31)
    fa, bam, fofn = str(tmp_path / "g.fa"), str(tmp_path / "r.bam"), str(tmp_path / "bam.fofn")
    st.write_files(fa, bam)
    st.close()
    open(fofn, "w").write(bam + "\n")
    got, err = run_polish(PRODUCT_SO, fa, fofn, 1, env={"NP2_INFLATE": "device", "NP2_TIMING": "1"})
    assert got is not None, err
    want = GOLD["mb_window"]
    assert [p[1] for p in got["ctg0"]] == want["lens"]
    assert [hashlib.md5(p[0].encode()).hexdigest() for p in got["ctg0"]] == want["md5"]
    import re
    m = re.search(r"device inflate [0-9.]+ ms \((\d+) windows\)", err)
    assert m and int(m.group(1)) > 0, err[-600:]


def test_gpu_deep_pileup_matches_reference_golden(tmp_path):
    """100x: few columns with <= 8 live entries, so the run decomposition switches to cuts of width 32."""
    import hashlib
    from nextpolish_amd import _native as nat
    st = nat.Stream.synth_long([150000], depth=100.0, seed=32)
    fa, bam, fofn = str(tmp_path / "g.fa"), str(tmp_path / "r.bam"), str(tmp_path / "bam.fofn")
    st.write_files(fa, bam)
    st.close()
    open(fofn, "w").write(bam + "\n")
    got, err = run_polish(PRODUCT_SO, fa, fofn, 1)
    assert got is not None, err
    want = GOLD["deep_window"]
    assert [p[1] for p in got["ctg0"]] == want["lens"]
    assert [hashlib.md5(p[0].encode()).hexdigest() for p in got["ctg0"]] == want["md5"]


@pytest.mark.skipif(not rb.available(), reason="oracle/_ref did not travel")
def test_gpu_full_5mb_window_matches_compiled_reference(tmp_path):
    """The reference's window size at BASELINE configs[3] shape (20x ONT-like reads): one 5 Mb window, 12 500 reads,
    123 M link observations; the compiled reference needs ~8 s for it."""
    import hashlib
    from nextpolish_amd import _native as nat
    st = nat.Stream.synth_long([5000000], depth=20.0, seed=5)
    fa, bam, fofn = str(tmp_path / "g.fa"), str(tmp_path / "r.bam"), str(tmp_path / "bam.fofn")
    st.write_files(fa, bam)
    st.close()
    open(fofn, "w").write(bam + "\n")
    got, err = run_polish(PRODUCT_SO, fa, fofn, 1)
    assert got is not None, err
    want, err = run_polish(os.path.realpath(rb.REF_SO), fa, fofn, 1)
    assert want is not None, err
    assert got == want
    assert abs(len(got["ctg0"][0][0]) - 5000000) < 50000


def test_gpu_reads_dealt_over_three_bam_files(tmp_path):
    """Several BAM files in the fofn: records merged by (position, strand, file order) like the reference's iterator."""
    cid, kw, rt = np2_cases.CASES[1]
    fa, fofn, contigs = np2_cases.materialise_multi(kw, 3, str(tmp_path))
    got, err = run_polish(PRODUCT_SO, fa, fofn, rt)
    assert got is not None, err
    for n, _ in contigs:
        assert got[n][0][0] == GOLD["multi_bam"]["expected"][n]


@pytest.mark.parametrize("cid", ["ont_35x_noisy", "ont_lq_regions", "hifi_35x_indels", "ont_reads_with_iupac_codes"])
def test_gpu_scatter_graph_path_matches_reference_goldens(cid, tmp_path):
    """The link graph has two builders: by tiles in LDS (default) and by scattering observations into column buckets (taken by
    windows whose tiles overflow).  NP2_GRAPH_SCATTER=1 sends every window down the second one."""
    kw, rt = next((k, r) for c, k, r in np2_cases.CASES if c == cid)
    fa, fofn, contigs = np2_cases.materialise(kw, str(tmp_path))
    got, err = run_polish(PRODUCT_SO, fa, fofn, rt, env={"NP2_GRAPH_SCATTER": "1"})
    assert got is not None, err
    for n, _ in contigs:
        assert got[n][0][0] == GOLD["cases"][cid]["expected"][n], "%s %s" % (cid, n)


@pytest.mark.skipif(not rb.available(), reason="oracle/_ref did not travel")
def test_gpu_tiles_with_more_streams_than_the_lds_list_holds(tmp_path):
    """220x over a short contig: every tile of 64 columns is crossed by more streams than its sorted list in LDS holds (128), so
    the tile kernel selects the next stream from HBM; tiles that overflow their entry pool send the window down the scatter path.
    Both must give the compiled reference's consensus."""
    from nextpolish_amd import _native as nat
    for seed, depth, sub in ((61, 220.0, None), (62, 160.0, None)):
        st = nat.Stream.synth_long([30000], depth=depth, seed=seed)
        d = tmp_path / ("d%d" % seed)
        d.mkdir()
        fa, bam, fofn = str(d / "g.fa"), str(d / "r.bam"), str(d / "bam.fofn")
        st.write_files(fa, bam)
        st.close()
        open(fofn, "w").write(bam + "\n")
        got, err = run_polish(PRODUCT_SO, fa, fofn, 1, env={"NP2_TIMING": "1"})
        assert got is not None, err
        want, err2 = run_polish(os.path.realpath(rb.REF_SO), fa, fofn, 1)
        assert want is not None, err2
        assert got == want


@pytest.mark.parametrize("cid", ["ont_lq_regions", "clr_lq_regions", "hifi_35x_indels", "ont_long_insertions"])
def test_gpu_device_pseudo_seeds_equal_the_host_version(cid, tmp_path):
    """The partial-order pseudo-seed of every low-quality region comes from the device (k2_poa, one wave per region); NP2_POA_CHECK=1
    makes the executor compare each of them with the host version (np2_poa.cpp, pinned to the reference's poa_to_consensus by the
    known-answer tests) and fail on the first difference; NP2_POA_HOST=1 sends all of them to the host version.  Both runs must give
    the goldens."""
    kw, rt = next((k, r) for c, k, r in np2_cases.CASES if c == cid)
    fa, fofn, contigs = np2_cases.materialise(kw, str(tmp_path))
    for env in ({"NP2_POA_CHECK": "1", "NP2_TIMING": "1"}, {"NP2_POA_HOST": "1"}):
        got, err = run_polish(PRODUCT_SO, fa, fofn, rt, env=env)
        assert got is not None, err
        if "NP2_POA_CHECK" in env and cid == "ont_lq_regions":      # (the CLR case has two regions and none of them needs a pseudo-seed)
            assert "device pseudo-seeds against the host version" in err
        for n, _ in contigs:
            assert got[n][0][0] == GOLD["cases"][cid]["expected"][n], "%s %s %r" % (cid, n, env)
