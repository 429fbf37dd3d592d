"""Long-read path (nextpolish2), CPU side: golden vectors of the compiled reference vs the host lockstep model of the
window pipeline (tests/model/libnp2_model.so = the product's host pipeline + per-lane bodies run by a host executor),
the 2-bit codec / read_ref known answers, the compiled reference itself when oracle/_ref is present, and the C ABI
surface of the product library (load + symbols only: no compute without a GPU)."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

import np2_cases
import np2_gen
import ref2_binding as rb

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "np2_golden.json")))
MODEL_SO = os.path.join(HERE, "model", "libnp2_model.so")
PRODUCT_SO = os.path.join(HERE, "..", "nextpolish_amd", "lib", "nextpolish2.so")
LQ_CASES = {"ont_lq_regions", "clr_lq_regions"}   # windows with low-quality regions: POA pseudo-seeds + graph re-consensus


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(MODEL_SO):
        subprocess.run(["make", "-C", os.path.join(HERE, "model"), "libnp2_model.so"], check=True, capture_output=True)
    return MODEL_SO


def run_polish(so_path, fa, fofn, read_type, window=5000000, split=0, env=None):
    """ctg_cns_core exits the process on unsupported input (the reference's error convention): run it in a child."""
    code = ("import sys, json; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(%r); "
            "print(json.dumps(rb.polish(L, %r, %r, read_type=%d, window=%d, split=%d)))" % (HERE, so_path, fa, fofn, read_type, window, split))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    if p.returncode != 0:
        return None, p.stderr
    return json.loads(p.stdout.strip().splitlines()[-1]), p.stderr


@pytest.mark.parametrize("cid", [c[0] for c in np2_cases.CASES])
def test_model_matches_reference_goldens(model, cid, tmp_path):
    kw, rt = next((k, r) for c, k, r in np2_cases.CASES if c == cid)
    fa, fofn, contigs = np2_cases.materialise(kw, str(tmp_path))
    got, err = run_polish(model, fa, fofn, rt)
    assert got is not None, err
    for n, _ in contigs:
        assert len(got[n]) == GOLD["cases"][cid]["pieces"][n]
        assert got[n][0][0] == GOLD["cases"][cid]["expected"][n], "%s %s" % (cid, n)


def test_unsupported_cigar_ops_abort_like_the_reference(model, tmp_path):
    """'=' / 'X' ops: the reference's bam2aln reports "unexpected cigar" and ctg_cns_core exits with "bamaln error"."""
    from nextpolish_amd import _native as nat
    contigs, reads = np2_gen.make_case(3, contig_lens=(3000,), depth=6, mean_len=1500)
    reads[2]["cigar"] = [("=" if o == "M" else o, n) for o, n in reads[2]["cigar"]]
    st = nat.Stream.from_reads(contigs, reads)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "r.bam")
    st.write_files(fa, bam)
    (tmp_path / "bam.fofn").write_text(bam + "\n")
    got, err = run_polish(model, fa, str(tmp_path / "bam.fofn"), 1)
    assert got is None and "bamaln error" in err


@pytest.mark.skipif(not rb.available(), reason="oracle/_ref not built (reference sources absent)")
def test_goldens_still_match_compiled_reference(tmp_path):
    L = rb.bind(rb.REF_SO)
    for cid, kw, rt in np2_cases.CASES[:3]:
        fa, fofn, contigs = np2_cases.materialise(kw, str(tmp_path))
        res = rb.polish(L, fa, fofn, read_type=rt)
        for n, _ in contigs:
            assert res[n][0][0] == GOLD["cases"][cid]["expected"][n]


def _libs():
    libs = [MODEL_SO]
    if os.path.exists(PRODUCT_SO):
        libs.append(PRODUCT_SO)
    return libs


def test_codec_known_answers(model):
    for so in _libs():
        L = rb.bind(so)
        for k in GOLD["codec"]:
            s = k["seq"]
            words = (C.c_uint32 * (len(s) // 16 + 1))()
            L.seq2bit1(words, len(s), s.encode())
            assert [int(w) for w in words][: (len(s) + 15) // 16] == k["words"], (so, s)
            buf = C.create_string_buffer(len(s) + 1)
            L.bit2seq1(words, len(s), buf)
            assert buf.value.decode() == k["round_trip"]


def test_read_ref_subset_order_and_qv(model, tmp_path):
    fa = tmp_path / "x.fa"
    fa.write_text(">b some comment\nACGTNN\nacgt\n>a node=00000002 qv=00000000100c8321:0000000200fffff\nTTTT\r\n>c\nGG\n@q1 x\nACGT\n+\nIIII\n>d\nC\n")
    for so in _libs() + ([rb.REF_SO] if rb.available() else []):
        L = rb.bind(so)
        refs = L.read_ref(str(fa).encode(), None, 0)
        names = [refs.contents.ref[i].n.decode() for i in range(refs.contents.i)]
        lens = [refs.contents.ref[i].length for i in range(refs.contents.i)]
        assert names == ["b", "a", "c", "q1", "d"] and lens == [10, 4, 2, 4, 1], so
        assert refs.contents.ref[1].qv_l == 2
        L.refs_destroy(refs)
        arr = (C.c_char_p * 2)(b"d", b"a")
        refs = L.read_ref(str(fa).encode(), arr, 2)
        assert [refs.contents.ref[i].n.decode() for i in range(refs.contents.i)] == ["a", "d"]   # file order
        assert [arr[0], arr[1]] == [b"a", b"d"]                                                  # sorted in place
        L.refs_destroy(refs)


def test_product_library_exports_the_abi():
    if not os.path.exists(PRODUCT_SO):
        pytest.skip("product library not built")
    L = C.CDLL(PRODUCT_SO)
    header = open(os.path.join(HERE, "..", "include", "nextpolish2.h")).read()
    for sym in ["read_ref", "refs_destroy", "seq2bit1", "bit2seq1", "ctg_cns_init", "ctg_cns_destroy", "ctg_cns_core",
                "free_consensus_trimed_data", "np2_last_error", "np2_device_index"]:
        assert sym in header
        assert getattr(L, sym) is not None


def test_poa_consensus_known_answers(model):
    import np2_strings
    M = C.CDLL(model)
    for k in GOLD["poa"]:
        assert np2_strings.model_poa(M, k["seqs"]) == k["consensus"]


def test_ond_align_known_answers(model):
    import np2_strings
    M = C.CDLL(model)
    for k in GOLD["align"]:
        n, ts, qs, tl, ql = np2_strings.model_align(M, k["q"], k["t"])
        assert n == k["aln_len"]
        if n > 2:
            assert (ts, qs, tl, ql) == (k["t_aln"], k["q_aln"], k["t_len"], k["q_len"])


@pytest.mark.skipif(not rb.available(), reason="oracle/_ref not built (reference sources absent)")
def test_string_algorithms_against_compiled_reference(model):
    import random
    import np2_strings
    M, R = C.CDLL(model), C.CDLL(rb.REF_SO)
    for seed in range(1000, 1120):
        seqs = np2_strings.poa_case(random.Random(seed))
        assert np2_strings.model_poa(M, seqs) == np2_strings.ref_poa(R, seqs), seed
        q, t = np2_strings.align_case(random.Random(seed), seed)
        want, got = np2_strings.ref_align(R, q, t), np2_strings.model_align(M, q, t)
        assert got[0] == want[0] and (want[0] <= 2 or got == want), seed


def _run_harness(argv, lib):
    exe = os.path.join(HERE, "..", "nextpolish_amd", "nextpolish2.py")
    return subprocess.run([sys.executable, exe] + argv + ["--library", lib], capture_output=True, text=True)


def test_harness_mirrors_reference_caller(model, tmp_path):
    cid, kw, rt = np2_cases.CASES[0]
    fa, fofn, contigs = np2_cases.materialise(kw, str(tmp_path))
    want = GOLD["cases"][cid]["expected"]
    p = _run_harness(["-g", fa, "-l", fofn, "-r", "ont", "-p", "1"], model)
    assert p.returncode == 0, p.stderr
    recs = p.stdout.strip().split("\n")
    assert recs == [">ctg0 %d" % len(want["ctg0"]), want["ctg0"], ">ctg1 %d" % len(want["ctg1"]), want["ctg1"]]
    # -u, block file selection, sharding over two node-level ranks
    blc = tmp_path / "g.blc"
    blc.write_text("ctg0 0\nctg1 1\n")
    p = _run_harness(["-g", fa, "-l", fofn, "-r", "ont", "-p", "1", "-u", "-b", str(blc), "-i", "1"], model)
    assert p.stdout.strip().split("\n") == [">ctg1 %d" % len(want["ctg1"]), want["ctg1"].upper()]
    got = []
    for rank in (0, 1):
        p = _run_harness(["-g", fa, "-l", fofn, "-r", "ont", "-p", "1", "--world", "2", "--rank", str(rank)], model)
        got.append(p.stdout.strip().split("\n")[0])
    assert got == [">ctg0 %d" % len(want["ctg0"]), ">ctg1 %d" % len(want["ctg1"])]
    # resume: a finished first contig is skipped, a trailing (possibly truncated) record is redone
    out = tmp_path / "o.fa"
    out.write_text(">ctg0 %d\n%s\n>ctg1 5\nACG" % (len(want["ctg0"]), want["ctg0"]))
    p = _run_harness(["-g", fa, "-l", fofn, "-r", "ont", "-p", "1", "-o", str(out)], model)
    assert p.returncode == 0, p.stderr
    assert out.read_text().strip().split("\n") == [">ctg0 %d" % len(want["ctg0"]), want["ctg0"], ">ctg1 %d" % len(want["ctg1"]), want["ctg1"]]


def test_two_windows_are_stitched_like_the_reference(model, tmp_path):
    """4.3 Mb contig, 4.1 Mb window: two windows overlapping by 1 Mb, joined at 50 agreeing bases (link_consensus)."""
    import hashlib
    fa, fofn, contigs = np2_cases.materialise(np2_cases.TWO_WINDOW_CASE, str(tmp_path))
    got, err = run_polish(model, fa, fofn, 1, window=np2_cases.TWO_WINDOW_W)
    assert got is not None, err
    s = got["ctg0"][0][0]
    assert len(got["ctg0"]) == GOLD["two_windows"]["pieces"] and len(s) == GOLD["two_windows"]["len"]
    assert hashlib.md5(s.encode()).hexdigest() == GOLD["two_windows"]["md5"]


@pytest.mark.parametrize("cid", [c[0] for c in np2_cases.SV_CASES])
def test_structural_layer_matches_reference_goldens(model, cid, tmp_path):
    """Split reads: gap clusters, supplementary streams, cluster candidates, split points, QV-track regions."""
    import hashlib
    kw, rt, split, qvs = next((k, r, s, q) for c, k, r, s, q in np2_cases.SV_CASES if c == cid)
    fa, fofn, contigs = np2_cases.materialise_sv(kw, qvs, str(tmp_path))
    got, err = run_polish(model, fa, fofn, rt, split=split)
    assert got is not None, err
    want = GOLD["sv"][cid]
    assert [p[1] for p in got["ctg0"]] == want["lens"]
    assert [hashlib.md5(p[0].encode()).hexdigest() for p in got["ctg0"]] == want["md5"]


@pytest.mark.skipif(not rb.available(), reason="oracle/_ref not built (reference sources absent)")
def test_structural_layer_stage_by_stage_against_reference(model, tmp_path):
    """The reference's helper functions are interposed (tests/shim) and their results logged; this library logs the
    same lines (NP2_SV_LOG): low-depth regions, clusters and medians, supplementary streams, per-gap read
    coordinates, split points must agree line by line."""
    shim_src = os.path.join(HERE, "shim", "np2_ref_shim.c")
    shim = str(tmp_path / "libshim.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", shim, shim_src, "-ldl"], check=True)
    ref_so = os.path.realpath(rb.REF_SO)
    for cid, kw, rt, split, qvs in np2_cases.SV_CASES[3:6]:
        d = tmp_path / cid
        d.mkdir()
        fa, fofn, contigs = np2_cases.materialise_sv(kw, qvs, str(d))
        want, _ = run_polish(ref_so, fa, fofn, rt, split=split, env=dict(LD_PRELOAD=shim, NP2_SHIM_LOG=str(d / "ref.log"), NP2_REF_SO=ref_so))
        got, err = run_polish(model, fa, fofn, rt, split=split, env=dict(NP2_SV_LOG=str(d / "mine.log")))
        assert got is not None, err
        assert (d / "ref.log").read_text() == (d / "mine.log").read_text(), cid
        assert got == want


def test_megabase_window_matches_reference_golden(model, tmp_path):
    """1.2 Mb window from the native generator through the host model (thread pool over ~4 000 low-quality regions)."""
    import hashlib
    from nextpolish_amd import _native as nat
    st = nat.Stream.synth_long([1200000], depth=20.0, seed=31)
    fa, bam, fofn = str(tmp_path / "g.fa"), str(tmp_path / "r.bam"), str(tmp_path / "bam.fofn")
    st.write_files(fa, bam)
    st.close()
    open(fofn, "w").write(bam + "\n")
    got, err = run_polish(model, fa, fofn, 1)
    assert got is not None, err
    want = GOLD["mb_window"]
    assert [p[1] for p in got["ctg0"]] == want["lens"]
    assert [hashlib.md5(p[0].encode()).hexdigest() for p in got["ctg0"]] == want["md5"]


def _cg_tagged_copy(kw, workdir):
    """The first case's reads, every third one written the way BAM stores a CIGAR of more than 65 535 operations: the
    placeholder <l_qseq>S<rlen>N in the record and the real operations in the tag CG:B:I (SAMv1 4.2.2)."""
    import struct
    from nextpolish_amd import _native as nat
    k = dict(kw)
    seed = k.pop("seed")
    contigs, reads = np2_gen.make_case(seed, **k)
    OPS = "MIDNSHP=X"
    aux = []
    for i, r in enumerate(reads):
        if i % 3:
            aux.append(b"")
            continue
        real = r["cigar"]
        rlen = sum(n for o, n in real if o in "MDN=X")
        aux.append(b"CGBI" + struct.pack("<I", len(real)) + b"".join(struct.pack("<I", n << 4 | OPS.index(o)) for o, n in real))
        r["cigar"] = [("S", len(r["seq"])), ("N", rlen)]
    st = nat.Stream.from_reads(contigs, reads)
    fa, bam, fofn = os.path.join(workdir, "g.fa"), os.path.join(workdir, "r.bam"), os.path.join(workdir, "bam.fofn")
    st.write_files(fa, bam, aux=aux)
    st.close()
    open(fofn, "w").write(bam + "\n")
    return fa, fofn, contigs


def test_long_cigars_in_the_cg_tag_are_swapped_in(model, tmp_path):
    """htslib swaps the CG tag in while reading, so the reference never sees the placeholder; the reader here must too:
    the output equals the golden vector of the same reads with ordinary CIGARs."""
    cid, kw, rt = np2_cases.CASES[0]
    fa, fofn, contigs = _cg_tagged_copy(kw, str(tmp_path))
    got, err = run_polish(model, fa, fofn, rt)
    assert got is not None, err
    for n, _ in contigs:
        assert got[n][0][0] == GOLD["cases"][cid]["expected"][n]
    if rb.available():
        want, err = run_polish(os.path.realpath(rb.REF_SO), fa, fofn, rt)
        assert want == got


def test_reads_dealt_over_three_bam_files(model, tmp_path):
    """The fofn may list several BAM files: the records are merged by (position, strand, file order) like the reference's
    multi-file iterator; the order decides the first-seen order of links."""
    cid, kw, rt = np2_cases.CASES[1]
    assert GOLD["multi_bam"]["case"] == cid
    fa, fofn, contigs = np2_cases.materialise_multi(kw, 3, str(tmp_path))
    got, err = run_polish(model, fa, fofn, rt)
    assert got is not None, err
    for n, _ in contigs:
        assert got[n][0][0] == GOLD["multi_bam"]["expected"][n]
