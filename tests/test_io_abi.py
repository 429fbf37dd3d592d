"""Host I/O substrate (BGZF/BAM/BAI/FAI written from the spec) and the C-ABI surface."""
import ctypes as C
import gzip
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from nextpolish_amd import _native as nat
from conftest import ROOT, ref_binary


def _py_bam_records(path):
    """Independent pure-Python BAM decoder (BGZF is a multi-member gzip stream)."""
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", data, 4)
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, off)
    off += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, off)
        name = data[off + 4:off + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<I", data, off + 4 + l_name)
        refs.append((name, l_ref))
        off += 8 + l_name
    recs = []
    while off < len(data):
        bs, = struct.unpack_from("<i", data, off)
        tid, pos, l_qname, mapq, _bin, n_cig, flag, l_seq, _mt, _mp, isize = struct.unpack_from("<iiBBHHHiiii", data, off + 4)
        p = off + 36 + l_qname
        cig = list(struct.unpack_from("<%dI" % n_cig, data, p))
        p += 4 * n_cig
        seq = data[p:p + (l_seq + 1) // 2]
        qual = data[p + (l_seq + 1) // 2:p + (l_seq + 1) // 2 + l_seq]
        recs.append((tid, pos, flag, mapq, isize, l_seq, cig, seq, qual))
        off += 4 + bs
    return refs, recs


def test_bam_fasta_roundtrip_and_python_crosscheck(tmp_path):
    st = nat.Stream.synth([9000, 2500, 800], depth=25, seed=21, with_qual=1, weird_rate=0.05, softclip_rate=0.1)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam, level=6)
    assert os.path.exists(bam + ".bai") and os.path.exists(fa + ".fai")
    # our reader == what we wrote
    st2 = nat.Stream.load(fa, bam, with_qual=True)
    for f in ["pos", "ctg", "flag", "n_cigar", "l_qseq", "mapq", "isize", "cigar", "seq", "qual", "draft", "ctg_len"]:
        assert np.array_equal(getattr(st, f), getattr(st2, f)), f
    # independent decoder agrees record by record
    refs, recs = _py_bam_records(bam)
    assert [r[0] for r in refs] == st.names and [r[1] for r in refs] == list(st.ctg_len)
    assert len(recs) == st.n_reads
    for i in (0, 1, len(recs) // 2, len(recs) - 1):
        tid, pos, flag, mapq, isize, l_seq, cig, seq, qual = recs[i]
        assert (tid, pos, flag, mapq, isize, l_seq) == (st.ctg[i], st.pos[i], st.flag[i], st.mapq[i], st.isize[i], st.l_qseq[i])
        assert cig == list(st.cigar[int(st.cigar_off[i]):int(st.cigar_off[i]) + int(st.n_cigar[i])])
        assert seq == st.seq[int(st.seq_off[i]):int(st.seq_off[i]) + (l_seq + 1) // 2].tobytes()
        assert qual == st.qual[int(st.qual_off[i]):int(st.qual_off[i]) + l_seq].tobytes()


def test_index_driven_subset_and_order(tmp_path):
    st = nat.Stream.synth([6000, 3000, 4000, 500], depth=20, seed=22)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam)
    os.remove(fa + ".fai")                      # fai_load behaviour: the index is rebuilt next to the FASTA
    sub = nat.Stream.load(fa, bam, names=[st.names[2], st.names[0]])
    assert os.path.exists(fa + ".fai")
    assert sub.names == [st.names[2], st.names[0]]
    for k, src in enumerate([2, 0]):
        r0, r1 = int(st.read_begin[src]), int(st.read_begin[src + 1])
        q0, q1 = int(sub.read_begin[k]), int(sub.read_begin[k + 1])
        assert q1 - q0 == r1 - r0
        assert np.array_equal(sub.pos[q0:q1], st.pos[r0:r1])
        assert sub.contig_draft(k) == st.contig_draft(src)


def test_fasta_with_odd_line_lengths(tmp_path):
    fa = tmp_path / "odd.fa"
    fa.write_text(">a desc here\nACGTAC\nGT\n>b\nAC\nG\n\n>c\nTTTTTTTTTT")
    bam = str(tmp_path / "e.bam")
    nat.Stream.from_reads([("a", "ACGTACGT"), ("b", "ACG"), ("c", "TTTTTTTTTT")], []).write_files(str(tmp_path / "tmp.fa"), bam)
    st = nat.Stream.load(str(fa), bam)
    assert [st.contig_draft(i) for i in range(3)] == [b"ACGTACGT", b"ACG", b"TTTTTTTTTT"]


def test_fasta_fetch_by_line_layout_and_its_fallback(tmp_path):
    """Fai::fetch copies line by line on the host threads when the index's line layout holds (np_bam.cpp) and falls back to the
    byte-by-byte walk when it does not: CRLF line ends, a last line that is full / partial / missing its newline, lines with blanks
    inside (the index counts them, fai_fetch drops them), a contig of many jobs -- always what the plain reading gives"""
    import random
    rng = random.Random(5)
    seqs = {"unix60": "".join(rng.choice("ACGTNacgtRYKM") for _ in range(60 * 41 + 17)),
            "crlf70": "".join(rng.choice("ACGT") for _ in range(70 * 9)),
            "one_line": "".join(rng.choice("ACGT") for _ in range(501)),
            "blanks": "".join(rng.choice("ACGT") for _ in range(50 * 7 + 3)),
            "big": "".join(rng.choice("ACGT") for _ in range(100 * 90000 + 55))}
    fa = tmp_path / "layout.fa"
    with open(fa, "wb") as f:
        def lines(s, w):
            return [s[i:i + w] for i in range(0, len(s), w)]
        f.write(b">unix60\n" + "\n".join(lines(seqs["unix60"], 60)).encode() + b"\n")
        f.write(b">crlf70 x\r\n" + "\r\n".join(lines(seqs["crlf70"], 70)).encode() + b"\r\n")
        f.write(b">one_line\n" + seqs["one_line"].encode() + b"\n")
        bl = lines(seqs["blanks"], 50)
        bl[2] = bl[2][:20] + " " + bl[2][20:]              # a blank inside a line: same byte width as the others would have with 51
        f.write(b">blanks\n" + "\n".join(bl).encode() + b"\n")
        f.write(b">big\n" + "\n".join(lines(seqs["big"], 100)).encode())     # no newline at the end of the file
    names = list(seqs)
    if "blanks" in names:
        pass
    bam = str(tmp_path / "e.bam")
    nat.Stream.from_reads([(n, seqs[n]) for n in names], []).write_files(str(tmp_path / "tmp.fa"), bam)
    st = nat.Stream.load(str(fa), bam)
    for i, n in enumerate(names):
        assert st.contig_draft(i) == seqs[n].encode(), n


def test_config_init_defaults_and_insert_probe(tmp_path):
    st = nat.Stream.synth([20000], depth=30, seed=23, with_qual=1)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam)
    L = nat.lib()
    cfg = L.config_init(fa.encode(), bam.encode(), b"/nonexistent.bam")
    c = cfg.contents
    assert (c.trim_len_edge, c.ext_len_edge, c.min_map_quality, c.max_len_kmer, c.max_count_kmer) == (2, 2, 0, 50, 50)
    assert (c.indel_balance_factor_sgs, c.min_count_ratio_skip, c.max_clip_ratio_sgs) == (0.5, 0.8, 0.15)
    assert c.thirdbamfn is None and c.bamfn == bam.encode() and c.fastafn == fa.encode()
    # reference: config.c:80-101 -- sum/count with count starting at 1 over the first 9999 qualifying records
    isz = [int(x) for x in st.isize if 0 < x < 10000][:9999]
    assert c.read_tlen == (sum(isz) // (len(isz) + 1)) * 5
    assert c.read_len == 150
    L.config_destory(cfg)
    cfg = L.config_init(fa.encode(), b"/nonexistent.bam", None)
    assert cfg.contents.bamfn is None and cfg.contents.read_tlen == 0
    L.config_destory(cfg)


def test_configure_layout_matches_reference_abi():
    """Natural C alignment of source/lib/config.h:25-67 (and the ctypes mirror in source/lib/nextpolish1.py:27-65)."""
    exp = {"trim_len_edge": 0, "indel_balance_factor_sgs": 8, "min_count_ratio_skip": 16, "min_len_ldr": 24,
           "min_count_snp_link": 30, "ploidy": 32, "min_snp_factor_sgs": 64, "region_count": 72, "max_variant_count_lgs": 88,
           "max_clip_ratio_sgs": 96, "max_clip_ratio_lgs": 104, "trace_polish_open": 112, "read_tlen": 116, "read_len": 120,
           "fastafn": 128, "bamfn": 136, "thirdbamfn": 144}
    for k, v in exp.items():
        assert getattr(nat.Configure, k).offset == v, k
    assert C.sizeof(nat.Configure) == 152
    assert C.sizeof(nat.PolishPoint) == 8 and C.sizeof(nat.PolishResult) == 24


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nextpolish1.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\(", hdr)) - {"defined", "float", "void"}   # "void (" = the function-pointer typedef
    names = {n for n in names if not n.startswith("__")}
    assert {"config_init", "config_destory", "score_chain", "kmer_count", "snp_phase", "snp_valid", "lgspolish",
            "polishresult_destory", "np1_batch_score_chain", "calgs"} <= names
    L = C.CDLL(nat.LIB_PATH)
    for n in sorted(names):
        assert hasattr(L, n), "nextpolish1.so does not export %s" % n
    assert hasattr(C.CDLL(os.path.join(ROOT, "nextpolish_amd", "lib", "calgs.so")), "calgs")


def test_calgs(tmp_path):
    fa = tmp_path / "a.fa"
    fa.write_text(">x\nACGT\nAC\n>y desc\nGGGGG\n")
    fq = tmp_path / "b.fq.gz"
    with gzip.open(str(fq), "wt") as f:
        f.write("@r1\nACGTA\n+\n@@@@@\n@r2\nAC\n+r2\n>>\n")
    assert nat.lib().calgs(str(fa).encode()) == 11
    assert nat.lib().calgs(str(fq).encode()) == 7
    out = subprocess.run([os.path.join(ROOT, "nextpolish_amd", "bin", "calgs"), str(fa)], stdout=subprocess.PIPE).stdout
    assert out == b"genome size: 11 bp\n"
    ref = os.path.join(ROOT, "oracle", "_ref", "calgs")
    if os.path.exists(ref):
        for p in (fa, fq):
            assert subprocess.run([ref, str(p)], stdout=subprocess.PIPE).stdout == \
                subprocess.run([os.path.join(ROOT, "nextpolish_amd", "bin", "calgs"), str(p)], stdout=subprocess.PIPE).stdout


def test_no_gpu_fails_loudly():
    """Without a HIP device the product path must refuse to work (no CPU fallback)."""
    if nat.lib().np1_device_count() > 0:
        pytest.skip("a GPU is present")
    from nextpolish_amd.device import Context
    with pytest.raises(RuntimeError, match="no HIP device"):
        Context(0)
    assert not nat.lib().np1_ctx_create(0)
    assert b"no HIP device" in nat.lib().np1_last_error()
