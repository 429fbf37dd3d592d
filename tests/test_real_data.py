"""Real-mapper inputs (BASELINE config 1 = the reference's bundled test data; SURVEY.md 8c goldens (1) and (3)).

tests/golden/real/ holds the bundled draft and the BAMs that the reference's vendored bwa / minimap2 / samtools wrote for the
bundled reads (tests/golden/make_real_golden.py), with the outputs of the compiled reference.  Nothing here was written by this
repository's own BAM writer: secondary + supplementary records, MC/MD/NM/SA/XA aux fields, real insert sizes, samtools' BGZF.

CPU tests: this repository's BAM reader + the oracle / the host models against the goldens (and against oracle/_ref when present).
GPU tests (-m gpu): the same files through lib/nextpolish1.so, lib/nextpolish2.so (drop-in symbols) and the CLIs."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import pytest

from nextpolish_amd import _native as nat
from nextpolish_amd import device as npdev
import oracle_binding as ob
import model_binding as mb
import ref2_binding as rb
from conftest import ROOT, ref_binary, run_ref, parse_cli_fasta

HERE = os.path.dirname(os.path.abspath(__file__))
REAL = os.path.join(HERE, "golden", "real")
GOLD = json.load(open(os.path.join(REAL, "real_golden.json")))
MODEL2_SO = os.path.join(HERE, "model", "libnp2_model.so")
PRODUCT2_SO = os.path.join(ROOT, "nextpolish_amd", "lib", "nextpolish2.so")
SR = sorted(GOLD["sr"])
LR = sorted(GOLD["lr"])


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


def digest(s):
    return {"len": len(s), "md5": md5(s)}


def sr_files(tag):
    g = GOLD["sr"][tag]
    return g, os.path.join(REAL, g["fasta"]), os.path.join(REAL, g["bam"])


# ------------------------------------------------------------------------------------------------- CPU: path A


@pytest.mark.parametrize("tag", SR)
def test_oracle_on_real_bwa_alignments(tag):
    g, fa, bam = sr_files(tag)
    st = nat.Stream.load(fa, bam, with_qual=True)
    cfg = ob.default_config(read_tlen=g["read_tlen"], read_len=g["read_len"])
    for i, n in enumerate(st.names):
        assert digest(ob.score_chain(st, i)) == g["score_chain"][n], "score_chain %s %s" % (tag, n)
        if "kmer_count" in g:
            assert digest(ob.kmer_count(st, i, cfg)) == g["kmer_count"][n], "kmer_count %s %s" % (tag, n)


@pytest.mark.parametrize("tag", SR)
def test_host_model_on_real_bwa_alignments(tag):
    """The per-lane kernel bodies run by the host executor (tests/model) on the real records."""
    g, fa, bam = sr_files(tag)
    st = nat.Stream.load(fa, bam, with_qual=True)
    got = mb.score_chain(st, fused=True)
    for i, n in enumerate(st.names):
        assert digest(got[i]) == g["score_chain"][n], "score_chain %s %s" % (tag, n)
    if "kmer_count" in g:
        cfg = nat.default_config()
        cfg.read_tlen, cfg.read_len = g["read_tlen"], g["read_len"]
        got = mb.kmer_count(st, cfg)
        for i, n in enumerate(st.names):
            assert digest(got[i]) == g["kmer_count"][n], "kmer_count %s %s" % (tag, n)


@pytest.mark.parametrize("tag", SR)
def test_insert_size_probe_on_real_pairs(tag):
    """config_init / bam_tlen (source/lib/config.c:80-101) on real insert-size tails."""
    g, fa, bam = sr_files(tag)
    cfg = nat.lib().config_init(fa.encode(), bam.encode(), None)
    assert (cfg.contents.read_tlen, cfg.contents.read_len) == (g["read_tlen"], g["read_len"])
    nat.lib().config_destory(cfg)


@pytest.mark.skipif(ref_binary() is None, reason="oracle/_ref not built")
@pytest.mark.parametrize("tag", SR)
def test_real_goldens_still_match_compiled_reference(tag):
    g, fa, bam = sr_files(tag)
    assert {n: digest(s) for n, s in run_ref("scorechain", fa, bam).items()} == g["score_chain"]
    if "kmer_count" in g:
        assert {n: digest(s) for n, s in run_ref("kmercount", fa, bam).items()} == g["kmer_count"]


# ------------------------------------------------------------------------------------------------- CPU: path B


def run_polish2(so_path, fa, bam, read_type, split, tmp_path):
    fofn = str(tmp_path / "reads.fofn")
    open(fofn, "w").write(bam + "\n")
    code = ("import sys, json; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(%r); "
            "print(json.dumps(rb.polish(L, %r, %r, read_type=%d, split=%d)))" % (HERE, so_path, fa, fofn, read_type, split))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return json.loads(p.stdout.strip().splitlines()[-1])


def check_lr(tag, so, tmp_path):
    g = GOLD["lr"][tag]
    got = run_polish2(so, os.path.join(REAL, g["fasta"]), os.path.join(REAL, g["bam"]), g["read_type"], g["split"], tmp_path)
    assert sorted(got) == sorted(g["expected"])
    for n, pieces in got.items():
        assert [{"len": l, "md5": md5(s)} for s, l in pieces] == g["expected"][n], "%s %s" % (tag, n)


@pytest.mark.parametrize("tag", LR)
def test_long_read_model_on_real_minimap2_alignments(tag, tmp_path):
    if not os.path.exists(MODEL2_SO):
        subprocess.run(["make", "-C", os.path.join(HERE, "model"), "libnp2_model.so"], check=True, capture_output=True)
    check_lr(tag, MODEL2_SO, tmp_path)


@pytest.mark.skipif(not rb.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("tag", LR)
def test_long_read_goldens_still_match_compiled_reference(tag, tmp_path):
    check_lr(tag, rb.REF_SO, tmp_path)


# ------------------------------------------------------------------------------------------------- GPU


@pytest.mark.gpu
@pytest.mark.parametrize("tag", SR)
def test_gpu_dropin_symbols_on_real_bwa_alignments(tag):
    """config_init -> score_chain / kmer_count per contig exactly like source/lib/nextpolish1.py:181-189,219, IN THIS PROCESS: a
    long-lived worker calling the drop-in symbols contig after contig is how the reference's caller uses the library.  (Round 3 moved
    this test into a subprocess after an intermittent SIGABRT of the one-process suite; round 4 found the cause -- DESIGN.md section
    12 -- and it runs here again.)"""
    dropin_symbols_body(tag)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", SR[:1])
def test_gpu_dropin_symbols_on_real_bwa_alignments_in_a_worker_process(tag):
    """The same calls in a worker process of their own, as the reference's caller runs them (nextpolish1.py:148-179 forks its workers
    before the library is touched)."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nimport test_real_data as t\nt.dropin_symbols_body(%r)\nprint('dropin ok')\n"
            % (ROOT, HERE, tag))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "dropin ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def dropin_symbols_body(tag):
    g, fa, bam = sr_files(tag)
    L = nat.lib()
    cfg = L.config_init(fa.encode(), bam.encode(), None)
    assert (cfg.contents.read_tlen, cfg.contents.read_len) == (g["read_tlen"], g["read_len"])
    cfg.contents.trace_polish_open = 1 if "points" in g else 0
    for n in sorted(g["score_chain"]):
        r = L.score_chain(n.encode(), cfg)
        seq = C.string_at(r.contents.contig).decode()
        assert r.contents.length == len(seq) and digest(seq) == g["score_chain"][n], "score_chain %s %s" % (tag, n)
        if "points" in g:
            pts = [[r.contents.data[k].pos, r.contents.data[k].index, r.contents.data[k].curbase.decode(),
                    r.contents.data[k].base.decode()] for k in range(r.contents.datalength)]
            assert pts == g["points"][n], "PolishPoint list %s %s" % (tag, n)
        L.polishresult_destory(r)
    cfg.contents.trace_polish_open = 0
    if "kmer_count" in g:
        for n in sorted(g["kmer_count"]):
            r = L.kmer_count(n.encode(), cfg)
            seq = C.string_at(r.contents.contig).decode()
            assert digest(seq) == g["kmer_count"][n], "kmer_count %s %s" % (tag, n)
            L.polishresult_destory(r)
    L.config_destory(cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", SR)
def test_gpu_cli_on_real_bwa_alignments(tag):
    g, fa, bam = sr_files(tag)
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    for task, key in (("scorechain", "score_chain"), ("kmercount", "kmer_count")):
        if key not in g:
            continue
        out = subprocess.run([exe, task, fa, bam], stdout=subprocess.PIPE, check=True).stdout.decode()
        assert {n: digest(s) for n, s in parse_cli_fasta(out).items()} == g[key], "%s %s" % (task, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", SR)
def test_gpu_python_caller_on_real_bwa_alignments(tag, tmp_path):
    """nextpolish_amd/nextpolish1.py (mirror of source/lib/nextpolish1.py), batched GPU path, task 1 (and 2)."""
    g, fa, bam = sr_files(tag)
    exe = [sys.executable, os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py")]
    for task, key in ((1, "score_chain"), (2, "kmer_count")):
        if key not in g:
            continue
        out = str(tmp_path / ("t%d.fa" % task))
        subprocess.run(exe + ["-g", fa, "-t", str(task), "-p", "1", "-s", bam, "-o", out], check=True)
        got, name = {}, None
        for line in open(out):
            if line.startswith(">"):
                name = line[1:].split()[0]
            else:
                got[name] = digest(line.strip())
        suffix = "_np%d" % task
        want = {}
        for n, d in g[key].items():      # naming rule of source/lib/nextpolish1.py:228-229
            toks = n.split("_")
            want[(n + str(task)) if (len(toks) > 1 and toks[-1].startswith("np")) else n + suffix] = d
        assert got == want, "%s task %d" % (tag, task)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", LR)
def test_gpu_long_read_library_on_real_minimap2_alignments(tag, tmp_path):
    check_lr(tag, PRODUCT2_SO, tmp_path)


@pytest.mark.gpu
def test_gpu_long_read_cli_on_real_minimap2_alignments(tmp_path):
    g = GOLD["lr"]["lgs.sort.rt1.split0"]
    fofn = str(tmp_path / "lgs.fofn")
    open(fofn, "w").write(os.path.join(REAL, g["bam"]) + "\n")
    p = subprocess.run([os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish2"), os.path.join(REAL, g["fasta"]), fofn],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().split("\n")
    got = {lines[i][1:].split()[0]: {"len": len(lines[i + 1]), "md5": md5(lines[i + 1])} for i in range(0, len(lines), 2)}
    assert got == {n + "_lgs": v[0] for n, v in g["expected"].items()}
