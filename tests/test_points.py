"""The -debug change list (PolishPoint; reference: source/lib/contig.c:743-797, printed by source/lib/nextpolish1.py:230-231) of
tasks 2, 3 and 4.  Every task of the reference ends in contig_get_contig (kmercount.c:121, snpphase.c:129, snpvalid.c:30), which
builds the list when trace_polish_open is set; goldens: tests/golden/points_golden.json, made by the compiled reference's shared
library (tests/golden/make_points_golden.py)."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from nextpolish_amd import _native as nat  # noqa: E402
import snpphase_gen  # noqa: E402

REAL = os.path.join(HERE, "golden", "real")
GOLD = json.load(open(os.path.join(HERE, "golden", "points_golden.json")))


def digest_points(pts):
    text = ";".join("%d,%d,%s,%s" % tuple(p) for p in pts)
    return {"n": len(pts), "md5": hashlib.md5(text.encode()).hexdigest(), "head": [list(p) for p in pts[:12]]}


def product():
    L = nat.lib()
    for f in ("kmer_count", "snp_valid", "snp_phase"):
        getattr(L, f).restype = C.POINTER(nat.PolishResult)
        getattr(L, f).argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    return L


def trace(L, task, fa, sr, lr, names):
    cfg = L.config_init(fa.encode(), sr.encode(), lr.encode() if lr else None)
    cfg.contents.trace_polish_open = 1
    out = {}
    for n in names:
        r = getattr(L, task)(n.encode(), cfg)
        pts = [[r.contents.data[k].pos, r.contents.data[k].index, r.contents.data[k].curbase.decode(), r.contents.data[k].base.decode()]
               for k in range(r.contents.datalength)]
        d = digest_points(pts)
        d["seq_md5"] = hashlib.md5(C.string_at(r.contents.contig)).hexdigest()
        out[n] = d
        L.polishresult_destory(r)
    L.config_destory(cfg)
    return out


def test_golden_file_covers_all_three_tasks_with_nonempty_lists():
    """(CPU) the fixture itself: every task has at least one contig whose list is not empty, so an empty list cannot pass."""
    n = {"kmer_count": 0, "snp_valid": 0, "snp_phase": 0}
    for e in GOLD["synth"]:
        for t in ("kmer_count", "snp_valid"):
            n[t] += sum(d["n"] for d in e[t])
    for e in GOLD["synth3"]:
        n["snp_phase"] += sum(d["n"] for d in e["snp_phase"])
    for e in GOLD["real"].values():
        for t in n:
            if t in e:
                n[t] += sum(d["n"] for d in e[t].values())
    assert all(v > 50 for v in n.values()), n


@pytest.mark.parametrize("k", range(len(GOLD["synth"])))
def test_oracle_change_lists_equal_the_compiled_reference_on_synthetic_workloads(k):
    """(CPU) the oracle's restatement of the list (np1_oracle.c: get_contig) against the goldens of the compiled reference, tasks 2 and 4"""
    import oracle_binding as ob
    e = GOLD["synth"][k]
    kw = dict(e["params"])
    lens = kw.pop("lens")
    st = nat.Stream.synth(lens, **kw)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fa, bam = os.path.join(td, "s.fa"), os.path.join(td, "s.bam")
        st.write_files(fa, bam)
        cp = nat.lib().config_init(fa.encode(), bam.encode(), None)
        ocfg = ob.default_config(read_tlen=cp.contents.read_tlen, read_len=cp.contents.read_len, trace_polish_open=1)
        nat.lib().config_destory(cp)
        for task, fn in (("kmer_count", ob.kmer_count), ("snp_valid", ob.snp_valid)):
            for i, n in enumerate(st.names):          # contig by contig from the files, the iterator replayed like the reference on files
                st2 = nat.Stream.load(fa, bam, names=[n], with_qual=True)
                seq = fn(st2, 0, ocfg, ob.Geometry(st2, bam))
                got = digest_points(ob.last_points())      # the list of this thread's last call
                st2.close()
                assert hashlib.md5(seq.encode()).hexdigest() == e[task][i]["seq_md5"]
                assert (got["n"], got["md5"]) == (e[task][i]["n"], e[task][i]["md5"]), (task, n, got["head"], e[task][i]["head"])


@pytest.mark.parametrize("k", range(len(GOLD["synth3"])))
def test_oracle_change_lists_of_snp_phase_equal_the_compiled_reference(k):
    """(CPU) task 3: the oracle's list against the reference's on the seeded diploid workloads"""
    import oracle_binding as ob
    import tempfile
    e = GOLD["synth3"][k]
    ctgs, srs, lrs = snpphase_gen.make_case(**e["params"])
    s, l = nat.Stream.from_reads(ctgs, srs), nat.Stream.from_reads(ctgs, lrs)
    with tempfile.TemporaryDirectory() as td:
        fa, bam = os.path.join(td, "s.fa"), os.path.join(td, "s.bam")
        s.write_files(fa, bam)
        cp = nat.lib().config_init(fa.encode(), bam.encode(), None)
        ocfg = ob.default_config(read_tlen=cp.contents.read_tlen, read_len=cp.contents.read_len, trace_polish_open=1)
        nat.lib().config_destory(cp)
    for i in range(len(ctgs)):
        seq = ob.snp_phase(s, l, i, ocfg)
        got = digest_points(ob.last_points())
        assert hashlib.md5(seq.encode()).hexdigest() == e["snp_phase"][i]["seq_md5"]
        assert (got["n"], got["md5"]) == (e["snp_phase"][i]["n"], e["snp_phase"][i]["md5"]), (i, got["head"], e["snp_phase"][i]["head"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(GOLD["real"]))
def test_gpu_dropin_symbols_give_the_reference_change_lists_on_real_alignments(tag):
    e = GOLD["real"][tag]
    L = product()
    fa = os.path.join(REAL, e["fasta"])
    names = [line.split("\t")[0] for line in open(fa + ".fai")]
    for t in ("kmer_count", "snp_valid", "snp_phase"):
        if t in e:
            got = trace(L, t, fa, os.path.join(REAL, e["sr"]), os.path.join(REAL, e["lr"]) if e["lr"] else None, names)
            assert got == e[t], "%s %s" % (tag, t)


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(GOLD["synth"])))
def test_gpu_kmer_count_and_snp_valid_change_lists_on_synthetic_workloads(k, tmp_path):
    e = GOLD["synth"][k]
    kw = dict(e["params"])
    lens = kw.pop("lens")
    st = nat.Stream.synth(lens, **kw)
    fa, bam = str(tmp_path / "s.fa"), str(tmp_path / "s.bam")
    st.write_files(fa, bam)
    L = product()
    for t in ("kmer_count", "snp_valid"):
        got = trace(L, t, fa, bam, None, st.names)
        assert [got[n] for n in st.names] == e[t], t


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(GOLD["synth3"])))
def test_gpu_snp_phase_change_lists_on_synthetic_diploids(k, tmp_path):
    e = GOLD["synth3"][k]
    ctgs, srs, lrs = snpphase_gen.make_case(**e["params"])
    s, l = nat.Stream.from_reads(ctgs, srs), nat.Stream.from_reads(ctgs, lrs)
    fa, bam, lbam = str(tmp_path / "s.fa"), str(tmp_path / "s.bam"), str(tmp_path / "l.bam")
    s.write_files(fa, bam)
    l.write_files(str(tmp_path / "l.fa"), lbam)
    got = trace(product(), "snp_phase", fa, bam, lbam, [n for n, _ in ctgs])
    assert [got[n] for n, _ in ctgs] == e["snp_phase"]


@pytest.mark.gpu
def test_gpu_python_caller_prints_the_change_list_of_task_2_on_stderr(tmp_path):
    """nextpolish1.py -t 2 -debug: `<name> <pos> <index> <curbase> <base>` per point on stderr (source/lib/nextpolish1.py:230-231)."""
    e = GOLD["synth"][1]
    kw = dict(e["params"])
    lens = kw.pop("lens")
    st = nat.Stream.synth(lens, **kw)
    fa, bam = str(tmp_path / "s.fa"), str(tmp_path / "s.bam")
    st.write_files(fa, bam)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py"), "-g", fa, "-t", "2", "-s", bam, "-debug", "-o", str(tmp_path / "o.fa")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    per = {n: [] for n in st.names}
    for line in p.stderr.splitlines():
        f = line.split(" ")
        if len(f) == 5 and f[0] in per:
            per[f[0]].append([int(f[1]), int(f[2]), f[3], f[4]])
    for n, want in zip(st.names, e["kmer_count"]):
        got = digest_points(per[n])
        assert (got["n"], got["md5"]) == (want["n"], want["md5"]), n
