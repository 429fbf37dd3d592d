"""kmer_count / snp_valid with the reference's region iterator replayed (np1_replay.h; reference: source/lib/contig.c:982-1043 under
source/lib/kmercount.c:196-217 and snpvalid.c:3-36) on EVERY product path: thinly covered contigs of several 16 kb index windows, where
"records in file order" and the reference part ways, and deep ones, where the first loop leaves through the max_count_kmer break
(DESIGN.md section 3).  Goldens: the compiled reference's output (tests/golden/make_replay_golden.py).
CPU: the host model with the replay.  GPU: batch API, drop-in symbols, `nextpolish1 kmercount|snpvalid` and `nextpolish1.py -t 2|-t 4`
with the DEFAULT (device-side) ingest and with the host loader."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import pytest

from nextpolish_amd import _native as nat
from test_oracle import deep_multiwindow_stream, thin_multiwindow_stream

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "replay_golden.json")))
KEYS = sorted(GOLD["kmercount"])
EXE = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")


def digest(s):
    return {"len": len(s), "md5": hashlib.md5(s.encode()).hexdigest()}


def files(key, tmp_path):
    kind, seed = key[:4], int(key[4:])
    st, level = (thin_multiwindow_stream if kind == "thin" else deep_multiwindow_stream)(seed)
    fa, bam = str(tmp_path / ("%s.fa" % key)), str(tmp_path / ("%s.bam" % key))
    st.write_files(fa, bam, level)
    return fa, bam


@pytest.mark.parametrize("key", KEYS)
def test_host_model_with_replay_equals_reference_golden(key, tmp_path):
    import model_binding as mb
    fa, bam = files(key, tmp_path)
    cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
    s2 = nat.Stream.load(fa, bam, with_qual=True)
    got = mb.kmer_count_replay(s2, cfgp.contents, bam)
    assert {n: digest(x) for n, x in zip(s2.names, got)} == GOLD["kmercount"][key]
    if key in GOLD["snpvalid"]:
        got = mb.snp_valid_replay(s2, cfgp.contents, bam)
        assert got is not None and {n: digest(x) for n, x in zip(s2.names, got)} == GOLD["snpvalid"][key]
    nat.lib().config_destory(cfgp)


def test_oracle_from_files_equals_reference_golden(tmp_path):
    """oracle_binding.from_files = the oracle in replay mode, what the GPU tests compare file-based paths with"""
    import oracle_binding as ob
    for key in ("thin44", "deep3"):
        fa, bam = files(key, tmp_path)
        cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
        ocfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
        nat.lib().config_destory(cfgp)
        assert {n: digest(s) for n, s in ob.from_files("kmer_count", fa, bam, ocfg).items()} == GOLD["kmercount"][key]
        assert {n: digest(s) for n, s in ob.from_files("snp_valid", fa, bam, ocfg).items()} == GOLD["snpvalid"][key]


@pytest.mark.gpu
def test_gpu_batch_with_replay_equals_reference_golden(tmp_path):
    from nextpolish_amd import device
    ctx = device.Context(0)
    try:
        for key in KEYS:
            fa, bam = files(key, tmp_path)
            cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
            s2 = nat.Stream.load(fa, bam, with_qual=True)
            for task in ("kmercount", "snpvalid"):
                if key not in GOLD[task]:
                    continue
                b = ctx.upload(s2)
                b.enable_replay(bam)
                (b.kmer_count if task == "kmercount" else b.snp_valid)(cfgp.contents)
                got = b.results()
                b.close()
                assert {n: digest(x) for n, x in zip(s2.names, got)} == GOLD[task][key], (task, key)
            nat.lib().config_destory(cfgp)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["thin44", "deep5"])
def test_gpu_dropin_symbols_with_replay(key, tmp_path):
    """kmer_count(tigname, cfg) / snp_valid(tigname, cfg) as source/lib/nextpolish1.py:93-98,181-189 calls them: the replay is the default"""
    L = nat.lib()
    fa, bam = files(key, tmp_path)
    cfg = L.config_init(fa.encode(), bam.encode(), None)
    for task, fn in (("kmercount", L.kmer_count), ("snpvalid", L.snp_valid)):
        fn.restype = C.POINTER(nat.PolishResult)
        fn.argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
        for n, want in GOLD[task][key].items():
            r = fn(n.encode(), cfg)
            assert digest(C.string_at(r.contents.contig).decode()) == want, (task, n)
            L.polishresult_destory(r)
    L.config_destory(cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("ingest", ["device", "host"])
def test_gpu_cli_with_replay(ingest, tmp_path):
    """`nextpolish1 kmercount|snpvalid fa bam` (source/lib/main.c:7-8): batches of the default device-side ingest bring the records'
    virtual offsets down with them, batches of the host loader keep them -- both replay the iterator and equal the reference"""
    from conftest import parse_cli_fasta
    env = dict(os.environ)
    env.pop("NP1_INGEST", None)
    env.pop("NP1_ITER_REPLAY", None)
    if ingest == "host":
        env["NP1_INGEST"] = "host"
    for key in KEYS:
        fa, bam = files(key, tmp_path)
        for task in ("kmercount", "snpvalid"):
            if key not in GOLD[task]:
                continue
            p = subprocess.run([EXE, task, fa, bam], capture_output=True, text=True, env=env)
            assert p.returncode == 0, p.stderr
            assert {n: digest(x) for n, x in parse_cli_fasta(p.stdout).items()} == GOLD[task][key], (task, key, ingest)


@pytest.mark.gpu
def test_gpu_python_caller_with_replay(tmp_path):
    """nextpolish_amd/nextpolish1.py -t 2 / -t 4 (mirror of source/lib/nextpolish1.py:196-231), default ingest, small batches"""
    for key in ("thin44", "thin102", "deep3"):
        fa, bam = files(key, tmp_path)
        for t, task in (("2", "kmercount"), ("4", "snpvalid")):
            if key not in GOLD[task]:
                continue
            out = str(tmp_path / ("o_%s_%s.fa" % (key, t)))
            p = subprocess.run([sys.executable, os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py"), "-g", fa, "-t", t, "-s", bam, "-o", out],
                               capture_output=True, text=True)
            assert p.returncode == 0, p.stderr
            recs = open(out).read().strip().split("\n")
            got = {recs[k].split()[0][1:].rsplit("_np", 1)[0]: recs[k + 1] for k in range(0, len(recs), 2)}
            assert {n: digest(s) for n, s in got.items()} == GOLD[task][key], (task, key)


@pytest.mark.gpu
def test_gpu_pipe_from_files_equals_oracle_in_replay_mode(tmp_path):
    """np1_pipe_run_files with several batches per file and a subset of the contigs: equal to the oracle with the iterator replayed,
    contig by contig, for both tasks and both ingest paths"""
    import oracle_binding as ob
    from nextpolish_amd.device import Pipe
    pipe = Pipe(0, lanes=2)
    try:
        for key in ("thin100", "deep5"):
            fa, bam = files(key, tmp_path)
            cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
            ocfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
            for task, name in ((2, "kmer_count"), (4, "snp_valid")):
                want = ob.from_files(name, fa, bam, ocfg)
                if any(v is None for v in want.values()):
                    continue
                for ingest in (None, "host"):
                    old = os.environ.pop("NP1_INGEST", None)
                    if ingest:
                        os.environ["NP1_INGEST"] = ingest
                    try:
                        out = pipe.run_files(fa, bam, batch_bp=20000, cfg=cfgp.contents, task=task)
                    finally:
                        os.environ.pop("NP1_INGEST", None)
                        if old is not None:
                            os.environ["NP1_INGEST"] = old
                    assert dict(out) == want, (key, name, ingest)
            nat.lib().config_destory(cfgp)
    finally:
        pipe.close()
