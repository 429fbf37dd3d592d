"""kmer_count with the reference's region iterator replayed (np1_replay.h + np1_batch_enable_replay; reference: source/lib/contig.c:982-1043):
thinly covered contigs of several 16 kb index windows, where "records in file order" and the reference part ways.
Goldens: the compiled reference's output (tests/golden/make_replay_golden.py).  CPU: the host model with the replay; GPU: the batch
API and the drop-in symbol with NP1_ITER_REPLAY=1."""
import ctypes as C
import hashlib
import json
import os

import pytest

from nextpolish_amd import _native as nat
from test_oracle import thin_multiwindow_stream

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "replay_golden.json")))


def digest(s):
    return {"len": len(s), "md5": hashlib.md5(s.encode()).hexdigest()}


def files(seed, tmp_path):
    st, level = thin_multiwindow_stream(seed)
    fa, bam = str(tmp_path / ("z%d.fa" % seed)), str(tmp_path / ("z%d.bam" % seed))
    st.write_files(fa, bam, level)
    return fa, bam


@pytest.mark.parametrize("seed", sorted(GOLD, key=int))
def test_host_model_with_replay_equals_reference_golden(seed, tmp_path):
    import model_binding as mb
    fa, bam = files(int(seed), tmp_path)
    cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
    s2 = nat.Stream.load(fa, bam, with_qual=True)
    got = mb.kmer_count_replay(s2, cfgp.contents, bam)
    nat.lib().config_destory(cfgp)
    assert {n: digest(x) for n, x in zip(s2.names, got)} == GOLD[seed]


@pytest.mark.gpu
def test_gpu_batch_with_replay_equals_reference_golden(tmp_path):
    from nextpolish_amd import device
    ctx = device.Context(0)
    try:
        for seed in sorted(GOLD, key=int):
            fa, bam = files(int(seed), tmp_path)
            cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
            s2 = nat.Stream.load(fa, bam, with_qual=True)
            b = ctx.upload(s2)
            b.enable_replay(bam)
            b.kmer_count(cfgp.contents)
            got = b.results()
            b.close()
            nat.lib().config_destory(cfgp)
            assert {n: digest(x) for n, x in zip(s2.names, got)} == GOLD[seed], seed
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_dropin_symbol_with_replay(tmp_path, monkeypatch):
    monkeypatch.setenv("NP1_ITER_REPLAY", "1")
    L = nat.lib()
    L.kmer_count.restype = C.POINTER(nat.PolishResult)
    L.kmer_count.argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    fa, bam = files(44, tmp_path)
    cfg = L.config_init(fa.encode(), bam.encode(), None)
    for n, want in GOLD["44"].items():
        r = L.kmer_count(n.encode(), cfg)
        assert digest(C.string_at(r.contents.contig).decode()) == want, n
        L.polishresult_destory(r)
    L.config_destory(cfg)


@pytest.mark.gpu
def test_gpu_cli_on_the_host_loader_with_replay(tmp_path):
    """`nextpolish1 kmercount` with NP1_INGEST=host: batches decoded by the host loader keep the records' virtual offsets and replay the
    iterator too (np1_pipe.cpp; batches of the default device-side ingest take the records in file order)"""
    import subprocess
    from conftest import parse_cli_fasta
    exe = os.path.join(os.path.dirname(HERE), "nextpolish_amd", "bin", "nextpolish1")
    fa, bam = files(44, tmp_path)
    p = subprocess.run([exe, "kmercount", fa, bam], capture_output=True, text=True, env=dict(os.environ, NP1_INGEST="host"))
    assert p.returncode == 0, p.stderr
    assert {n: digest(x) for n, x in parse_cli_fasta(p.stdout).items()} == GOLD["44"]
