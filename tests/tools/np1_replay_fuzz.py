#!/usr/bin/env python
"""Fuzz of kmer_count / snp_valid on thinly covered contigs of several 16 kb index windows (tests/test_oracle.thin_multiwindow_stream):
the compiled reference against the oracle in its record-stream mode, the oracle with the region iterator replayed (oracle_binding.Geometry)
and the host model with the product-side replay (np1_replay.h).  usage: np1_replay_fuzz.py SEED [SEED ...]   (needs oracle/_ref; CPU only)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import model_binding as mb  # noqa: E402
import oracle_binding as ob  # noqa: E402
from conftest import parse_cli_fasta, ref_binary  # noqa: E402
from nextpolish_amd import _native as nat  # noqa: E402
from test_oracle import thin_multiwindow_stream  # noqa: E402

d = tempfile.mkdtemp(prefix="np1replayfz_")
tot = dict(n=0, record_stream_wrong=0, oracle_replay_wrong=0, model_replay_wrong=0)
for seed in [int(x) for x in sys.argv[1:]]:
    st, level = thin_multiwindow_stream(seed)
    fa, bam = d + "/s.fa", d + "/s.bam"
    st.write_files(fa, bam, level)
    cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
    ocfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
    s2 = nat.Stream.load(fa, bam, with_qual=True)
    geom = ob.Geometry(s2, bam)
    model = mb.kmer_count_replay(s2, cfgp.contents, bam)
    nat.lib().config_destory(cfgp)
    for cmd, fn in (("kmercount", ob.kmer_count), ("snpvalid", ob.snp_valid)):
        p = subprocess.run([ref_binary(), cmd, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        if p.returncode != 0:
            continue
        ref = parse_cli_fasta(p.stdout.decode())
        for i, n in enumerate(s2.names):
            plain, replay = fn(s2, i, ocfg), fn(s2, i, ocfg, geom)
            tot["n"] += 1
            tot["record_stream_wrong"] += plain is not None and plain != ref[n]
            if replay is not None and replay != ref[n]:
                tot["oracle_replay_wrong"] += 1
                print("oracle with replay != reference:", cmd, seed, n)
            if cmd == "kmercount" and model[i] != ref[n]:
                tot["model_replay_wrong"] += 1
                print("model with replay != reference:", seed, n)
print(tot)
