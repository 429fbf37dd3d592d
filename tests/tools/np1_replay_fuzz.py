#!/usr/bin/env python
"""Fuzz of kmer_count / snp_valid where the reference's region iterator decides the result: the compiled reference
against the oracle in its record-stream mode, the oracle with the iterator replayed (oracle_binding.Geometry) and the host model with the
product-side replay (np1_replay.h: chunk lists, re-use, saved offsets, the max_count_kmer break) for both tasks.
Two families: thinly covered contigs of several 16 kb index windows (tests/test_oracle.thin_multiwindow_stream) and deep ones
(80-250x: the first loop of nearly every part leaves through the break).
usage: np1_replay_fuzz.py [--deep] SEED [SEED ...]   (needs oracle/_ref; CPU only)"""
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
from test_oracle import deep_multiwindow_stream, thin_multiwindow_stream  # noqa: E402


def main():
    args = sys.argv[1:]
    deep = "--deep" in args
    seeds = [int(x) for x in args if x != "--deep"]
    d = tempfile.mkdtemp(prefix="np1replayfz_")
    tot = dict(n=0, record_stream_wrong=0, oracle_replay_wrong=0, model_replay_wrong=0, ref_crashed=0, undefined=0)
    for seed in seeds:
        st, level = (deep_multiwindow_stream if deep else thin_multiwindow_stream)(seed)
        fa, bam = d + "/s.fa", d + "/s.bam"
        st.write_files(fa, bam, level)
        cfgp = nat.lib().config_init(fa.encode(), bam.encode(), None)
        ocfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
        s2 = nat.Stream.load(fa, bam, with_qual=True)
        geom = ob.Geometry(s2, bam)
        models = {"kmercount": mb.kmer_count_replay(s2, cfgp.contents, bam), "snpvalid": mb.snp_valid_replay(s2, cfgp.contents, bam)}
        nat.lib().config_destory(cfgp)
        for cmd, fn in (("kmercount", ob.kmer_count), ("snpvalid", ob.snp_valid)):
            p = subprocess.run([ref_binary(), cmd, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            if p.returncode != 0:
                tot["ref_crashed"] += 1
                continue
            ref = parse_cli_fasta(p.stdout.decode())
            model = models[cmd]
            for i, n in enumerate(s2.names):
                plain, replay = fn(s2, i, ocfg), fn(s2, i, ocfg, geom)
                tot["n"] += 1
                if plain is not None and plain != ref[n]:
                    tot["record_stream_wrong"] += 1
                    print("records in file order != reference:", cmd, seed, n)
                if replay is None:
                    tot["undefined"] += 1
                elif replay != ref[n]:
                    tot["oracle_replay_wrong"] += 1
                    print("oracle with replay != reference:", cmd, seed, n)
                if model is not None and model[i] != ref[n]:
                    tot["model_replay_wrong"] += 1
                    print("model with replay != reference:", cmd, seed, n)
    tot["model_revotes"] = mb.replay_revotes()
    tot["model_breaks"] = mb.replay_breaks()
    print(tot)


if __name__ == "__main__":
    main()
