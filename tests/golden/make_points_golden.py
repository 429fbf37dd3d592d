#!/usr/bin/env python
"""Golden vectors of the -debug change list (PolishPoint, source/lib/contig.c:743-797) for tasks 2, 3 and 4, from the COMPILED
REFERENCE's shared library (oracle/_ref/nextpolish1.so) called the way source/lib/nextpolish1.py:181-189 calls it with
trace_polish_open = 1:
  * the real-mapper fixtures under tests/golden/real/,
  * seeded synthetic workloads (the test rebuilds them from the parameters below).
Runs in the build container only (needs oracle/_ref).  Output: tests/golden/points_golden.json -- per contig the number of
points, the md5 of their compact text form and the first 12 of them (for a readable failure)."""
import ctypes as C
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nextpolish_amd import _native as nat  # noqa: E402
import snpphase_gen  # noqa: E402

REAL = os.path.join(HERE, "real")
SYNTH = [dict(lens=[5000 + 733 * k, 900 + 17 * k], depth=[25, 8, 60][k % 3], seed=9100 + k, with_qual=1, weird_rate=0.02, softclip_rate=0.05,
              draft_lower=[0.05, 0.15, 0.3][k % 3], read_indel=[0.002, 0.01][k % 2]) for k in range(3)]
SYNTH3 = [dict(seed=9300 + k, lens=[3000 + 211 * k, 800], sr_depth=[40, 12][k % 2], lr_depth=[25, 10][k % 2], het=[0.01, 0.03][k % 2], het_indel=[0.002, 0.0][k % 2],
               draft_err=0.005, lower=0.05, sr_holes=k % 2, lr_err=0.05, lr_len=1500) for k in range(2)]


def bind(path):
    L = C.CDLL(path)
    L.config_init.restype = C.POINTER(nat.Configure)
    L.config_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    for f in ("score_chain", "kmer_count", "snp_valid", "snp_phase"):
        getattr(L, f).restype = C.POINTER(nat.PolishResult)
        getattr(L, f).argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    return L


def points_of(r):
    return [[r.contents.data[k].pos, r.contents.data[k].index, r.contents.data[k].curbase.decode(), r.contents.data[k].base.decode()]
            for k in range(r.contents.datalength)]


def digest_points(pts):
    text = ";".join("%d,%d,%s,%s" % tuple(p) for p in pts)
    return {"n": len(pts), "md5": hashlib.md5(text.encode()).hexdigest(), "head": pts[:12]}


def trace(L, task, fa, sr, lr, names):
    cfg = L.config_init(fa.encode(), sr.encode(), lr.encode() if lr else None)
    cfg.contents.trace_polish_open = 1
    out = {}
    for n in names:
        r = getattr(L, task)(n.encode(), cfg)
        d = digest_points(points_of(r))
        d["seq_md5"] = hashlib.md5(C.string_at(r.contents.contig)).hexdigest()
        out[n] = d
    return out


def fai_names(fa):
    return [line.split("\t")[0] for line in open(fa + ".fai")]


def main():
    L = bind(os.path.join(ROOT, "oracle", "_ref", "nextpolish1.so"))
    gold = {"real": {}, "synth": [], "synth3": []}
    for tag, fa, sr, lr, tasks in (("r1.slice", "r1.fa", "r1.slice.bam", None, ("kmer_count", "snp_valid")),
                                   ("sgs.s30", "g.fa", "sgs.s30.bam", None, ("kmer_count", "snp_valid")),
                                   ("s30+ont", "g.fa", "sgs.s30.bam", "lgs.sort.bam", ("snp_phase",)),
                                   ("s30+hifi", "g.fa", "sgs.s30.bam", "hifi.sort.bam", ("snp_phase",))):
        faf = os.path.join(REAL, fa)
        e = {"fasta": fa, "sr": sr, "lr": lr}
        for t in tasks:
            e[t] = trace(L, t, faf, os.path.join(REAL, sr), os.path.join(REAL, lr) if lr else None, fai_names(faf))
        gold["real"][tag] = e
    td = tempfile.mkdtemp()
    fa, bam, lbam = os.path.join(td, "s.fa"), os.path.join(td, "s.bam"), os.path.join(td, "l.bam")
    for kw in SYNTH:
        kw2 = dict(kw)
        lens = kw2.pop("lens")
        st = nat.Stream.synth(lens, **kw2)
        st.write_files(fa, bam)
        e = {"params": kw}
        for t in ("kmer_count", "snp_valid"):
            got = trace(L, t, fa, bam, None, st.names)
            e[t] = [got[n] for n in st.names]
        gold["synth"].append(e)
    for kw in SYNTH3:
        ctgs, srs, lrs = snpphase_gen.make_case(**kw)
        s, l = nat.Stream.from_reads(ctgs, srs), nat.Stream.from_reads(ctgs, lrs)
        s.write_files(fa, bam)
        l.write_files(os.path.join(td, "l.fa"), lbam)
        got = trace(L, "snp_phase", fa, bam, lbam, [n for n, _ in ctgs])
        gold["synth3"].append({"params": kw, "snp_phase": [got[n] for n, _ in ctgs]})
    json.dump(gold, open(os.path.join(HERE, "points_golden.json"), "w"), indent=0, sort_keys=True)
    print("wrote points_golden.json:", {k: len(v) for k, v in gold.items()},
          {t: sum(d["n"] for d in e[t].values()) for e in gold["real"].values() for t in e if isinstance(e[t], dict)})


if __name__ == "__main__":
    main()
