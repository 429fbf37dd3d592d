#!/usr/bin/env python
"""Real-mapper fixtures (SURVEY.md 8c goldens (1) and (3), BASELINE config 1): the reference's bundled test data
(source/test_data: raw.genome.fasta, sreads.R[12].fastq.gz, lreads.fasta.gz, hifi.fasta.gz) mapped with the mappers
the reference vendors (bwa mem, minimap2, samtools sort/fixmate/markdup) and polished by the COMPILED REFERENCE
(oracle/_ref/nextpolish1, nextpolish1.so, nextpolish2.so).  Runs in the build container only:

    # mappers: built in a scratch directory from the vendored sources, never inside the repo
    O=/tmp/npmap; cp -r /root/reference/source/lib $O/lib; cp -r /root/reference/source/util $O/util; chmod -R u+w $O
    (cd $O/lib/htslib && printf '#define HAVE_FSEEKO 1\n#define HAVE_DRAND48 1\n' > config.h && make lib-static CPPFLAGS+=-fPIC)
    (cd $O/util/minimap2 && make); (cd $O/util/bwa && make CFLAGS="-g -Wall -Wno-unused-function -O2 -fcommon")
    (cd $O/util/samtools && sed -i 's/ -lbz2 -llzma//' ../../lib/htslib/htslib_static.mk && make HTSDIR=../../lib/htslib samtools)
    NP_MAPPERS=$O/util python tests/golden/make_real_golden.py

What is committed under tests/golden/real/ is DATA: the draft (a data file of the reference's own test set), the BAM
(+BAI) files exactly as the mappers wrote them (every aux tag, secondary/supplementary records, real insert sizes,
samtools' BGZF), and real_golden.json with the reference's outputs (md5 + length per contig, the -debug PolishPoint
list of the subsampled case).  No reference source text is stored.

Files:
  g.fa                 the bundled draft (2 contigs, 51 kb + 60 kb)
  sgs.sort.bam         all bundled PE150 reads, bwa mem | view -F4 | fixmate -m | sort | markdup -r   (~150x; config 1)
  sgs.s30.bam          1 in 5 read pairs of it (samtools view -s), ~30x
  r1.fa                score_chain output of the reference on sgs.sort.bam (lowercase = low-confidence, the input of task 2)
  r1.slice.bam         the same reads re-mapped to r1.fa, records overlapping four regions (kmer_count round)
  lgs.sort.bam         bundled ONT reads, minimap2 -ax map-ont | sort  (secondary + supplementary + SA tags)
  hifi.sort.bam        bundled HiFi reads, minimap2 -ax asm20 | sort
"""
import ctypes as C
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import run_ref  # noqa: E402
import ref2_binding as rb  # noqa: E402
from nextpolish_amd import _native as nat  # noqa: E402  (struct layouts only; the library called below is the reference's)

T = "/root/reference/source/test_data"
OUT = os.path.join(HERE, "real")
SLICES = ["tig0000001_1:1-9000", "tig0000001_1:30000-36000", "tig0000002_1:1-500", "tig0000002_1:52000-60498"]


def sh(cmd, cwd):
    subprocess.run(cmd, shell=True, check=True, cwd=cwd, stderr=subprocess.DEVNULL, executable="/bin/bash")


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


def digest(d):
    return {n: {"len": len(s), "md5": md5(s)} for n, s in d.items()}


def ref_rates(fa, bam, rates):
    """score_chain of the reference's shared library for other values of indel_balance_factor_sgs (the caller mutates the
    Configure after config_init, source/lib/nextpolish1.py:102-133)."""
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "nextpolish1.so"))
    L.config_init.restype = C.POINTER(nat.Configure)
    L.config_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.score_chain.restype = C.POINTER(nat.PolishResult)
    L.score_chain.argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    cfg = L.config_init(fa.encode(), bam.encode(), None)
    out = {}
    for rate in rates:
        cfg.contents.indel_balance_factor_sgs = rate
        out[repr(rate)] = {}
        for line in open(fa + ".fai"):
            name = line.split("\t")[0]
            r = L.score_chain(name.encode(), cfg)
            out[repr(rate)][name] = md5(C.string_at(r.contents.contig).decode())
    return out


def ref_trace(fa, bam):
    """score_chain through the reference's own shared library with trace_polish_open=1 (what nextpolish1.py -debug does)."""
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "nextpolish1.so"))
    L.config_init.restype = C.POINTER(nat.Configure)
    L.config_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.score_chain.restype = C.POINTER(nat.PolishResult)
    L.score_chain.argtypes = [C.c_char_p, C.POINTER(nat.Configure)]
    cfg = L.config_init(fa.encode(), bam.encode(), None)
    cfg.contents.trace_polish_open = 1
    res = {"read_tlen": cfg.contents.read_tlen, "read_len": cfg.contents.read_len, "points": {}}
    for line in open(fa + ".fai"):
        name = line.split("\t")[0]
        r = L.score_chain(name.encode(), cfg)
        pts = [[r.contents.data[k].pos, r.contents.data[k].index, r.contents.data[k].curbase.decode(),
                r.contents.data[k].base.decode()] for k in range(r.contents.datalength)]
        res["points"][name] = pts
    return res


def main():
    mp = os.environ.get("NP_MAPPERS")
    if not mp or not os.path.exists(os.path.join(mp, "bwa", "bwa")):
        sys.exit("set NP_MAPPERS to the directory holding bwa/ minimap2/ samtools/ built from the vendored sources")
    os.environ["PATH"] = "%s/bwa:%s/minimap2:%s/samtools:" % (mp, mp, mp) + os.environ["PATH"]
    w = tempfile.mkdtemp(prefix="npreal_")
    shutil.copy(os.path.join(T, "raw.genome.fasta"), os.path.join(w, "g.fa"))
    os.chmod(os.path.join(w, "g.fa"), 0o644)
    r12 = "%s/sreads.R1.fastq.gz %s/sreads.R2.fastq.gz" % (T, T)
    post = "samtools view -F 0x4 -b - | samtools fixmate -m - - | samtools sort - | samtools markdup -r - "
    sh("bwa index g.fa && samtools faidx g.fa && bwa mem -t8 g.fa %s | %s sgs.sort.bam && samtools index sgs.sort.bam" % (r12, post), w)
    sh("samtools view -b -s 7.2 sgs.sort.bam > sgs.s30.bam && samtools index sgs.s30.bam", w)
    gold = {"sr": {}, "lr": {}}
    for tag in ("sgs.sort", "sgs.s30"):
        bam = os.path.join(w, tag + ".bam")
        tr = ref_trace(os.path.join(w, "g.fa"), bam)
        gold["sr"][tag] = {"fasta": "g.fa", "bam": tag + ".bam", "read_tlen": tr["read_tlen"], "read_len": tr["read_len"],
                           "score_chain": digest(run_ref("scorechain", os.path.join(w, "g.fa"), bam))}
        if tag == "sgs.s30":
            gold["sr"][tag]["points"] = tr["points"]
            gold["sr"][tag]["rates"] = ref_rates(os.path.join(w, "g.fa"), bam, [0.3, 0.55, 0.9])
    # task 2 round: the reference's own task-1 output, re-mapped
    sc = run_ref("scorechain", os.path.join(w, "g.fa"), os.path.join(w, "sgs.sort.bam"))
    with open(os.path.join(w, "r1.fa"), "w") as f:
        for n, s in sc.items():
            f.write(">%s_1\n%s\n" % (n, s))
    sh("bwa index r1.fa && samtools faidx r1.fa && bwa mem -t8 r1.fa %s | %s r1.sort.bam && samtools index r1.sort.bam" % (r12, post), w)
    sh("samtools view -b r1.sort.bam %s > r1.slice.bam && samtools index r1.slice.bam" % " ".join(SLICES), w)
    bam = os.path.join(w, "r1.slice.bam")
    tr = ref_trace(os.path.join(w, "r1.fa"), bam)
    gold["sr"]["r1.slice"] = {"fasta": "r1.fa", "bam": "r1.slice.bam", "read_tlen": tr["read_tlen"], "read_len": tr["read_len"],
                              "score_chain": digest(run_ref("scorechain", os.path.join(w, "r1.fa"), bam)),
                              "kmer_count": digest(run_ref("kmercount", os.path.join(w, "r1.fa"), bam))}
    # long reads
    sh("minimap2 -ax map-ont -t8 g.fa %s/lreads.fasta.gz | samtools sort -o lgs.sort.bam && samtools index lgs.sort.bam" % T, w)
    sh("minimap2 -ax asm20 -t8 g.fa %s/hifi.fasta.gz | samtools sort -o hifi.sort.bam && samtools index hifi.sort.bam" % T, w)
    L = rb.bind(rb.REF_SO)
    for tag, rts in (("lgs.sort", (1, 2)), ("hifi.sort", (3,))):
        fofn = os.path.join(w, tag + ".fofn")
        open(fofn, "w").write(os.path.join(w, tag + ".bam") + "\n")
        for rt in rts:
            for split in (0, 1):
                got = rb.polish(L, os.path.join(w, "g.fa"), fofn, read_type=rt, split=split)
                gold["lr"]["%s.rt%d.split%d" % (tag, rt, split)] = {
                    "fasta": "g.fa", "bam": tag + ".bam", "read_type": rt, "split": split,
                    "expected": {n: [{"len": l, "md5": md5(s)} for s, l in pieces] for n, pieces in got.items()}}
    os.makedirs(OUT, exist_ok=True)
    for fn in ("g.fa", "g.fa.fai", "r1.fa", "r1.fa.fai", "sgs.sort.bam", "sgs.s30.bam", "r1.slice.bam", "lgs.sort.bam",
               "hifi.sort.bam"):
        shutil.copy(os.path.join(w, fn), os.path.join(OUT, fn))
        if fn.endswith(".bam"):
            shutil.copy(os.path.join(w, fn + ".bai"), os.path.join(OUT, fn + ".bai"))
    json.dump(gold, open(os.path.join(OUT, "real_golden.json"), "w"), indent=0, sort_keys=True)
    shutil.rmtree(w)
    print("wrote", OUT, {k: list(v) for k, v in gold.items()})


if __name__ == "__main__":
    main()
