#!/usr/bin/env python
"""BASELINE config 5 as ONE run, at a reduced but multi-batch / multi-window size: the reference's `task = best` order for short + long reads
(lib/config_parser.py:88: 5 5 [6 6] 1 2 1 2 -> lgs, lgs, score_chain, kmer_count, score_chain, kmer_count), every step polishing the FASTA the
step before wrote, both libraries in one process (nextpolish2.so for the long-read steps, nextpolish1.so's file pipe for the short-read ones).

The reference re-maps the reads to the new assembly before every step (source/nextPolish:389-510: index_genome, map_genome, merge_bam); there
is no mapper in this image, so the reads of a step are generated ON the assembly the step before wrote, aligned by construction
(nat.Stream.synth_on: a truth derived from the assembly by the generator's edit process, short pairs sampled from it; long reads as noisy
copies of the assembly) and written as FASTA + sorted BAM -- the same files for both sides.  Each step's output (every contig: length + md5)
is compared with what the COMPILED REFERENCE wrote for the same files (tests/golden/config5_chain_golden.json, made here with
--make-golden: oracle/_ref/nextpolish1 for tasks 1 and 2, oracle/_ref/nextpolish2.so for task 5); the chain goes on from this side's own
output, which is the reference's when the step was identical.

usage: check_config5_chain.py [--quick] [--make-golden]          prints one JSON line; exit code 1 on a mismatch"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT)
sys.path.insert(0, TESTS)

GOLDEN = os.path.join(TESTS, "golden", "config5_chain_golden.json")
STEPS = [5, 5, 1, 2, 1, 2]
# contig lengths, draft bases per short-read batch, long-read window (the reference's own 5 Mb: ctg_cns.c:3371 refuses windows of less than four
# overlaps of 1 Mb).  quick: four short-read batches, one window per contig; full: three short-read batches, the first contig in three windows
SIZES = {
    "quick": dict(lens=[600000, 400000, 250000, 150000], batch_bp=500000, window=5000000),
    "full": dict(lens=[11000000, 4000000, 2500000, 1500000], batch_bp=4000000, window=5000000),
}
REF2_CHILD = ("import sys, json; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(sys.argv[1]); "
              "r = rb.polish(L, sys.argv[2], sys.argv[3], window=int(sys.argv[4]), read_type=1); "
              "print(json.dumps({n: [s for s, l in p] for n, p in r.items()}))" % TESTS)


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


def first_assembly(lens):
    rng = np.random.RandomState(20260105)
    return [("chr%d" % (i + 1), bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n)).decode()) for i, n in enumerate(lens)]


def pieces_to_contigs(order, res):
    """ctg_cns_core's pieces as the caller names them (lib/nextpolish2.py:190-197: name, or name_s<i> when a contig comes back in pieces)"""
    out = []
    for n in order:
        p = res[n]
        for i, s in enumerate(p):
            out.append((n + ("_s%d" % i if len(p) != 1 else ""), s))
    return out


def write_step(nat, contigs, task, k, d):
    """the step's input files: the assembly + reads generated on it"""
    st = nat.Stream.synth_on(contigs, long_reads=(task == 5), seed=7100 + k)
    fa, bam = os.path.join(d, "s%d.fa" % k), os.path.join(d, "s%d.bam" % k)
    st.write_files(fa, bam)
    n = st.n_reads
    st.close()
    return fa, bam, n


def run(size, make_golden=False, workdir=None, engine="gpu"):
    """engine "gpu": the product (nextpolish2.so, nextpolish1.so's file pipe); "model": the host models of both launch sequences
    (tests/model/libnp2_model.so, libnp1_model.so) -- the same chain on the CPU, for the suite that runs without a GPU"""
    from nextpolish_amd import _native as nat
    from conftest import parse_cli_fasta
    import ref2_binding as rb
    S = SIZES[size]
    gold = json.load(open(GOLDEN)) if os.path.exists(GOLDEN) else {}
    d = workdir or tempfile.mkdtemp(prefix="np_c5chain_")
    contigs = first_assembly(S["lens"])
    info = {"size": size, "draft_bp": sum(S["lens"]), "steps": [], "identical": True}
    pipe = L2 = None
    if not make_golden and engine == "model":
        import model_binding as mb
        L2 = rb.bind(os.path.join(TESTS, "model", "libnp2_model.so"))
    elif not make_golden:
        from nextpolish_amd.device import Pipe
        pipe = Pipe(0, lanes=2)
        L2 = rb.bind(os.path.join(ROOT, "nextpolish_amd", "lib", "nextpolish2.so"))
    made = []
    try:
        for k, task in enumerate(STEPS):
            t0 = time.time()
            fa, bam, n_reads = write_step(nat, contigs, task, k, d)
            order = [n for n, _ in contigs]
            t1 = time.time()
            if task == 5:
                fofn = os.path.join(d, "s%d.fofn" % k)
                open(fofn, "w").write(bam + "\n")
                if make_golden:
                    p = subprocess.run([sys.executable, "-c", REF2_CHILD, os.path.join(ROOT, "oracle", "_ref", "nextpolish2.so"), fa, fofn, str(S["window"])],
                                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
                    res = json.loads(p.stdout.strip().splitlines()[-1])
                else:
                    res = {n: [s for s, _ in p] for n, p in rb.polish(L2, fa, fofn, window=S["window"], read_type=1).items()}
                nxt = pieces_to_contigs(order, res)
            else:
                cmd = "scorechain" if task == 1 else "kmercount"
                if make_golden:
                    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "nextpolish1"), cmd, fa, bam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                         check=True).stdout.decode()
                    got = parse_cli_fasta(out)
                elif engine == "model":
                    st = nat.Stream.load(fa, bam, with_qual=(task == 2))
                    res1 = mb.score_chain(st, fused=1) if task == 1 else mb.kmer_count_replay(st, nat.default_config(), bam)
                    got = dict(zip(st.names, res1))
                    st.close()
                else:
                    got = dict(pipe.run_files(fa, bam, batch_bp=S["batch_bp"], task=task))
                nxt = [(n, got[n]) for n in order]
            digest = {n: [len(s), md5(s)] for n, s in nxt}
            step = {"task": task, "reads": n_reads, "contigs": len(nxt), "bp_out": sum(len(s) for _, s in nxt), "write_s": round(t1 - t0, 1), "polish_s": round(time.time() - t1, 1)}
            if make_golden:
                made.append(digest)
            else:
                want = gold.get(size, [None] * len(STEPS))[k]
                step["identical"] = want == digest
                if not step["identical"]:
                    info["identical"] = False
                    step["differing"] = sorted(n for n in set(digest) | set(want or {}) if (want or {}).get(n) != digest.get(n))[:8]
            info["steps"].append(step)
            contigs = nxt
            for f in (fa, fa + ".fai", bam, bam + ".bai"):
                if os.path.exists(f):
                    os.remove(f)
            if not info["identical"]:
                break      # (what follows would polish a different assembly than the golden's)
    finally:
        if pipe is not None:
            pipe.close()
        if workdir is None:
            shutil.rmtree(d, ignore_errors=True)
    if make_golden:
        gold[size] = made
        json.dump(gold, open(GOLDEN, "w"), indent=0, sort_keys=True)
    return info


if __name__ == "__main__":
    r = run("quick" if "--quick" in sys.argv else "full", make_golden="--make-golden" in sys.argv)
    print(json.dumps(r))
    sys.exit(0 if r["identical"] else 1)
