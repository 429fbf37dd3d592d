#!/usr/bin/env python
"""Generates tests/golden/np2_golden.json: expected ctg_cns_core output of the COMPILED REFERENCE
(oracle/_ref/nextpolish2.so, built by oracle/Makefile from /root/reference) for the cases in tests/np2_cases.py,
plus known answers of the 2-bit codec and read_ref.  Run in the build container (needs oracle/_ref)."""
import ctypes as C
import hashlib
import json
import tempfile
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import np2_cases  # noqa: E402
import ref2_binding as rb  # noqa: E402


def main():
    L = rb.bind(rb.REF_SO)
    out = {"cases": {}, "codec": []}
    for cid, kw, rt in np2_cases.CASES:
        fa, fofn, contigs = np2_cases.materialise(kw)
        res = rb.polish(L, fa, fofn, read_type=rt)
        out["cases"][cid] = {
            "read_type": rt,
            "draft_md5": {n: hashlib.md5(d.encode()).hexdigest() for n, d in contigs},
            "expected": {n: res[n][0][0] for n, _ in contigs},
            "pieces": {n: len(res[n]) for n, _ in contigs},
        }
    # one contig longer than the (minimum-size) window: two windows with 1 Mb overlap, stitched by link_consensus
    fa, fofn, contigs = np2_cases.materialise(np2_cases.TWO_WINDOW_CASE)
    res = rb.polish(L, fa, fofn, window=np2_cases.TWO_WINDOW_W)
    s = res["ctg0"][0][0]
    out["two_windows"] = {"md5": hashlib.md5(s.encode()).hexdigest(), "len": len(s), "pieces": len(res["ctg0"])}
    # split-read structural layer: md5 + piece lengths of the reference's output
    out["sv"] = {}
    for cid, kw, rt, split, qvs in np2_cases.SV_CASES:
        fa, fofn, contigs = np2_cases.materialise_sv(kw, qvs)
        res = rb.polish(L, fa, fofn, read_type=rt, split=split)["ctg0"]
        out["sv"][cid] = {"lens": [r[1] for r in res], "md5": [hashlib.md5(r[0].encode()).hexdigest() for r in res]}
    # a megabase window from the native generator (enough runs for the several-runs-per-wave DP kernels and the two-level scan)
    from nextpolish_amd import _native as nat
    d = tempfile.mkdtemp(prefix="np2mb_")
    st = nat.Stream.synth_long([1200000], depth=20.0, seed=31)
    st.write_files(os.path.join(d, "g.fa"), os.path.join(d, "r.bam"))
    st.close()
    open(os.path.join(d, "bam.fofn"), "w").write(os.path.join(d, "r.bam") + "\n")
    res = rb.polish(L, os.path.join(d, "g.fa"), os.path.join(d, "bam.fofn"), read_type=1)["ctg0"]
    out["mb_window"] = {"lens": [r[1] for r in res], "md5": [hashlib.md5(r[0].encode()).hexdigest() for r in res]}
    # a deep pileup (100x): narrow columns are rare, the GPU path widens its cuts from 8 to 32 live entries
    d = tempfile.mkdtemp(prefix="np2deep_")
    st = nat.Stream.synth_long([150000], depth=100.0, seed=32)
    st.write_files(os.path.join(d, "g.fa"), os.path.join(d, "r.bam"))
    st.close()
    open(os.path.join(d, "bam.fofn"), "w").write(os.path.join(d, "r.bam") + "\n")
    res = rb.polish(L, os.path.join(d, "g.fa"), os.path.join(d, "bam.fofn"), read_type=1)["ctg0"]
    out["deep_window"] = {"lens": [r[1] for r in res], "md5": [hashlib.md5(r[0].encode()).hexdigest() for r in res]}
    # several BAM files in the fofn (merge by position, strand, file)
    cid, kw, rt = np2_cases.CASES[1]
    fa, fofn, contigs = np2_cases.materialise_multi(kw, 3)
    res = rb.polish(L, fa, fofn, read_type=rt)
    out["multi_bam"] = {"case": cid, "expected": {n: res[n][0][0] for n, _ in contigs}}
    # codec known answers: pack then unpack through the reference (incl. the non-ACGT spill of bseq.c:91)
    for s in ["ACGT", "AANAA", "GGNGG", "acgtn", "TTTTTTTTTTTTTTTTA", "NACGT", "ACGTACGTACGTACGTN", "RYKM", "A", "TU"]:
        words = (C.c_uint32 * (len(s) // 16 + 1))()
        L.seq2bit1(words, len(s), s.encode())
        buf = C.create_string_buffer(len(s) + 1)
        L.bit2seq1(words, len(s), buf)
        out["codec"].append({"seq": s, "words": [int(w) for w in words][: (len(s) + 15) // 16], "round_trip": buf.value.decode()})
    # known answers of the two string algorithms the reference exports: poa_to_consensus (dag.c) and align (align.c)
    import random
    import np2_strings
    R = C.CDLL(rb.REF_SO)
    out["poa"], out["align"] = [], []
    for seed in range(60):
        seqs = np2_strings.poa_case(random.Random(seed))
        out["poa"].append({"seqs": seqs, "consensus": np2_strings.ref_poa(R, seqs)})
    for seed in range(80):
        q, t = np2_strings.align_case(random.Random(seed), seed)
        n, ts, qs, tl, ql = np2_strings.ref_align(R, q, t)
        out["align"].append({"q": q, "t": t, "aln_len": n, "t_aln": ts if n > 2 else "", "q_aln": qs if n > 2 else "",
                             "t_len": tl if n > 2 else 0, "q_len": ql if n > 2 else 0})
    with open(os.path.join(HERE, "np2_golden.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
