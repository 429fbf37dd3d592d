"""The long-read parity cases shared by the golden generator, the host-model tests and the GPU tests."""
import os
import tempfile

import np2_gen

# (case id, generator keywords, read type 1 = ONT / 2 = CLR)
CASES = [
    ("ont_20x_two_contigs", dict(seed=1, contig_lens=(20000, 6000), depth=20), 1),
    ("clr_20x", dict(seed=5, contig_lens=(9000,), depth=20, max_indel=1), 2),
    ("ont_8x_three_contigs", dict(seed=6, contig_lens=(30000, 1500, 700), depth=8, sub=0.005, max_indel=1, mean_len=9000), 1),
    ("ont_35x_noisy", dict(seed=8, contig_lens=(20000, 6000), depth=35, sub=0.08, ins=0.04, dele=0.002, max_indel=1, mean_len=1500), 1),
    ("clr_35x", dict(seed=9, contig_lens=(9000,), depth=35, sub=0.03, ins=0.04, dele=0.05, max_indel=1), 2),
    ("ont_draft_with_N", dict(seed=21, contig_lens=(9000,), depth=8, n_rate=0.001), 1),
    ("ont_60x", dict(seed=33, contig_lens=(12000, 12000), depth=60, max_indel=2, ins=0.002, dele=0.02), 1),
    # inputs whose windows contain low-quality regions (>= 3 bp insertions): the re-consensus stage decides the result
    ("ont_lq_regions", dict(seed=10, contig_lens=(30000, 1500, 700), depth=35, max_indel=6), 1),
    ("clr_lq_regions", dict(seed=13, contig_lens=(9000,), depth=60, max_indel=6, sub=0.08), 2),
    # HiFi (read type 3): its own DP tie rule, low-qv-run regions and identical-candidate vote
    ("hifi_20x", dict(seed=40, contig_lens=(20000, 6000), depth=20, sub=0.002, ins=0.002, dele=0.002, mean_len=9000, clip_rate=0.02), 3),
    ("hifi_35x_indels", dict(seed=46, contig_lens=(20000, 6000), depth=35, sub=0.002, ins=0.01, dele=0.008, max_indel=6, mean_len=9000,
                             clip_rate=0.02), 3),
    # ambiguity codes in the READS: M is also the marker character of the reference's low-quality concatenation, so an
    # M base is skipped by the coverage count and drops its own and the next link (ctg_cns.c:1232,334)
    ("ont_reads_with_iupac_codes", dict(seed=52, contig_lens=(15000, 4000), depth=25, max_indel=4, iupac_rate=0.002), 1),
    # single reads carrying insertions of 300-2500 bases: columns with thousands of nodes (real ONT data has them)
    ("ont_long_insertions", dict(seed=71, contig_lens=(40000,), depth=25, max_indel=4, mean_len=8000, long_ins_rate=2e-5), 1),
    # a contig without a single read (comes back as the draft in lower case), and a contig two reads touch
    ("contig_without_reads", dict(seed=3, contig_lens=(6000, 3000, 400), depth=12, mean_len=2000, drop_ctg=1), 1),
    ("two_reads_only", dict(seed=4, contig_lens=(5000,), depth=1, mean_len=3000, keep_reads=2), 1),
    ("hifi_reads_with_iupac_codes", dict(seed=53, contig_lens=(15000,), depth=25, sub=0.002, ins=0.002, dele=0.002, mean_len=9000, iupac_rate=0.001), 3),
]

# a contig longer than the smallest window the reference accepts (window must exceed 4 x the 1 Mb overlap)
TWO_WINDOW_CASE = dict(seed=77, contig_lens=(4300000,), depth=3, mean_len=12000, max_indel=3)
TWO_WINDOW_W = 4100000


def materialise(case_kw, workdir=None):
    """Writes FASTA + BAM(+BAI) + BAM list for one case; returns (fasta, bam_list, contigs)."""
    from nextpolish_amd import _native as nat
    kw = dict(case_kw)
    seed = kw.pop("seed")
    drop_ctg, keep_reads = kw.pop("drop_ctg", None), kw.pop("keep_reads", None)
    contigs, reads = np2_gen.make_case(seed, **kw)
    if drop_ctg is not None:     # a contig no read maps to
        reads = [r for r in reads if r["ctg"] != drop_ctg]
    if keep_reads is not None:   # only the first few reads
        reads = reads[:keep_reads]
    d = workdir or tempfile.mkdtemp(prefix="np2case_")
    st = nat.Stream.from_reads(contigs, reads)
    fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
    st.write_files(fa, bam)
    st.close()
    fofn = os.path.join(d, "bam.fofn")
    with open(fofn, "w") as f:
        f.write(bam + "\n")
    return fa, fofn, contigs


# split-read structural layer (contig > 100 kb, >= 150 reads, SA-split supplementary records): gap clusters from
# blocks the draft lacks, a stretch no read crosses (split point), an assembly QV track in the FASTA comment
# (id, make_sv_case keywords, read type, split mode, [(pos, ide, ort, irt)] or None)
SV_CASES = [
    ("sv_ont_two_blocks", dict(seed=2, depth=40), 1, 1, None),
    ("sv_hifi_blocks_recovered", dict(seed=6, depth=40, sub=0.005, ins=0.003, dele=0.003), 3, 1, None),
    ("sv_clr", dict(seed=7, depth=40, mean_len=12000), 2, 1, None),
    ("sv_hole_split_pieces", dict(seed=9, depth=40, hole=(70000, 70300)), 1, 1, None),
    ("sv_hole_split_N", dict(seed=9, depth=40, hole=(70000, 70300)), 1, 2, None),
    ("sv_refqv_split_point", dict(seed=14, depth=40, hole=(70000, 70300)), 1, 1,
     [(10000, 800, 900, 900), (30000, 790, 900, 900), (69800, 800, 900, 990), (100000, 805, 800, 820)]),
    ("sv_refqv_merged_regions", dict(seed=12, depth=40, hole=(70000, 70300)), 1, 1,
     [(20000, 700, 900, 900), (50000, 300, 300, 300), (69500, 200, 250, 240), (70100, 100, 100, 100), (100000, 650, 800, 820)]),
]


def materialise_sv(case_kw, qvs=None, workdir=None):
    from nextpolish_amd import _native as nat
    kw = dict(case_kw)
    seed = kw.pop("seed")
    contigs, reads, aux = np2_gen.make_sv_case(seed, **kw)
    d = workdir or tempfile.mkdtemp(prefix="np2sv_")
    st = nat.Stream.from_reads(contigs, reads)
    fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
    st.write_files(fa, bam, aux=aux)
    st.close()
    if qvs:   # per-contig QV track in the FASTA comment (set_ref_qv, ctg_cns.c:2233-2267)
        hexes = ":".join("%x" % (p << 32 | ide << 20 | ort << 10 | irt) for p, ide, ort, irt in qvs)
        lines = open(fa).read().split("\n")
        lines[0] = ">%s node_c=%d qv_h=%s" % (contigs[0][0], len(qvs), hexes)
        with open(fa, "w") as f:
            f.write("\n".join(lines))
        os.remove(fa + ".fai")
    fofn = os.path.join(d, "bam.fofn")
    with open(fofn, "w") as f:
        f.write(bam + "\n")
    return fa, fofn, contigs


def materialise_multi(case_kw, n_files=3, workdir=None):
    """The reads of a case dealt over several BAM files (the reference merges the files of the fofn by position,
    strand and file order, bsort.c:174-199): FASTA, fofn, contigs."""
    from nextpolish_amd import _native as nat
    kw = dict(case_kw)
    seed = kw.pop("seed")
    contigs, reads = np2_gen.make_case(seed, **kw)
    d = workdir or tempfile.mkdtemp(prefix="np2mf_")
    fa = os.path.join(d, "g.fa")
    bams = []
    for f in range(n_files):
        part = [r for i, r in enumerate(reads) if (i * 7 + i // 5) % n_files == f]
        st = nat.Stream.from_reads(contigs, part)
        bam = os.path.join(d, "r%d.bam" % f)
        st.write_files(fa, bam)
        st.close()
        bams.append(bam)
    fofn = os.path.join(d, "bam.fofn")
    with open(fofn, "w") as fh:
        fh.write("\n".join(bams) + "\n")
    return fa, fofn, contigs


# BASELINE config 4 at its stated size: ~100 Mb draft, many contigs, 20x ONT-like reads; contigs longer than the 5 Mb window
# are polished in overlapping windows and stitched (12.5 Mb = 3 windows, 9 Mb = 2 ...).  Groups = one FASTA + BAM + fofn
# each (one per worker process of the harness); golden md5s of the compiled reference: tests/golden/config4_golden.json
# (tests/golden/make_config4_golden.py).
def config4_groups(total=100000000, seed=20250117 + 4, n_groups=8):
    import math
    import random
    rng = random.Random(seed)
    lens = [12500000, 9000000, 6000000, 5200000]
    acc = sum(lens)
    while acc < total:
        L = int(math.exp(rng.uniform(math.log(50e3), math.log(5e6))))
        L = min(L, total - acc) if total - acc > 50000 else total - acc
        lens.append(L)
        acc += L
    groups = [[] for _ in range(n_groups)]
    load = [0] * n_groups
    for L in sorted(lens, reverse=True):        # longest first onto the lightest group
        k = load.index(min(load))
        groups[k].append(L)
        load[k] += L
    return groups


def materialise_config4_group(k, lens, workdir, depth=20.0):
    from nextpolish_amd import _native as nat
    d = os.path.join(workdir, "g%d" % k)
    os.makedirs(d, exist_ok=True)
    st = nat.Stream.synth_long(lens, depth=depth, seed=9000 + k, prefix="g%dc" % k)
    fa, bam, fofn = os.path.join(d, "g.fa"), os.path.join(d, "r.bam"), os.path.join(d, "bam.fofn")
    st.write_files(fa, bam)
    names = list(st.names)
    st.close()
    with open(fofn, "w") as f:
        f.write(bam + "\n")
    return fa, fofn, names
