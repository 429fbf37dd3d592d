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
]

# a contig longer than the smallest window the reference accepts (window must exceed 4 x the 1 Mb overlap)
TWO_WINDOW_CASE = dict(seed=77, contig_lens=(4300000,), depth=3, mean_len=12000, max_indel=3)
TWO_WINDOW_W = 4100000


def materialise(case_kw, workdir=None):
    """Writes FASTA + BAM(+BAI) + BAM list for one case; returns (fasta, bam_list, contigs)."""
    from nextpolish_amd import _native as nat
    kw = dict(case_kw)
    seed = kw.pop("seed")
    contigs, reads = np2_gen.make_case(seed, **kw)
    d = workdir or tempfile.mkdtemp(prefix="np2case_")
    st = nat.Stream.from_reads(contigs, reads)
    fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
    st.write_files(fa, bam)
    st.close()
    fofn = os.path.join(d, "bam.fofn")
    with open(fofn, "w") as f:
        f.write(bam + "\n")
    return fa, fofn, contigs
