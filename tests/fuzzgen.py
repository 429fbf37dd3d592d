"""Random micro-workloads with deliberately odd CIGAR shapes, contig edges and draft letters
(SURVEY.md §8c "hand-made micro-BAMs hitting each quirk").  Deterministic per seed."""
import random


def random_case(seed, n_contigs=2, max_len=160, max_reads=40, letters="ACGT", odd_letters=True, odd_cigars=True, double_ins=False):
    """double_ins: insertion operations that meet at one reference position (I P I, I N I: the walk ignores P and N), which no aligner
    writes but contig.c:299-320 votes on twice"""
    rng = random.Random(seed)
    contigs, reads = [], []
    for c in range(n_contigs):
        L = rng.randint(12, max_len)
        alpha = letters
        d = [rng.choice(alpha) for _ in range(L)]
        if odd_letters:
            for _ in range(rng.randint(0, 4)):
                i = rng.randrange(L)
                d[i] = rng.choice("acgtNnMRYKm")
        draft = "".join(d)
        contigs.append(("tig%d" % c, draft))
        rs = []
        for _ in range(rng.randint(0, max_reads)):
            ref_len = rng.randint(6, min(L, 60))
            pos = rng.randint(0, L - ref_len)
            if rng.random() < 0.25:
                pos = 0 if rng.random() < 0.5 else L - ref_len
            # alignment ops over the reference span
            cig, seq = [], []
            def add(op, n):
                if n <= 0:
                    return
                if cig and cig[-1][0] == op:
                    cig[-1] = (op, cig[-1][1] + n)
                else:
                    cig.append((op, n))
            if odd_cigars and rng.random() < 0.1:
                add("H", rng.randint(1, 6))
            if rng.random() < 0.3:
                n = rng.randint(1, 8)
                add("S", n)
                seq += [rng.choice("ACGT") for _ in range(n)]
            if odd_cigars and rng.random() < 0.15:
                n = rng.randint(1, 4)          # leading insertion (also at pos 0)
                add("I", n)
                seq += [rng.choice("ACGT") for _ in range(n)]
                last_was_ins = True
            else:
                last_was_ins = False
            p = pos
            end = pos + ref_len
            while p < end:
                x = rng.random()
                n = min(end - p, rng.randint(1, 12))
                if x < 0.62:
                    for k in range(n):
                        b = draft[p + k].upper()
                        if b not in "ACGT" or rng.random() < 0.08:
                            b = rng.choice("ACGT")
                        seq.append(b)
                    if rng.random() < 0.06:
                        seq[-1] = "N"
                    add("M", n); p += n; last_was_ins = False
                elif x < 0.74:
                    n = min(n, 3)
                    add("D", n); p += n; last_was_ins = False
                elif x < 0.88:
                    if last_was_ins:
                        if not double_ins or rng.random() < 0.3:
                            continue
                        add(rng.choice("PN"), 1)
                    n = rng.randint(1, 4)
                    add("I", n); seq += [rng.choice("ACGT") for _ in range(n)]; last_was_ins = True
                elif odd_cigars and x < 0.92:
                    add("N", min(n, 3)); p += min(n, 3)   # the walk ignores N entirely (no pos advance)
                elif odd_cigars and x < 0.97:
                    op = rng.choice("=X")
                    add(op, n); p += n
                    seq += [rng.choice("ACGT") for _ in range(n)]
                elif odd_cigars:
                    add("P", 1)
            if rng.random() < 0.3:
                n = rng.randint(1, 8)
                add("S", n)
                seq += [rng.choice("ACGT") for _ in range(n)]
            if odd_cigars and rng.random() < 0.1:
                add("H", rng.randint(1, 6))
            if len(seq) >= 2 and rng.random() < 0.1:   # homopolymer-rich read ends / whole read
                b = rng.choice("ACGT")
                k = rng.randint(2, len(seq))
                if rng.random() < 0.5:
                    seq[:k] = [b] * k
                else:
                    seq[-k:] = [b] * k
            flag = rng.choice([0, 16, 99, 147, 83, 163])
            if rng.random() < 0.05:
                flag |= rng.choice([0x400, 0x800, 0x100, 0x4])
            rs.append(dict(ctg=c, pos=pos, flag=flag, mapq=rng.choice([60, 60, 60, 0, 13, 30]),
                           isize=rng.choice([0, 300, -300, 250, 20000, -7]), cigar=cig, seq="".join(seq),
                           qual=[rng.randint(2, 41) for _ in seq]))
        rs.sort(key=lambda r: r["pos"])
        reads += rs
    return contigs, reads


# ---- inputs beyond the bounds of the fast launch sequence (tests of the lifted limits) -----------------------------------------
def long_record_case(seed, n_ops=70000, plain_bases=70000):
    """one contig; a record of more than 65 535 CIGAR operations (what a CG tag carries), one of more than 65 535 bases in a single
    match, and a shallow layer of ordinary reads over the same stretch"""
    import random
    rng = random.Random(seed)
    L = 2 * n_ops + plain_bases + 3000
    draft = "".join(rng.choice("ACGT") for _ in range(L))
    reads = []
    # (a) alternating 1M / 1I / 1M / 1D ... : n_ops operations
    cig, seq, g = [], [], 500
    pos_a = g
    for k in range(n_ops // 4):
        cig += [("M", 2), ("I", 1), ("M", 1), ("D", 1)]
        seq += [draft[g], draft[g + 1], rng.choice("ACGT"), draft[g + 2]]
        g += 4
    cig.append(("M", 40))
    seq += list(draft[g:g + 40])
    reads.append(dict(ctg=0, pos=pos_a, cigar=cig, seq="".join(seq)))
    # (b) one long match with a few substitutions
    pos_b = pos_a + 1000
    sb = list(draft[pos_b:pos_b + plain_bases])
    for _ in range(50):
        sb[rng.randrange(len(sb))] = rng.choice("ACGT")
    reads.append(dict(ctg=0, pos=pos_b, cigar=[("M", plain_bases)], seq="".join(sb)))
    # (c) ordinary reads
    for _ in range(L // 30):
        p = rng.randrange(0, L - 150)
        s = list(draft[p:p + 150])
        if rng.random() < 0.3:
            s[rng.randrange(150)] = rng.choice("ACGT")
        reads.append(dict(ctg=0, pos=p, cigar=[("M", 150)], seq="".join(s)))
    reads.sort(key=lambda r: r["pos"])
    return [("long", draft)], reads


def crowded_context_case(seed, depth=900, L=400):
    """a pileup whose reads carry every nt16 code at random: far more than 160 distinct 3-symbol contexts in a slot"""
    import random
    rng = random.Random(seed)
    draft = "".join(rng.choice("ACGT") for _ in range(L))
    codes = "=ACMGRSVTWYHKDBN"
    reads = []
    for _ in range(depth):
        p = rng.randrange(0, L - 100)
        s = "".join(rng.choice(codes) for _ in range(100))
        if rng.random() < 0.3:
            k = rng.randrange(10, 90)
            reads.append(dict(ctg=0, pos=p, cigar=[("M", k), ("I", 2), ("M", 98 - k)], seq=s))
        else:
            reads.append(dict(ctg=0, pos=p, cigar=[("M", 100)], seq=s))
    reads.sort(key=lambda r: r["pos"])
    return [("crowd", draft)], reads
