"""Synthetic long-read workloads for the nextpolish2 path (test helper): a random draft and noisy reads with known
CIGARs, written as FASTA + coordinate-sorted BAM + BAI through the library's own writer."""
import random


def make_case(seed, contig_lens=(20000,), depth=20, mean_len=4000, sub=0.03, ins=0.02, dele=0.02, max_indel=2,
              clip_rate=0.2, lower=False, n_rate=0.0, name_prefix="ctg", iupac_rate=0.0, long_ins_rate=0.0):
    rng = random.Random(seed)
    rng3 = random.Random(seed * 104729 + 7)   # long insertions (own generator: the other cases keep their sequences)
    contigs, reads = [], []
    for ci, L in enumerate(contig_lens):
        draft = "".join(rng.choice("ACGT") for _ in range(L))
        if n_rate:
            draft = "".join("N" if rng.random() < n_rate else c for c in draft)
        if lower:
            draft = "".join(c.lower() if rng.random() < 0.05 else c for c in draft)
        contigs.append(("%s%d" % (name_prefix, ci), draft))
        up = draft.upper()
        n_reads = max(1, int(depth * L / mean_len))
        starts = sorted(rng.randrange(0, max(1, L - 600)) for _ in range(n_reads))
        for st in starts:
            rl = max(700, int(rng.lognormvariate(0, 0.5) * mean_len))
            en = min(L, st + rl)
            ops, seq = [], []
            pos = st
            # the alignment must start and end with a match column
            def push(op, n=1):
                if ops and ops[-1][0] == op:
                    ops[-1][1] += n
                else:
                    ops.append([op, n])
            while pos < en:
                first_or_last = pos == st or pos >= en - 1
                x = rng.random()
                if not first_or_last and x < dele:
                    n = min(rng.randint(1, max_indel), en - 1 - pos)
                    if n > 0:
                        push("D", n)
                        pos += n
                        continue
                if not first_or_last and x < dele + ins:
                    n = rng.randint(1, max_indel)
                    push("I", n)
                    seq += [rng.choice("ACGT") for _ in range(n)]
                    # an insertion is followed by a match column
                if long_ins_rate and not first_or_last and rng3.random() < long_ins_rate:
                    n = rng3.randint(300, 2500)   # one read carries a long insertion: a column with thousands of nodes
                    push("I", n)
                    seq += [rng3.choice("ACGT") for _ in range(n)]
                c = up[pos] if up[pos] in "ACGT" else rng.choice("ACGT")
                if rng.random() < sub:
                    c = rng.choice([b for b in "ACGT" if b != c])
                push("M", 1)
                seq.append(c)
                pos += 1
            cig = [(o, n) for o, n in ops]
            if rng.random() < clip_rate:
                n = rng.randint(1, 300)
                cig = [("S", n)] + cig
                seq = [rng.choice("ACGT") for _ in range(n)] + seq
            if rng.random() < clip_rate:
                n = rng.randint(1, 300)
                cig = cig + [("S", n)]
                seq = seq + [rng.choice("ACGT") for _ in range(n)]
            reads.append(dict(ctg=ci, pos=st, flag=16 if rng.random() < 0.5 else 0, mapq=60, cigar=cig, seq="".join(seq)))
    if iupac_rate:   # ambiguity codes in the reads (M is also the reference's internal marker character); own generator,
        rng2 = random.Random(seed * 7919 + 1)   # so the cases without them keep their sequences
        for r in reads:
            s = list(r["seq"])
            for i in range(len(s)):
                if rng2.random() < iupac_rate:
                    s[i] = rng2.choice("MMMRN")
            r["seq"] = "".join(s)
    return contigs, reads


def make_sv_case(seed, L=120000, depth=20, mean_len=8000, sv=((40000, 1500), (85000, 900)), sub=0.03, ins=0.02, dele=0.02,
                 max_indel=2, name="ctg0", hole=None):
    """One contig whose TRUE sequence carries extra blocks (position in the draft, block length) the draft lacks.  A read
    spanning such a block is reported the way a long-read mapper does: a primary alignment of its longer side with the
    rest soft-clipped, a supplementary alignment (flag 0x800, hard-clipped) of the other side, and SA tags on both.
    Returns (contigs, reads, aux) for Stream.from_reads / write_files(aux=...)."""
    rng = random.Random(seed)
    draft = "".join(rng.choice("ACGT") for _ in range(L))
    blocks = {p: "".join(rng.choice("ACGT") for _ in range(k)) for p, k in sv}

    def noisy(lo, hi):
        """alignment of draft[lo:hi) with errors: returns (cigar ops list, read string); starts and ends with a match"""
        ops, seq, pos = [], [], lo

        def push(op, n=1):
            if ops and ops[-1][0] == op:
                ops[-1][1] += n
            else:
                ops.append([op, n])
        while pos < hi:
            edge = pos == lo or pos >= hi - 1
            x = rng.random()
            if not edge and x < dele:
                n = min(rng.randint(1, max_indel), hi - 1 - pos)
                if n > 0:
                    push("D", n)
                    pos += n
                    continue
            if not edge and x < dele + ins:
                n = rng.randint(1, max_indel)
                push("I", n)
                seq += [rng.choice("ACGT") for _ in range(n)]
            c = draft[pos]
            if rng.random() < sub:
                c = rng.choice([b for b in "ACGT" if b != c])
            push("M", 1)
            seq.append(c)
            pos += 1
        return [(o, n) for o, n in ops], "".join(seq)

    def cigar_str(cig):
        return "".join("%d%s" % (n, o) for o, n in cig)

    recs = []   # (pos, read dict, aux bytes)
    n_reads = max(1, int(depth * L / mean_len))
    for _ in range(n_reads):
        st = rng.randrange(0, L - 1000)
        en = min(L, st + max(1500, int(rng.lognormvariate(0, 0.4) * mean_len)))
        if hole and st < hole[1] and en > hole[0]:   # a stretch no read crosses: a low-depth region for the split logic
            if st < hole[0] - 1500:
                en = hole[0]
            elif en > hole[1] + 1500:
                st = hole[1]
            else:
                continue
        flag = 16 if rng.random() < 0.5 else 0
        strand = "-" if flag & 16 else "+"
        cut = next((p for p in sorted(blocks) if st + 800 < p < en - 800), None)
        if cut is None:
            cig, seq = noisy(st, en)
            recs.append((st, dict(ctg=0, pos=st, flag=flag, mapq=60, cigar=cig, seq=seq), b""))
            continue
        blk = blocks[cut]
        blk = "".join(c if rng.random() > sub else rng.choice("ACGT") for c in blk)
        c1, s1 = noisy(st, cut)
        c2, s2 = noisy(cut, en)
        full = s1 + blk + s2
        left_primary = len(s1) >= len(s2)
        # primary keeps the whole read (soft clip), the supplementary is hard-clipped
        if left_primary:
            pc, pseq, ppos = c1 + [("S", len(blk) + len(s2))], full, st
            sc, sseq, spos = [("H", len(s1) + len(blk))] + c2, s2, cut
            p_sa = cigar_str([("S", len(s1) + len(blk))] + c2)
            s_sa = cigar_str(pc)
        else:
            pc, pseq, ppos = [("S", len(s1) + len(blk))] + c2, full, cut
            sc, sseq, spos = c1 + [("H", len(blk) + len(s2))], s1, st
            p_sa = cigar_str(c1 + [("S", len(blk) + len(s2))])
            s_sa = cigar_str(pc)
        sa_for_primary = ("%s,%d,%s,%s,60,10;" % (name, spos + 1, strand, p_sa)).encode()
        sa_for_supp = ("%s,%d,%s,%s,60,10;" % (name, ppos + 1, strand, s_sa)).encode()
        recs.append((ppos, dict(ctg=0, pos=ppos, flag=flag, mapq=60, cigar=pc, seq=pseq), b"SAZ" + sa_for_primary + b"\0"))
        recs.append((spos, dict(ctg=0, pos=spos, flag=flag | 0x800, mapq=60, cigar=sc, seq=sseq), b"SAZ" + sa_for_supp + b"\0"))
    recs.sort(key=lambda r: r[0])
    return [(name, draft)], [r[1] for r in recs], [r[2] for r in recs]
