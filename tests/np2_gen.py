"""Synthetic long-read workloads for the nextpolish2 path (test helper): a random draft and noisy reads with known
CIGARs, written as FASTA + coordinate-sorted BAM + BAI through the library's own writer."""
import random


def make_case(seed, contig_lens=(20000,), depth=20, mean_len=4000, sub=0.03, ins=0.02, dele=0.02, max_indel=2,
              clip_rate=0.2, lower=False, n_rate=0.0, name_prefix="ctg"):
    rng = random.Random(seed)
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
    return contigs, reads
