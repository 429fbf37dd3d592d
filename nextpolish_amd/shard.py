"""Contig -> rank assignment shared by the two callers (nextpolish1.py, nextpolish2.py); no native code involved."""
import os


def deal_contigs(names, lengths, world):
    """Contig -> rank, longest first onto the least loaded rank (ties: lower rank; equal lengths: list order).  A pure function
    of the block's FULL name list and the contig lengths: independent of what any rank already wrote, so a restarted rank
    gets the same contigs again.  Returns {name: rank}.  (The reference's own unit of distribution is the contig block,
    filled in .fai order: source/nextPolish:93-117; longest-first balances a node whose block holds one giant contig.)"""
    order = sorted(range(len(names)), key=lambda k: (-lengths.get(names[k], 0), k))
    load = [0] * world
    owner = {}
    for k in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[names[k]] = r
        load[r] += max(1, lengths.get(names[k], 0))
    return owner


def fasta_lengths(genome):
    """Contig lengths from <genome>.fai, or from the FASTA itself when the index does not exist yet."""
    lens = {}
    fai = genome + ".fai"
    if os.path.exists(fai):
        with open(fai) as IN:
            for line in IN:
                f = line.rstrip("\n").split("\t")
                lens[f[0]] = int(f[1])
        return lens
    name = None
    with open(genome) as IN:
        for line in IN:
            if line.startswith(">"):
                name = line[1:].split()[0]
                lens[name] = 0
            elif name is not None:
                lens[name] += len("".join(line.split()))
    return lens
