#!/usr/bin/env python
"""Host-side mirror of the reference's long-read polish driver (reference: source/lib/nextpolish2.py).

Same command line (``-g -l -r -b -i -o -p -u -w -a -sp -id -as``), same block-file / resume / naming behaviour and the
same output records (``>name len`` or ``>name_s<i> len`` when a contig was split, nextpolish2.py:139-151,198-202);
the consensus goes through the in-tree HIP library ``lib/nextpolish2.so`` instead of the CPU library.

Process model = the reference's: ``read_ref`` and ``ctg_cns_init`` run in the parent, the worker pool is forked
afterwards (nextpolish2.py:184-194) and every worker creates its HIP context lazily on its first contig; worker k uses
GPU ``pid mod n_gpus`` (``NP2_DEVICE`` pins it).  ``--world N --rank r`` additionally shards the contigs of the
block over N node-level processes (dealt longest-first from the block's full list, nextpolish_amd/shard.py: the deal does
not depend on what a rank already wrote, so a restart resumes the same share): contigs are independent, so no collective is involved.  Records are written in completion order, like the reference (``imap_unordered``).
"""
from __future__ import print_function

import argparse
import ctypes as C
import os
import sys
from multiprocessing import Pool

HERE = os.path.dirname(os.path.realpath(__file__))
if __package__ in (None, ""):      # run as a script (the way the reference's workflow starts its callers)
    sys.path.insert(0, os.path.dirname(HERE))
from nextpolish_amd import resume  # noqa: E402


class ConsensusTrimed(C.Structure):   # reference: nextpolish2.py:21-26
    _fields_ = [("len", C.c_uint), ("identity", C.c_float), ("seq", C.c_char_p)]


class ConsensusTrimedData(C.Structure):   # nextpolish2.py:28-32
    _fields_ = [("data", C.POINTER(ConsensusTrimed)), ("i_m", C.c_int)]


class Ref(C.Structure):   # nextpolish2.py:37-44
    _fields_ = [("n", C.c_char_p), ("s", C.POINTER(C.c_uint32)), ("qv", C.c_void_p), ("qv_l", C.c_uint32), ("length", C.c_uint32)]


class Refs(C.Structure):   # nextpolish2.py:46-51
    _fields_ = [("ref", C.POINTER(Ref)), ("i", C.c_uint32), ("i_m", C.c_uint32)]


def load_library(path=None):
    P = C.CDLL(path or os.path.join(HERE, "lib", "nextpolish2.so"))
    P.read_ref.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
    P.read_ref.restype = C.POINTER(Refs)
    P.refs_destroy.argtypes = [C.POINTER(Refs)]
    P.ctg_cns_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    P.ctg_cns_init.restype = C.c_void_p
    P.ctg_cns_destroy.argtypes = [C.c_void_p]
    P.ctg_cns_core.argtypes = [C.c_void_p, C.POINTER(Ref), C.c_char_p]
    P.ctg_cns_core.restype = C.POINTER(ConsensusTrimedData)
    P.free_consensus_trimed_data.argtypes = [C.POINTER(ConsensusTrimedData)]
    return P


def parse_num_unit(s):
    """reference: source/lib/kit.py parse_num_unit ('5M' -> 5000000)."""
    s = str(s).strip().lower()
    mult = {"k": 1000, "m": 1000000, "g": 1000000000}
    if s and s[-1] in mult:
        return int(float(s[:-1]) * mult[s[-1]])
    return int(float(s))


def read_corrected_seqs(infile, corrected_seqs):
    """Resume support (nextpolish2.py:116-137): adds the contigs `infile` already holds to `corrected_seqs`, except the one the file ends
    in (it may be incomplete); returns the offset of that contig's first record (nextpolish_amd/resume.py)."""
    return resume.finished_contigs(infile, resume.corrected_piece, corrected_seqs)


def read_uncorrected_seqs(infile, index, corrected_seqs, keep_corrected=False):
    """nextpolish2.py:97-115: contigs of block `index` of a block file, or every FASTA header when index == 'all'.
    Returned in file order (the reference keeps a set).  keep_corrected=True: the block's full list (the ranks of a node are
    dealt from it, so the deal does not move when a rank restarts with part of its output already written)."""
    names = []
    with open(infile) as IN:
        for line in IN:
            if index != "all":
                f = line.strip().split()
                if f and (keep_corrected or f[0] not in corrected_seqs) and f[1] == index:
                    names.append(f[0])
            elif line.startswith(">"):
                n = line.strip().split()[0][1:]
                if keep_corrected or n not in corrected_seqs:
                    names.append(n)
    return names


def set_window_process(args):
    """nextpolish2.py:67-79 (host RAM bound of the reference; the device path needs ~60 B per alignment column of a window in HBM, so the
    288 GB of an MI355X are not the limit at the default 5 Mb window).  Unlike the reference this never leaves zero workers."""
    import psutil
    args.window, workers = resume.fit_workers(args.window, args.process, psutil.virtual_memory().available, psutil.cpu_count())
    args.process = max(1, workers)


_P = _CFG = _REFS = None


def _worker(job):
    n, bam_list = job
    r = _REFS.contents.ref[n]
    c_seq = _P.ctg_cns_core(_CFG, C.byref(r), bam_list)
    name = C.string_at(r.n).decode()
    out = []
    for i in range(c_seq.contents.i_m):
        seq = C.string_at(c_seq.contents.data[i].seq).decode()
        seq_len = int(c_seq.contents.data[i].len)
        out.append([name + ("_s%d" % i if c_seq.contents.i_m != 1 else ""), seq, seq_len])
    _P.free_consensus_trimed_data(c_seq)
    return out


def main(args):
    global _P, _CFG, _REFS
    OUT = sys.stdout
    corrected_seqs = set()
    if args.out != "stdout":
        if os.path.exists(args.out):
            last_seq_position = read_corrected_seqs(args.out, corrected_seqs)
            OUT = open(args.out, "r+")
            OUT.seek(last_seq_position, os.SEEK_SET)
            OUT.truncate()
        else:
            OUT = open(args.out, "w")
    blockfile = args.block
    if args.block_index == "all" or not args.block:
        args.block_index = "all"
        blockfile = args.genome
    names = read_uncorrected_seqs(blockfile, args.block_index, corrected_seqs, keep_corrected=args.world > 1)
    if args.world > 1:
        # deal on the block's full list (longest first, order-stable), only then drop what this rank already wrote
        sys.path.insert(0, os.path.dirname(HERE))
        from nextpolish_amd.shard import deal_contigs, fasta_lengths
        owner = deal_contigs(names, fasta_lengths(args.genome), args.world)
        names = [n for n in names if owner[n] == args.rank and n not in corrected_seqs]
    if not names:
        return 0
    _P = load_library(args.library)
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    _REFS = _P.read_ref(args.genome.encode(), arr, len(names))
    if args.auto:
        set_window_process(args)
    _CFG = _P.ctg_cns_init(args.window, args.read_type, args.split, args.alignment_identity_ratio, args.alignment_score_ratio,
                           args.alignment_score_ratio)
    jobs = [(i, args.bam_list.encode()) for i in range(_REFS.contents.i)]
    pool = Pool(max(1, args.process)) if args.process > 1 and len(jobs) > 1 else None
    results = pool.imap_unordered(_worker, jobs, chunksize=1) if pool else map(_worker, jobs)
    rc = 0
    for seq_data in results:
        for seq_name, seq, seq_len in seq_data:
            if args.uppercase:
                seq = seq.upper()
            if seq_len > 10:
                print(">%s %d\n%s" % (seq_name, seq_len, seq), file=OUT)
            else:
                sys.stderr.write("Failed to correct sequence: %s\n" % seq_name)   # nextpolish2.py:198-202
                rc = 1
        if rc:
            break
    if pool:
        pool.close()
        pool.join()
    if args.out != "stdout":
        OUT.close()
    _P.ctg_cns_destroy(_CFG)
    _P.refs_destroy(_REFS)
    return rc


def build_parser():
    p = argparse.ArgumentParser(description="Correct structural & base errors in the genome with long reads on MI355X GPUs "
                                            "(drop-in for lib/nextpolish2.py).")
    p.add_argument("-g", "--genome", metavar="FILE", required=True, type=str)
    p.add_argument("-l", "--bam_list", metavar="FILE", required=True, type=str)
    p.add_argument("-r", "--read_type", required=True, type=str.lower, choices=["clr", "hifi", "ont"])
    p.add_argument("-b", "--block", metavar="FILE", type=str)
    p.add_argument("-i", "--block_index", type=str, default="all")
    p.add_argument("-o", "--out", metavar="FILE", default="stdout")
    p.add_argument("-p", "--process", metavar="INT", type=int, default=10)
    p.add_argument("-u", "--uppercase", action="store_true", default=False)
    p.add_argument("-w", "--window", metavar="STR", type=str, default="5M")
    p.add_argument("-a", "--auto", action="store_false", default=True)
    p.add_argument("-sp", "--split", action="store_false", default=True)
    p.add_argument("-id", "--alignment_identity_ratio", metavar="FLOAT", type=float, default=0.8)
    p.add_argument("-as", "--alignment_score_ratio", metavar="FLOAT", type=float, default=0.8)
    p.add_argument("--world", type=int, default=1, help="node-level processes sharing the block (one per GPU)")
    p.add_argument("--rank", type=int, default=0)
    p.add_argument("--library", default=None, help=argparse.SUPPRESS)
    return p


if __name__ == "__main__":
    a, _unknown = build_parser().parse_known_args()
    a.window = parse_num_unit(a.window)
    a.split = 1 if a.split else 0
    a.read_type = {"ont": 1, "clr": 2, "hifi": 3}[a.read_type]
    sys.exit(main(a))
