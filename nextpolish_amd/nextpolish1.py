#!/usr/bin/env python
"""Host-side mirror of the reference's per-block polish driver (reference: source/lib/nextpolish1.py).

Same command line, same block-file / resume / naming behaviour, same output records
(``>name_np1 length`` + sequence); the compute goes through the in-tree HIP library instead of a
``multiprocessing.Pool`` of CPU workers:

* tasks 1, 2 and 4 (score_chain, kmer_count, snp_valid): the still-unpolished contigs of this block are packed into batches of
  ``--batch_bp`` draft bases that flow loader threads -> pinned host arrays -> H2D -> kernels -> D2H on ``--lanes``
  device lanes (include/nextpolish1.h, np1_pipe_*).  ``--world N --rank r`` deals the block's contigs over N
  GPUs of the node longest-first, one process per GPU (contigs are independent: no collective on the data path;
  reference: nextpolish1.py:181-189,223-224 and source/nextPolish:93-117 for the block split).
* ``-debug`` needs the per-base change list of the drop-in ABI and therefore goes contig by contig through
  ``score_chain(tigname, cfg)`` exactly like the reference worker (nextpolish1.py:181-189).
* task 3 (snp_phase: short reads + long reads): batches of ``--batch_bp`` draft bases, per batch two resident record batches and one
  ``np1_batch_snp_phase`` pass (np1_phase_device.hip); with ``-debug`` contig by contig through the drop-in symbol.  Task 5 reports that it is
  not available (the reference's own caller refuses it).

Record order is the order of the block file / FASTA (the reference's order is nondeterministic: it
iterates a Python set through imap_unordered, nextpolish1.py:148-161,224).
"""
from __future__ import print_function

import argparse
import ctypes as C
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
from nextpolish_amd import _native as nat  # noqa: E402
from nextpolish_amd import resume  # noqa: E402
from nextpolish_amd.shard import deal_contigs, fasta_lengths  # noqa: E402


def parse_num_unit(s):
    """reference: source/lib/kit.py parse_num_unit ('150k' -> 150000)."""
    s = str(s).strip().lower()
    mult = {"k": 1000, "m": 1000000, "g": 1000000000}
    if s and s[-1] in mult:
        return int(float(s[:-1]) * mult[s[-1]])
    return int(float(s))


def read_polished_seqs(infile, polished_seqs):
    """Resume support (reference: nextpolish1.py:163-179): adds the contigs `infile` already holds to `polished_seqs`, except the last
    one (its record may be cut off); returns that record's offset for the rewrite (nextpolish_amd/resume.py)."""
    return resume.finished_contigs(infile, resume.polished_record, polished_seqs)


def read_unpolished_seqs(infile, index, polished_seqs, keep_polished=False):
    """reference: nextpolish1.py:148-161, but order preserving.  keep_polished=True returns the block's full list (what the
    ranks of a node are dealt from, so that the deal does not depend on how far each rank got before a restart)."""
    names = []
    if index != "all":
        with open(infile) as IN:
            for line in IN:
                lines = line.strip().split()
                if lines and (keep_polished or lines[0].split("_np")[0] not in polished_seqs) and lines[1] == index:
                    names.append(lines[0])
    else:
        with open(infile) as IN:
            for line in IN:
                if line.startswith(">"):
                    names.append(line.strip().split()[0][1:])
    seen, out = set(), []
    for n in names:
        if n not in seen:
            seen.add(n)
            out.append(n)
    return out


def output_name(seq_pname, task):
    """reference: nextpolish1.py:228 -- 'x' -> 'x_np1'; 'x_np1' -> 'x_np12'."""
    return seq_pname + (str(task) if seq_pname.split("_")[-1].startswith("np") else ("_np" + str(task)))


def update_cfg(cfg, args):
    """reference: nextpolish1.py:102-133 (the struct is mutated in place after config_init)."""
    c = cfg.contents
    c.trim_len_edge = args.trim_len_edge
    c.ext_len_edge = args.ext_len_edge
    c.min_map_quality = args.min_map_quality
    c.indel_balance_factor_sgs = args.indel_balance_factor_sgs
    c.min_count_ratio_skip = args.min_count_ratio_skip
    c.min_len_ldr = args.min_len_ldr
    c.min_len_inter_kmer = args.min_len_inter_kmer
    c.max_len_kmer = args.max_len_kmer
    c.max_count_kmer = args.max_count_kmer
    c.min_depth_snp = args.min_depth_snp
    c.min_count_snp = args.min_count_snp
    c.min_count_snp_link = args.min_count_snp_link
    c.ploidy = args.ploidy
    c.indel_balance_factor_lgs = args.indel_balance_factor_lgs
    c.max_indel_factor_lgs = args.max_indel_factor_lgs
    c.max_snp_factor_lgs = args.max_snp_factor_lgs
    c.min_snp_factor_sgs = args.min_snp_factor_sgs
    c.region_count = 10000
    c.count_read_ins_sgs = args.count_read_ins_sgs
    c.max_ins_len_sgs = args.max_ins_len_sgs
    c.max_ins_fold_sgs = args.max_ins_fold_sgs
    c.max_variant_count_lgs = args.max_variant_count_lgs
    c.max_clip_ratio_sgs = args.max_clip_ratio_sgs
    c.max_clip_ratio_lgs = args.max_clip_ratio_lgs
    c.trace_polish_open = 1 if args.debug else 0


def plan_batches(names, lengths, max_bp):
    """Greedy in-order packing of contigs into batches of at most max_bp draft bases (a contig longer
    than that gets a batch of its own)."""
    batches, cur, cur_bp = [], [], 0
    for n in names:
        L = lengths[n]
        if cur and cur_bp + L > max_bp:
            batches.append(cur)
            cur, cur_bp = [], 0
        cur.append(n)
        cur_bp += L
    if cur:
        batches.append(cur)
    return batches


def rank_share(all_names, lengths, world, rank, polished_seqs, filter_polished):
    """This rank's contigs of the block in list order, minus those already in its own output (block mode only, like the
    reference: nextpolish1.py:154)."""
    owner = deal_contigs(all_names, lengths, world) if world > 1 else None
    out = []
    for n in all_names:
        if owner is not None and owner[n] != rank:
            continue
        if filter_polished and n.split("_np")[0] in polished_seqs:
            continue
        out.append(n)
    return out


def polish_batched(args, cfg, names, device, emit, shared=(), polished_seqs=()):
    """Tasks 1, 2 and 4 without -debug: the rank's contigs flow through the device in batches of --batch_bp draft bases on
    --lanes device lanes while host threads inflate and split the records of the next batches (np1_pipe_run_files).
    shared: contigs longer than --tile_bp whose tiles all ranks share (main); this rank writes its pieces first, polishes its own
    contigs, and joins the shared contigs it is the joiner of at the end."""
    lengths = fasta_lengths(args.genome)
    names = [n for n in names if n in lengths]
    if shared:
        from nextpolish_amd.device import Context
        token = tile_run_token(args.genome, args.bam_sgs, cfg, args.tile_bp, args.tile_halo, args.world, getattr(args, "launch_id", ""))
        stale = os.path.join(args.tile_dir, "FAILED.%d" % args.rank)
        if os.path.exists(stale):
            os.remove(stale)
        sctx = None
        try:
            sctx = Context(device)
            pieces = device_tile_pieces(sctx, args.genome, args.bam_sgs, cfg, args.tile_bp, args.tile_halo)
            for n in shared:
                write_tile_pieces(None, args.tile_dir, n, lengths[n], args.tile_bp, args.world, args.rank, token=token, pieces=pieces)
        except BaseException as e:
            mark_tile_failure(args.tile_dir, args.rank, token, "rank %d: %s" % (args.rank, e))
            raise
        finally:
            if sctx is not None:
                sctx.close()
    if names:
        polish_own_batched(args, cfg, names, lengths, device, emit)
    for i, n in enumerate(shared):
        if i % args.world == args.rank and not (args.block_index != "all" and n.split("_np")[0] in polished_seqs):      # (the filter of rank_share)
            emit(n, join_tile_pieces(args.tile_dir, n, lengths[n], args.tile_bp, args.tile_wait, token=token), [])


def polish_own_batched(args, cfg, names, lengths, device, emit):
    from nextpolish_amd.device import Pipe
    # --tile_bp (task 1): a contig longer than that is polished as independent tiles with a halo and joined exactly (np1_tile.cpp;
    # the reference takes contigs up to 2^31 bases in one score_chain call, source/nextPolish:101-102) -- it need not fit an HBM batch.
    # The runs of shorter contigs between such contigs go through the pipe as before, so the output keeps the order of `names`.
    tile_bp = args.tile_bp if args.task == 1 else 0
    pipe = Pipe(device, args.lanes)
    ctx = None
    try:
        run = []

        def flush():
            if run:
                pipe.run_files(args.genome, args.bam_sgs, names=run, batch_bp=args.batch_bp, cfg=cfg.contents, task=args.task,
                               sink=lambda name, seq: emit(name, seq, []))
                del run[:]
        for n in names:
            if tile_bp <= 0 or lengths[n] <= tile_bp:
                run.append(n)
                continue
            flush()
            if ctx is None:
                from nextpolish_amd.device import Context
                ctx = Context(device)
            emit(n, score_chain_tiled(ctx, args.genome, args.bam_sgs, n, cfg, tile_bp, args.tile_halo), [])
        flush()
    finally:
        if ctx is not None:
            ctx.close()
        pipe.close()


def score_chain_tiled(ctx, fasta, bam, name, cfg, tile_bp, halo_bp):
    """np1_score_chain_tiled (include/nextpolish1.h): one contig, every tile, on this process's GPU"""
    L = nat.lib()
    L.np1_score_chain_tiled.restype = C.c_int
    L.np1_score_chain_tiled.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(nat.Configure), C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    L.np1_free_string.argtypes = [C.c_void_p]
    out, n = C.c_void_p(), C.c_int64(0)
    if L.np1_score_chain_tiled(ctx.handle, fasta.encode(), bam.encode(), name.encode(), cfg, tile_bp, halo_bp, 0, 1, C.byref(out), C.byref(n), None) != 0:
        raise SystemExit("np1_score_chain_tiled: " + nat.last_error())
    seq = C.string_at(out, n.value).decode()
    L.np1_free_string(out)
    return seq


# ---- the tiles of one dominant contig over the ranks of a node (--world > 1 with --tile_bp) ------------------------
# Tiles are independent (np1_tile.cpp), so rank r polishes tiles r, r + world, ... of every contig longer than --tile_bp and leaves each
# piece as a file in --tile_dir (a directory all ranks see); the contig's joiner -- rank (its position among such contigs) mod world --
# concatenates the pieces in tile order and writes the record into its own -o part.  No collective, no process group: files, written
# under a temporary name and renamed.  Every rank writes ALL its pieces before any rank waits, so nobody waits for a rank that waits.

def tile_count(length, tile_bp):
    return (length + tile_bp - 1) // tile_bp


def tile_piece_dir(tile_dir, name):
    import hashlib
    return os.path.join(tile_dir, hashlib.md5(name.encode()).hexdigest())


def device_tile_pieces(ctx, fasta, bam, cfg, tile_bp, halo_bp):
    """pieces(name, first, stride) -> [piece of tile first, first + stride, ...]: ONE np1_tiler_run per contig and rank -- the contig's draft, its
    FASTA index entry and the BAM index are read once (np1_tile.cpp), the records of tile t + 1 are read while the device runs tile t"""
    L = nat.lib()
    L.np1_tiler_open.restype = C.c_void_p
    L.np1_tiler_open.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.np1_tiler_close.argtypes = [C.c_void_p]
    L.np1_tiler_run.restype = C.c_int
    L.np1_tiler_run.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(nat.Configure), C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p),
                                C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    L.np1_free_string.argtypes = [C.c_void_p]

    def pieces(name, first, stride, n_tiles):
        mine = len(range(first, n_tiles, stride))
        if mine == 0:
            return []
        t = L.np1_tiler_open(fasta.encode(), bam.encode(), name.encode())
        if not t:
            raise SystemExit("np1_tiler_open: " + nat.last_error())
        try:
            out, n = C.c_void_p(), C.c_int64(0)
            plen = (C.c_int64 * mine)()
            if L.np1_tiler_run(t, ctx.handle, cfg, tile_bp, halo_bp, first, stride, C.byref(out), C.byref(n), plen, None) != 0:
                raise SystemExit("np1_tiler_run: " + nat.last_error())
            seq = C.string_at(out, n.value).decode()
            L.np1_free_string(out)
        finally:
            L.np1_tiler_close(t)
        res, at = [], 0
        for k in range(mine):
            res.append(seq[at:at + plen[k]])
            at += plen[k]
        assert at == len(seq)
        return res
    return pieces


def tile_run_token(genome, bam, cfg, tile_bp, halo_bp, world, launch_id=""):
    """what the pieces of one launch have in common and the leftovers of another launch have not: the input files as they are now (size,
    mtime), every parameter of the task and the tiling.  Pieces carry it in their first line; the joiner takes no piece without it (a
    directory left by a killed run on other inputs or parameters is not stitched into this run's output: ADVICE r4)."""
    import hashlib
    h = hashlib.md5()
    for path in (genome, bam):
        st = os.stat(path)
        h.update(("%s:%d:%d;" % (os.path.abspath(path), st.st_size, st.st_mtime_ns)).encode())
    c = cfg.contents if hasattr(cfg, "contents") else cfg
    for f, _t in c._fields_:
        v = getattr(c, f)
        if isinstance(v, (int, float)):
            h.update(("%s=%r;" % (f, v)).encode())
    h.update(("%d:%d:%d" % (tile_bp, halo_bp, world)).encode())
    # --launch_id: what the ranks of ONE launch share and a re-run of the same command does not (the token above is a pure function
    # of inputs and parameters, so a rank of the re-run that reaches the join before its slower peer has restarted would take the peer's
    # FAILED marker -- or a piece -- of the failed launch for this launch's)
    h.update(("|" + str(launch_id)).encode())
    return h.hexdigest()


def write_tile_pieces(piece, tile_dir, name, length, tile_bp, world, rank, token="", pieces=None):
    """this rank's tiles of one contig, each left as <tile_dir>/<md5 of the name>/<k>.seq (first line: the run token).  piece(name, k, n)
    makes one piece; pieces(name, first, stride, n) -- when given -- makes all of this rank's in one call."""
    n = tile_count(length, tile_bp)
    d = tile_piece_dir(tile_dir, name)
    os.makedirs(d, exist_ok=True)
    mine = list(range(rank, n, world))
    seqs = pieces(name, rank, world, n) if pieces is not None else (piece(name, k, n) for k in mine)
    for k, seq in zip(mine, seqs):
        tmp = os.path.join(d, "%d.tmp.%d" % (k, os.getpid()))
        with open(tmp, "w") as f:
            f.write(token + "\n")
            f.write(seq)
        os.replace(tmp, os.path.join(d, "%d.seq" % k))


def mark_tile_failure(tile_dir, rank, token, why):
    """a rank that cannot deliver its pieces says so, so that the joiners stop waiting for it"""
    try:
        os.makedirs(tile_dir, exist_ok=True)
        with open(os.path.join(tile_dir, "FAILED.%d" % rank), "w") as f:
            f.write(token + "\n" + why + "\n")
    except OSError:
        pass


def failed_ranks(tile_dir, token):
    import glob
    bad = []
    for p in glob.glob(os.path.join(tile_dir, "FAILED.*")):
        try:
            with open(p) as f:
                if f.readline().rstrip("\n") == token:
                    bad.append((p, f.read().strip()))
        except OSError:
            pass
    return bad


def join_tile_pieces(tile_dir, name, length, tile_bp, wait_s, poll_s=0.2, token=""):
    """the polished contig from the pieces of all ranks, in tile order; waits up to wait_s seconds for each missing piece (a piece of another
    launch -- wrong token -- counts as missing: its owner will replace it), gives up at once when a rank of this launch reported a failure"""
    import shutil
    import time
    n = tile_count(length, tile_bp)
    d = tile_piece_dir(tile_dir, name)
    parts = []
    for k in range(n):
        path = os.path.join(d, "%d.seq" % k)
        t0 = time.time()
        while True:
            if os.path.exists(path):
                with open(path) as f:
                    head = f.readline().rstrip("\n")
                    if head == token:
                        parts.append(f.read())
                        break
            bad = failed_ranks(tile_dir, token)
            if bad:
                raise SystemExit("tile %d of %s will not arrive: %s" % (k, name, "; ".join("%s: %s" % b for b in bad)))
            if time.time() - t0 > wait_s:
                raise SystemExit("tile %d of %s did not arrive in %s within %d s (is every rank of --world running with the same --tile_dir?)" % (k, name, d, wait_s))
            time.sleep(poll_s)
    shutil.rmtree(d, ignore_errors=True)
    return "".join(parts)


def shared_tile_contigs(all_names, lengths, tile_bp):
    """contigs of the block every rank takes tiles of, in block order"""
    return [n for n in all_names if lengths.get(n, 0) > tile_bp]


def polish_phase_batched(args, cfg, names, device, emit):
    """Task 3 without -debug: the rank's contigs in batches of --batch_bp draft bases; per batch the short reads come through the
    device-side ingest, the long reads through the host loader, then one np1_batch_snp_phase pass (np1_pipe_run_phase_files;
    reference: one snp_phase(tigname, cfg) call per contig and worker, source/lib/nextpolish1.py:95-96,181-189)."""
    from nextpolish_amd.device import Pipe
    lengths = fasta_lengths(args.genome)
    names = [n for n in names if n in lengths]
    if not names:
        return
    if not args.bam_sgs or not args.bam_lgs:
        raise SystemExit("task 3 needs both -s (short-read BAM) and -l (long-read BAM)")
    pipe = Pipe(device, 1)
    try:
        pipe.run_phase_files(args.genome, args.bam_sgs, args.bam_lgs, names=names, batch_bp=args.batch_bp, cfg=cfg.contents,
                             sink=lambda name, seq: emit(name, seq, []))
    finally:
        pipe.close()


def polish_per_contig(args, cfg, names, fun, emit):
    L = nat.lib()
    for name in names:
        r = fun(name.encode(), cfg)
        seq = C.string_at(r.contents.contig).decode()
        pts = [(r.contents.data[p].pos, r.contents.data[p].index, r.contents.data[p].curbase.decode(),
                r.contents.data[p].base.decode()) for p in range(r.contents.datalength)]
        L.polishresult_destory(r)
        emit(name, seq, pts)


def main(args):
    OUT = sys.stdout
    polished_seqs = set()
    if args.out != "stdout":
        if os.path.exists(args.out):
            last_seq_position = read_polished_seqs(args.out, polished_seqs)
            OUT = open(args.out, "r+")
            OUT.seek(last_seq_position, os.SEEK_SET)
            OUT.truncate()
        else:
            OUT = open(args.out, "w")
    blockfile = args.block
    if args.block_index == "all" or not args.block:
        args.block_index = "all"
        blockfile = args.genome
    all_names = read_unpolished_seqs(blockfile, args.block_index, polished_seqs, keep_polished=True)
    shared = []
    if args.world > 1 and args.task == 1 and args.tile_bp > 0 and not args.debug:      # tiles of dominant contigs are shared by all ranks
        lens_all = fasta_lengths(args.genome)
        shared = shared_tile_contigs(all_names, lens_all, args.tile_bp)
        if shared and not args.tile_dir:
            args.tile_dir = args.genome + ".np1_tiles"
        held = set(shared)
        all_names = [n for n in all_names if n not in held]
    names = rank_share(all_names, fasta_lengths(args.genome) if args.world > 1 else {}, args.world, args.rank, polished_seqs,
                       args.block_index != "all")

    L = nat.lib()
    cfg = L.config_init(args.genome.encode(), (args.bam_sgs or "").encode() or None,
                        (args.bam_lgs or "").encode() or None)
    update_cfg(cfg, args)

    def emit(name, seq, pts):
        if args.uppercase:
            seq = seq.upper()
        print(">%s %d\n%s" % (output_name(name, args.task), len(seq), seq), file=OUT)
        for p in pts:
            print(name + " %d %d %c %c" % p, file=sys.stderr)

    fun = {1: L.score_chain, 2: L.kmer_count, 3: L.snp_phase, 4: L.snp_valid, 5: L.lgspolish}[args.task]
    if args.task in (1, 2, 4) and not args.debug:
        device = args.device if args.device >= 0 else args.rank
        polish_batched(args, cfg, names, device, emit, shared, polished_seqs)
    elif args.task == 3 and not args.debug:
        polish_phase_batched(args, cfg, names, args.device if args.device >= 0 else args.rank, emit)
    else:
        polish_per_contig(args, cfg, names, fun, emit)
    if args.out != "stdout":
        OUT.close()
    L.config_destory(cfg)


def build_parser():
    p = argparse.ArgumentParser(description="Polish the genome on MI355X GPUs (drop-in for lib/nextpolish1.py).")
    io = p.add_argument_group("Input/Output arguments")
    io.add_argument("-g", "--genome", metavar="FILE", required=True, type=str)
    io.add_argument("-s", "--bam_sgs", metavar="FILE", type=str)
    io.add_argument("-l", "--bam_lgs", metavar="FILE", type=str)
    io.add_argument("-b", "--block", metavar="FILE", type=str)
    io.add_argument("-i", "--block_index", type=str, default="all")
    io.add_argument("-u", "--uppercase", action="store_true", default=False)
    io.add_argument("-debug", action="store_true", default=False)
    io.add_argument("-o", "--out", metavar="FILE", default="stdout")
    alg = p.add_argument_group("Algorithm arguments")
    alg.add_argument("-t", "--task", metavar="N", type=int, required=True, choices=[1, 2, 3, 4, 5])
    alg.add_argument("-p", "--process", metavar="N", type=int, default=10,
                     help="accepted for compatibility; contig-level parallelism lives on the GPU")
    alg.add_argument("-count_read_ins_sgs", metavar="N", type=int, default=10000)
    alg.add_argument("-min_map_quality", metavar="N", type=int, default=0)
    alg.add_argument("-max_ins_len_sgs", metavar="N", type=int, default=10000)
    alg.add_argument("-max_ins_fold_sgs", metavar="N", type=int, default=5)
    alg.add_argument("-max_clip_ratio_sgs", metavar="F", type=float, default=0.15)
    alg.add_argument("-max_clip_ratio_lgs", metavar="F", type=float, default=0.4)
    alg.add_argument("-trim_len_edge", metavar="N", type=int, default=2)
    alg.add_argument("-ext_len_edge", metavar="N", type=int, default=2)
    sc = p.add_argument_group("score_chain")
    sc.add_argument("-indel_balance_factor_sgs", metavar="F", type=float, default=0.5)
    sc.add_argument("-min_count_ratio_skip", metavar="F", type=float, default=0.8)
    kc = p.add_argument_group("kmer_count")
    kc.add_argument("-min_len_ldr", metavar="N", type=int, default=3)
    kc.add_argument("-max_len_kmer", metavar="N", type=int, default=50)
    kc.add_argument("-min_len_inter_kmer", metavar="N", type=int, default=5)
    kc.add_argument("-max_count_kmer", metavar="N", type=int, default=50)
    sp = p.add_argument_group("snp_phase")
    sp.add_argument("-ploidy", metavar="N", type=int, default=2)
    sp.add_argument("-max_variant_count_lgs", metavar="N", type=str, default="150k")
    sp.add_argument("-indel_balance_factor_lgs", metavar="F", type=float, default=0.33)
    sp.add_argument("-min_depth_snp", metavar="N", type=int, default=3)
    sp.add_argument("-min_count_snp", metavar="N", type=int, default=5)
    sp.add_argument("-min_count_snp_link", metavar="N", type=int, default=5)
    sp.add_argument("-max_indel_factor_lgs", metavar="F", type=float, default=0.21)
    sp.add_argument("-max_snp_factor_lgs", metavar="F", type=float, default=0.53)
    sp.add_argument("-min_snp_factor_sgs", metavar="F", type=float, default=0.34)
    gpu = p.add_argument_group("GPU placement (this implementation)")
    gpu.add_argument("--device", type=int, default=-1, help="HIP device of this process (default: its rank)")
    gpu.add_argument("--rank", type=int, default=int(os.environ.get("LOCAL_RANK", "0")))
    gpu.add_argument("--world", type=int, default=int(os.environ.get("WORLD_SIZE", "1")),
                     help="number of processes sharing this block (one per GPU); each writes its own -o part")
    gpu.add_argument("--batch_bp", type=parse_num_unit, default=parse_num_unit("16m"),
                     help="draft bases per HBM-resident batch (a longer contig is a batch of its own)")
    gpu.add_argument("--lanes", type=int, default=2, help="batches in flight on the device (HIP stream + host thread each)")
    gpu.add_argument("--tile_bp", type=parse_num_unit, default=0,
                     help="task 1: polish contigs longer than this many bases as independent tiles of that size, joined exactly (0 = off)")
    gpu.add_argument("--tile_halo", type=parse_num_unit, default=1000, help="bases of halo on each side of a tile (doubled when too small)")
    gpu.add_argument("--tile_dir", type=str, default="",
                     help="--world > 1 with --tile_bp: directory all ranks see, where the pieces of contigs longer than --tile_bp meet (default: <genome>.np1_tiles)")
    gpu.add_argument("--launch_id", default="", help="any string all ranks of ONE launch share (e.g. the scheduler's job id): pieces and failure markers of another "
                                                      "launch in --tile_dir, even of the same command run again, are then never taken for this launch's")
    gpu.add_argument("--tile_wait", type=int, default=7200,
                     help="seconds the joiner of such a contig waits for a piece of another rank (a rank that fails says so and ends the wait at once)")
    return p


if __name__ == "__main__":
    a, _unknown = build_parser().parse_known_args()
    a.max_variant_count_lgs = parse_num_unit(a.max_variant_count_lgs)
    if a.task == 5:
        sys.stderr.write("Please use nextpolish2 to polish the genome with long reads.\n")
        sys.exit(1)
    main(a)
