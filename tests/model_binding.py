"""ctypes binding of tests/model/libnp1_model.so (host lockstep model of the HIP launch sequence)."""
import ctypes as C
import os

from nextpolish_amd import _native as nat

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model", "libnp1_model.so"))
        L.np1m_score_chain.argtypes = [C.POINTER(nat.StreamView), C.POINTER(nat.Configure), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.np1m_score_chain.restype = C.c_int
        L.np1m_free.argtypes = [C.c_void_p]
        L.np1m_kmer_count.argtypes = [C.POINTER(nat.StreamView), C.POINTER(nat.Configure), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_uint32)]
        L.np1m_kmer_count.restype = C.c_int
        L.np1m_snp_valid.argtypes = L.np1m_kmer_count.argtypes
        L.np1m_snp_valid.restype = C.c_int
        L.np1m_snp_phase.argtypes = [C.POINTER(nat.StreamView), C.POINTER(nat.StreamView), C.POINTER(nat.Configure), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_uint32)]
        L.np1m_snp_phase.restype = C.c_int
        L.np1m_kmer_count_replay.argtypes = [C.POINTER(nat.StreamView), C.POINTER(nat.Configure), C.c_char_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p,
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.np1m_kmer_count_replay.restype = C.c_int
        L.np1m_snp_valid_replay.argtypes = L.np1m_kmer_count_replay.argtypes
        L.np1m_snp_valid_replay.restype = C.c_int
        L.np1m_replay_revote_count.restype = C.c_ulonglong
        L.np1m_replay_break_count.restype = C.c_ulonglong
        _LIB = L
    return _LIB


def score_chain(stream, cfg=None, want_stats=False, fused=False):
    """fused=0: staged launch sequence (symbol rows); 1: descriptor-based sequence (k_desc + k_tile3); 2: k_desc + k_tile9 (np1_tile9.h),
    k_tile3 for the waves it hands back.  want_stats adds t9 = (agreeing pairs, deferred entries, lanes tallied by the whole wave, waves handed back)."""
    cfg = cfg or nat.default_config()
    C.c_int.in_dll(lib(), "np1m_fused").value = int(fused)   # 0 staged, 1 k_tile3, 2 k_tile9
    out = C.c_void_p()
    bounds = (C.c_uint32 * (stream.n_contigs + 1))()
    stats = (C.c_uint64 * 4)()
    rc = lib().np1m_score_chain(C.byref(stream.view), C.byref(cfg), C.byref(out), bounds, stats)
    if rc != 0:
        raise RuntimeError("model failed rc=%d" % rc)
    blob = C.string_at(out, bounds[stream.n_contigs])
    lib().np1m_free(out)
    res = [blob[bounds[i]:bounds[i + 1]].decode() for i in range(stream.n_contigs)]
    if want_stats:
        return res, dict(slots=stats[0], heads=stats[1], pool_words=stats[2], escalations=stats[3],
                         restarts=C.c_int.in_dll(lib(), "np1m_restarts").value,        # staged restarts (a record beyond the descriptors, > 160 contexts)
                         deep_chunks=C.c_int.in_dll(lib(), "np1m_deep_chunks").value,   # chunks voted with one list entry per possible context
                         t9=tuple((C.c_ulonglong * 4).in_dll(lib(), "np1m_t9_stats")))
    return res


def score_chain_tiled(stream, tile_bp, halo_bp, cfg=None, fused=False):
    """every contig cut into tiles of tile_bp bases, polished independently with a halo and joined (np1_model.cpp: the intra-contig
    tiling of DESIGN.md 8); returns (contigs, dict(tiles, recomputed, records))"""
    cfg = cfg or nat.default_config()
    C.c_int.in_dll(lib(), "np1m_fused").value = int(fused)
    out = C.c_void_p()
    bounds = (C.c_uint32 * (stream.n_contigs + 1))()
    ts = (C.c_uint64 * 3)()
    rc = lib().np1m_score_chain_tiled(C.byref(stream.view), C.byref(cfg), C.c_uint32(tile_bp), C.c_uint32(halo_bp), C.byref(out), bounds, ts)
    if rc != 0:
        raise RuntimeError("tiled model failed rc=%d" % rc)
    blob = C.string_at(out, bounds[stream.n_contigs])
    lib().np1m_free(out)
    return [blob[bounds[i]:bounds[i + 1]].decode() for i in range(stream.n_contigs)], dict(tiles=ts[0], recomputed=ts[1], records=ts[2])


def upload_roundtrip(stream):
    """the forms a stream crosses PCIe in (np1_upload.h), built by the product's builders and undone by host restatements of the device
    kernels; returns dict(seq_bytes, seq2_bytes, fields_bytes, compact_bytes, plain, full_positions) or raises naming the form that differs"""
    sizes = (C.c_uint64 * 6)()
    rc = lib().np1m_upload_roundtrip(C.byref(stream.view), sizes)
    if rc != 0:
        raise RuntimeError("upload form %d does not round-trip" % rc)
    return dict(seq_bytes=sizes[0], seq2_bytes=sizes[1], fields_bytes=sizes[2], compact_bytes=sizes[3], plain=sizes[4], full_positions=sizes[5])


def kmer_count(stream, cfg):
    """kmer_count through the per-region bodies of np1_kmer.h (stream must carry qualities; cfg.read_tlen set)."""
    out = C.c_void_p()
    bounds = (C.c_uint32 * (stream.n_contigs + 1))()
    rc = lib().np1m_kmer_count(C.byref(stream.view), C.byref(cfg), C.byref(out), bounds)
    if rc != 0:
        raise RuntimeError("kmer_count model failed rc=%d" % rc)
    blob = C.string_at(out, bounds[stream.n_contigs])
    lib().np1m_free(out)
    return [blob[bounds[i]:bounds[i + 1]].decode() for i in range(stream.n_contigs)]


def snp_valid(stream, cfg):
    """snp_valid (task 4) through the per-region bodies of np1_kmer.h, the rounds driven the way the kernels drive them; raises
    ValueError for inputs the reference has no defined result for (the product's ERR_KC_UNDEFINED)."""
    out = C.c_void_p()
    bounds = (C.c_uint32 * (stream.n_contigs + 1))()
    rc = lib().np1m_snp_valid(C.byref(stream.view), C.byref(cfg), C.byref(out), bounds)
    if rc == 512:
        raise ValueError("undefined upstream")
    if rc != 0:
        raise RuntimeError("snp_valid model failed rc=%d" % rc)
    blob = C.string_at(out, bounds[stream.n_contigs])
    lib().np1m_free(out)
    return [blob[bounds[i]:bounds[i + 1]].decode() for i in range(stream.n_contigs)]


def snp_phase(sr, lr, cfg):
    """snp_phase (task 3) through the stage bodies of np1_phase.h driven the way the kernels drive them; ValueError where the
    reference has no defined result (the product's ERR_SP_UNDEFINED)."""
    out = C.c_void_p()
    bounds = (C.c_uint32 * (sr.n_contigs + 1))()
    rc = lib().np1m_snp_phase(C.byref(sr.view), C.byref(lr.view), C.byref(cfg), C.byref(out), bounds)
    if rc > 0 and rc & 1024:
        raise ValueError("undefined upstream")
    if rc != 0:
        raise RuntimeError("snp_phase model failed rc=%d" % rc)
    blob = C.string_at(out, bounds[sr.n_contigs])
    lib().np1m_free(out)
    return [blob[bounds[i]:bounds[i + 1]].decode() for i in range(sr.n_contigs)]


ERR_KC_UNDEFINED = 512     # np1_kmer.h: an input the reference has no defined result for


def replay_revotes():
    """vote rounds the max_count_kmer break made necessary in the replay models since the library was loaded"""
    return int(lib().np1m_replay_revote_count())


def replay_breaks():
    """parts whose first loop left through the max_count_kmer break in the replay models since the library was loaded"""
    return int(lib().np1m_replay_break_count())


def snp_valid_replay(stream, cfg, bam):
    """task 4 the same way: both rounds of snp_valid on the replayed iterator (None: undefined upstream)"""
    return kmer_count_replay(stream, cfg, bam, fn="np1m_snp_valid_replay")


def kmer_count_replay(stream, cfg, bam, fn="np1m_kmer_count_replay"):
    """kmer_count with the region iterator of the reference replayed on the host (np1_replay.h: chunk lists of the BAM index, re-use,
    saved offsets, the max_count_kmer break) feeding kc_part_winner: `stream` must have been read from `bam` (it carries the records'
    virtual offsets)."""
    import oracle_binding as ob
    names = ob.bam_reference_names(bam)
    tid = (C.c_int32 * stream.n_contigs)(*[names.index(n) if n in names else -1 for n in stream.names])
    vb, ve = stream.voffs()
    out = C.c_void_p()
    bounds = (C.c_uint32 * (stream.n_contigs + 1))()
    rc = getattr(lib(), fn)(C.byref(stream.view), C.byref(cfg), (bam + ".bai").encode(), tid, vb.ctypes.data, ve.ctypes.data, C.byref(out), bounds)
    if rc == ERR_KC_UNDEFINED:
        return None
    if rc != 0:
        raise RuntimeError("%s failed rc=%d" % (fn, rc))
    blob = C.string_at(out, bounds[stream.n_contigs])
    lib().np1m_free(out)
    return [blob[bounds[i]:bounds[i + 1]].decode() for i in range(stream.n_contigs)]


def score_chain_tiled_files(fasta, bam, name, tile_bp, halo_bp, cfg=None, fused=1, first_tile=0, tile_stride=1):
    """the product's tiling driver (np1_tile.cpp) with the model in place of the device: tiles read from the files through the index
    (np_stream.cpp: load_stream_region); returns (string, dict(tiles, recomputed, records))"""
    cfg = cfg or nat.default_config()
    C.c_int.in_dll(lib(), "np1m_fused").value = int(fused)
    out, n, ts = C.c_void_p(), C.c_int64(0), (C.c_uint64 * 3)()
    f = lib().np1m_score_chain_tiled_files
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(nat.Configure), C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                  C.POINTER(C.c_uint64)]
    rc = f(fasta.encode(), bam.encode(), name.encode(), C.byref(cfg), tile_bp, halo_bp, first_tile, tile_stride, C.byref(out), C.byref(n), ts)
    if rc != 0:
        raise RuntimeError("tiled model (files) failed rc=%d" % rc)
    s = C.string_at(out, n.value).decode()
    lib().np1m_free(out)
    return s, dict(tiles=ts[0], recomputed=ts[1], records=ts[2])
