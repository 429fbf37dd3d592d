"""ctypes binding of the in-tree shared library (include/nextpolish1.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C nextpolish_amd/csrc``
as ``nextpolish_amd/lib/nextpolish1.so`` -- the same file name the reference's caller loads
(reference: source/lib/nextpolish1.py:84).  Importing this module never touches a GPU; device
work starts at ``np1_ctx_create``.  A missing library is a hard error (no fallback path).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.realpath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "nextpolish1.so")


class Configure(C.Structure):
    """reference: source/lib/config.h:25-67, source/lib/nextpolish1.py:27-65"""
    _fields_ = [
        ("trim_len_edge", C.c_uint8), ("ext_len_edge", C.c_uint8), ("min_map_quality", C.c_uint8),
        ("indel_balance_factor_sgs", C.c_double), ("min_count_ratio_skip", C.c_double),
        ("min_len_ldr", C.c_uint8), ("min_len_inter_kmer", C.c_uint8), ("max_len_kmer", C.c_uint8),
        ("max_count_kmer", C.c_uint8),
        ("min_depth_snp", C.c_uint8), ("min_count_snp", C.c_uint8), ("min_count_snp_link", C.c_int8),
        ("ploidy", C.c_double), ("indel_balance_factor_lgs", C.c_double), ("max_indel_factor_lgs", C.c_double),
        ("max_snp_factor_lgs", C.c_double), ("min_snp_factor_sgs", C.c_double),
        ("region_count", C.c_int32), ("count_read_ins_sgs", C.c_uint32), ("max_ins_len_sgs", C.c_uint32),
        ("max_ins_fold_sgs", C.c_int32), ("max_variant_count_lgs", C.c_int32),
        ("max_clip_ratio_sgs", C.c_double), ("max_clip_ratio_lgs", C.c_double),
        ("trace_polish_open", C.c_int32), ("read_tlen", C.c_int32), ("read_len", C.c_int32),
        ("fastafn", C.c_char_p), ("bamfn", C.c_char_p), ("thirdbamfn", C.c_char_p),
    ]


class PolishPoint(C.Structure):
    _fields_ = [("pos", C.c_int32), ("index", C.c_int16), ("curbase", C.c_char), ("base", C.c_char)]


class PolishResult(C.Structure):
    _fields_ = [("contig", C.c_void_p), ("data", C.POINTER(PolishPoint)), ("length", C.c_int32),
                ("datalength", C.c_int32)]


class StreamView(C.Structure):
    _fields_ = [
        ("n_contigs", C.c_int64), ("n_reads", C.c_int64),
        ("ctg_len", C.c_void_p), ("ctg_off", C.c_void_p), ("read_begin", C.c_void_p),
        ("draft", C.c_void_p), ("draft_len", C.c_int64),
        ("pos", C.c_void_p), ("ctg", C.c_void_p), ("flag", C.c_void_p), ("n_cigar", C.c_void_p),
        ("l_qseq", C.c_void_p), ("cigar_off", C.c_void_p), ("seq_off", C.c_void_p),
        ("mapq", C.c_void_p), ("isize", C.c_void_p), ("qual_off", C.c_void_p),
        ("cigar", C.c_void_p), ("cigar_len", C.c_int64),
        ("seq", C.c_void_p), ("seq_len", C.c_int64),
        ("qual", C.c_void_p), ("qual_len", C.c_int64),
    ]


class SynthParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_contigs", C.c_int32), ("contig_len", C.POINTER(C.c_int32)),
        ("depth", C.c_double), ("read_len", C.c_int32), ("frag_mean", C.c_double), ("frag_sd", C.c_double),
        ("draft_sub", C.c_double), ("draft_indel", C.c_double), ("draft_lower", C.c_double),
        ("read_sub", C.c_double), ("read_indel", C.c_double), ("softclip_rate", C.c_double),
        ("dup_rate", C.c_double), ("supp_rate", C.c_double), ("sec_rate", C.c_double),
        ("unmapped_rate", C.c_double), ("lowmapq_rate", C.c_double), ("weird_rate", C.c_double),
        ("with_qual", C.c_int32),
    ]


NP1_MAX_STAGES = 16
_lib = None


class DiploidParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_contigs", C.c_int32), ("contig_len", C.POINTER(C.c_int32)), ("sr_depth", C.c_double), ("lr_depth", C.c_double),
                ("read_len", C.c_int32), ("frag_mean", C.c_double), ("lr_len", C.c_double), ("het_sub", C.c_double), ("het_indel", C.c_double),
                ("draft_err", C.c_double), ("sr_err", C.c_double), ("lr_err", C.c_double), ("sr_holes", C.c_int32)]


class SynthLongParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_contigs", C.c_int32), ("contig_len", C.POINTER(C.c_int32)), ("depth", C.c_double),
                ("mean_len", C.c_double), ("sub", C.c_double), ("ins", C.c_double), ("dele", C.c_double), ("max_indel", C.c_int32),
                ("clip_rate", C.c_double)]


SINK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64)   # (user, name, seq pointer, length)


def lib():
    """Loads nextpolish1.so once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no non-HIP fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.config_init.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.config_init.restype = C.POINTER(Configure)
    L.config_destory.argtypes = [C.POINTER(Configure)]
    L.config_destory.restype = None
    for fn in ("score_chain", "kmer_count", "snp_phase", "snp_valid", "lgspolish"):
        getattr(L, fn).argtypes = [C.c_char_p, C.POINTER(Configure)]
        getattr(L, fn).restype = C.POINTER(PolishResult)
    L.polishresult_destory.argtypes = [C.POINTER(PolishResult)]
    L.polishresult_destory.restype = None
    L.np1_last_error.restype = C.c_char_p
    L.np1_stream_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int]
    L.np1_stream_load.restype = C.c_void_p
    L.np1_stream_build.argtypes = [C.POINTER(StreamView), C.POINTER(C.c_char_p)]
    L.np1_stream_build.restype = C.c_void_p
    L.np1_stream_get_view.argtypes = [C.c_void_p, C.POINTER(StreamView)]
    L.np1_stream_get_view.restype = None
    L.np1_stream_contig_name.argtypes = [C.c_void_p, C.c_int64]
    L.np1_stream_contig_name.restype = C.c_char_p
    L.np1_stream_algorithmic_bytes.argtypes = [C.c_void_p, C.c_int]
    L.np1_stream_algorithmic_bytes.restype = C.c_uint64
    L.np1_stream_write_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
    L.np1_stream_write_files.restype = C.c_int
    L.np1_stream_free.argtypes = [C.c_void_p]
    L.np1_stream_free.restype = None
    L.np1_synth_defaults.argtypes = [C.POINTER(SynthParams)]
    L.np1_synth_defaults.restype = None
    L.np1_stream_synth.argtypes = [C.POINTER(SynthParams), C.c_char_p]
    L.np1_stream_synth.restype = C.c_void_p
    L.np1_stream_synth_long.argtypes = [C.POINTER(SynthLongParams), C.c_char_p]
    L.np1_stream_synth_long.restype = C.c_void_p
    L.np1_stream_synth_diploid.argtypes = [C.POINTER(DiploidParams), C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.np1_stream_synth_diploid.restype = C.c_int
    L.np1_stream_voffs.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint64))]
    L.np1_stream_voffs.restype = C.c_int64
    L.np1_device_count.restype = C.c_int
    L.np1_ctx_create.argtypes = [C.c_int]
    L.np1_ctx_create.restype = C.c_void_p
    L.np1_ctx_destroy.argtypes = [C.c_void_p]
    L.np1_ctx_destroy.restype = None
    L.np1_batch_upload.argtypes = [C.c_void_p, C.c_void_p]
    L.np1_batch_upload.restype = C.c_void_p
    L.np1_batch_free.argtypes = [C.c_void_p]
    L.np1_batch_free.restype = None
    L.np1_stage_count.restype = C.c_int
    L.np1_stage_name.argtypes = [C.c_int]
    L.np1_stage_name.restype = C.c_char_p
    L.np1_batch_score_chain.argtypes = [C.c_void_p, C.POINTER(Configure), C.POINTER(C.c_float)]
    L.np1_batch_score_chain.restype = C.c_int
    L.np1_batch_kmer_count.argtypes = [C.c_void_p, C.POINTER(Configure), C.POINTER(C.c_float)]
    L.np1_batch_kmer_count.restype = C.c_int
    L.np1_batch_snp_valid.argtypes = [C.c_void_p, C.POINTER(Configure), C.POINTER(C.c_float)]
    L.np1_batch_snp_valid.restype = C.c_int
    L.np1_batch_snp_phase.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Configure)]
    L.np1_batch_snp_phase.restype = C.c_int
    L.np1_batch_enable_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    L.np1_batch_enable_replay.restype = C.c_int
    L.np1_batch_sync.argtypes = [C.c_void_p]
    L.np1_batch_sync.restype = C.c_int
    L.np1_batch_result_len.argtypes = [C.c_void_p, C.c_int64]
    L.np1_batch_result_len.restype = C.c_int64
    L.np1_batch_result_copy.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int64]
    L.np1_batch_result_copy.restype = C.c_int
    L.np1_batch_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
    L.np1_batch_debug_counters.restype = C.c_int
    L.np1_batch_debug_slots.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.np1_batch_debug_slots.restype = C.c_int64
    L.np1_batch_update_count.argtypes = [C.c_void_p]
    L.np1_batch_update_count.restype = C.c_int64
    L.np1_batch_device_bytes.argtypes = [C.c_void_p]
    L.np1_batch_device_bytes.restype = C.c_int64
    L.np1_alloc_stats.argtypes = [C.POINTER(C.c_uint64)]
    L.np1_alloc_stats.restype = None
    L.np1_alloc_trim.argtypes = []
    L.np1_alloc_trim.restype = None
    L.np1_diag_report.argtypes = [C.c_int]
    L.np1_diag_report.restype = None
    L.calgs.argtypes = [C.c_char_p]
    L.calgs.restype = C.c_uint64
    L.np1_stream_pin.argtypes = [C.c_void_p]
    L.np1_stream_pin.restype = C.c_int
    L.np1_batch_create.argtypes = [C.c_void_p]
    L.np1_batch_create.restype = C.c_void_p
    L.np1_batch_reload.argtypes = [C.c_void_p, C.c_void_p]
    L.np1_batch_reload.restype = C.c_int
    L.np1_batch_results_fetch.argtypes = [C.c_void_p]
    L.np1_batch_results_fetch.restype = C.c_int
    L.np1_pipe_open.argtypes = [C.c_int, C.c_int]
    L.np1_pipe_open.restype = C.c_void_p
    L.np1_pipe_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(Configure), C.c_int]
    L.np1_pipe_run.restype = C.c_int
    L.np1_pipe_result.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_int64)]
    L.np1_pipe_result.restype = C.c_void_p
    L.np1_pipe_close.argtypes = [C.c_void_p]
    L.np1_pipe_close.restype = None
    L.np1_pipe_run_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int64,
                                     C.POINTER(Configure), C.c_int, SINK_FN, C.c_void_p]
    L.np1_pipe_run_files.restype = C.c_int
    L.np1_pipe_run_phase_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int64,
                                           C.POINTER(Configure), SINK_FN, C.c_void_p]
    L.np1_pipe_run_phase_files.restype = C.c_int
    _lib = L
    return L


def last_error():
    return (lib().np1_last_error() or b"").decode()


def default_config():
    """A Configure carrying config_init's defaults (reference: source/lib/config.c:11-38), no files."""
    cfg = Configure()
    cfg.trim_len_edge, cfg.ext_len_edge, cfg.min_map_quality = 2, 2, 0
    cfg.indel_balance_factor_sgs, cfg.min_count_ratio_skip = 0.5, 0.8
    cfg.min_len_ldr, cfg.min_len_inter_kmer, cfg.max_len_kmer, cfg.max_count_kmer = 3, 5, 50, 50
    cfg.min_depth_snp, cfg.min_count_snp, cfg.min_count_snp_link = 3, 5, 5
    cfg.ploidy, cfg.indel_balance_factor_lgs, cfg.max_indel_factor_lgs = 2, 0.33, 0.21
    cfg.max_snp_factor_lgs, cfg.min_snp_factor_sgs = 0.53, 0.34
    cfg.region_count, cfg.count_read_ins_sgs, cfg.max_ins_len_sgs = 10000, 10000, 10000
    cfg.max_ins_fold_sgs, cfg.max_variant_count_lgs = 5, 150000
    cfg.max_clip_ratio_sgs, cfg.max_clip_ratio_lgs = 0.15, 0.4
    return cfg


def _arr(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=int(n))


class Stream(object):
    """Owning wrapper of a host decoded record stream (np1_stream)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("stream creation failed: " + last_error())
        self.handle = handle
        v = StreamView()
        lib().np1_stream_get_view(handle, C.byref(v))
        self.view = v
        nc, nr = v.n_contigs, v.n_reads
        self.n_contigs, self.n_reads = int(nc), int(nr)
        self.ctg_len = _arr(v.ctg_len, nc, np.int32)
        self.ctg_off = _arr(v.ctg_off, nc + 1, np.uint32)
        self.read_begin = _arr(v.read_begin, nc + 1, np.uint64)
        self.draft = _arr(v.draft, v.draft_len, np.uint8)
        self.pos = _arr(v.pos, nr, np.int32)
        self.ctg = _arr(v.ctg, nr, np.uint32)
        self.flag = _arr(v.flag, nr, np.uint16)
        self.n_cigar = _arr(v.n_cigar, nr, np.uint32)
        self.l_qseq = _arr(v.l_qseq, nr, np.int32)
        self.cigar_off = _arr(v.cigar_off, nr, np.uint64)
        self.seq_off = _arr(v.seq_off, nr, np.uint64)
        self.mapq = _arr(v.mapq, nr, np.uint8)
        self.isize = _arr(v.isize, nr, np.int32)
        self.qual_off = _arr(v.qual_off, nr, np.uint64)
        self.cigar = _arr(v.cigar, v.cigar_len, np.uint32)
        self.seq = _arr(v.seq, v.seq_len, np.uint8)
        self.qual = _arr(v.qual, v.qual_len, np.uint8)
        self.names = [lib().np1_stream_contig_name(handle, i).decode() for i in range(self.n_contigs)]

    @classmethod
    def load(cls, fasta, bam, names=None, with_qual=False):
        names = list(names or [])
        arr = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        return cls(lib().np1_stream_load(fasta.encode(), bam.encode(), arr, len(names), 1 if with_qual else 0))

    @classmethod
    def synth(cls, contig_len, depth=30.0, seed=20250117, prefix="ctg", **kw):
        p = SynthParams()
        lib().np1_synth_defaults(C.byref(p))
        lens = (C.c_int32 * len(contig_len))(*contig_len)
        p.seed, p.n_contigs, p.contig_len, p.depth = seed, len(contig_len), lens, depth
        for k, val in kw.items():
            if not hasattr(p, k):
                raise TypeError("unknown synth parameter " + k)
            setattr(p, k, val)
        return cls(lib().np1_stream_synth(C.byref(p), prefix.encode()))

    @classmethod
    def synth_long(cls, contig_len, depth=20.0, mean_len=8000.0, sub=0.03, ins=0.02, dele=0.02, max_indel=4, clip_rate=0.2,
                   seed=20250117, prefix="ctg"):
        """Long-read workload of the nextpolish2 path (random drafts, noisy reads with known CIGARs)."""
        p = SynthLongParams()
        lens = (C.c_int32 * len(contig_len))(*contig_len)
        p.seed, p.n_contigs, p.contig_len, p.depth, p.mean_len = seed, len(contig_len), lens, depth, mean_len
        p.sub, p.ins, p.dele, p.max_indel, p.clip_rate = sub, ins, dele, max_indel, clip_rate
        return cls(lib().np1_stream_synth_long(C.byref(p), prefix.encode()))

    @classmethod
    def synth_on(cls, contigs, long_reads=False, depth=None, seed=20250117, **kw):
        """The synthetic short-read (or long-read) workload over contigs that are handed in -- [(name, sequence)], e.g. the FASTA a
        polishing step wrote: the "re-mapped" reads of the next step of a multi-step run, aligned by construction (np_synth.h)."""
        L = lib()
        L.np1_stream_synth_on.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int]
        L.np1_stream_synth_on.restype = C.c_void_p
        if long_reads:
            p = SynthLongParams()
            p.seed, p.depth, p.mean_len = seed, 20.0 if depth is None else depth, 8000.0
            p.sub, p.ins, p.dele, p.max_indel, p.clip_rate = 0.03, 0.02, 0.02, 4, 0.2
        else:
            p = SynthParams()
            L.np1_synth_defaults(C.byref(p))
            p.seed, p.depth = seed, 30.0 if depth is None else depth
        for k, val in kw.items():
            if not hasattr(p, k):
                raise TypeError("unknown synth parameter " + k)
            setattr(p, k, val)
        n = len(contigs)
        seqs = [s.encode() if isinstance(s, str) else bytes(s) for _, s in contigs]
        names = (C.c_char_p * n)(*[nm.encode() for nm, _ in contigs])
        sq = (C.c_char_p * n)(*seqs)
        lens = (C.c_int64 * n)(*[len(x) for x in seqs])
        h = L.np1_stream_synth_on(C.byref(p), 1 if long_reads else 0, names, sq, lens, n)
        if not h:
            raise RuntimeError("np1_stream_synth_on: " + last_error())
        return cls(h)

    @classmethod
    def synth_diploid(cls, contig_len, sr_depth=30.0, lr_depth=20.0, seed=7, read_len=150, frag_mean=400.0, lr_len=8000.0, het_sub=0.002, het_indel=0.0003,
                      draft_err=0.002, sr_err=0.003, lr_err=0.04, sr_holes=0, prefix="ctg"):
        """(short-read stream, long-read stream) of the same diploid contigs: the workload of task 3 (np1_diploid_params)"""
        lens = (C.c_int32 * len(contig_len))(*contig_len)
        p = DiploidParams(seed, len(contig_len), lens, sr_depth, lr_depth, read_len, frag_mean, lr_len, het_sub, het_indel, draft_err, sr_err, lr_err, sr_holes)
        a, b = C.c_void_p(), C.c_void_p()
        if lib().np1_stream_synth_diploid(C.byref(p), prefix.encode(), C.byref(a), C.byref(b)) != 0:
            raise RuntimeError("np1_stream_synth_diploid: " + last_error())
        return cls(a.value), cls(b.value)

    @classmethod
    def from_reads(cls, contigs, reads):
        """contigs: [(name, draft_str)]; reads: [dict(ctg=int, pos=int, flag=int, mapq=int, isize=int,
        cigar=[(op_char, len)], seq=str, qual=bytes|None)] already sorted by (ctg, pos).  Test helper."""
        OPS = "MIDNSHP=X"
        NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
        nc, nr = len(contigs), len(reads)
        draft = "".join(d for _, d in contigs).encode()
        ctg_len = np.array([len(d) for _, d in contigs], dtype=np.int32)
        ctg_off = np.zeros(nc + 1, dtype=np.uint32)
        ctg_off[1:] = np.cumsum(ctg_len)
        read_begin = np.zeros(nc + 1, dtype=np.uint64)
        for r in reads:
            read_begin[r["ctg"] + 1:] += 1
        cig, seq, qual = [], bytearray(), bytearray()
        cigar_off, seq_off, qual_off = [], [], []
        for r in reads:
            cigar_off.append(len(cig)); seq_off.append(len(seq)); qual_off.append(len(qual))
            cig += [(n << 4) | OPS.index(o) for o, n in r["cigar"]]
            s = r["seq"]
            for k in range(0, len(s), 2):
                seq.append(NT16.get(s[k], 15) << 4 | (NT16.get(s[k + 1], 15) if k + 1 < len(s) else 0))
            q = r.get("qual")
            qual += bytes(q) if q is not None else bytes([30] * len(s))
        arrs = dict(
            ctg_len=ctg_len, ctg_off=ctg_off, read_begin=read_begin, draft=np.frombuffer(draft, dtype=np.uint8),
            pos=np.array([r["pos"] for r in reads], dtype=np.int32), ctg=np.array([r["ctg"] for r in reads], dtype=np.uint32),
            flag=np.array([r.get("flag", 0) for r in reads], dtype=np.uint16),
            n_cigar=np.array([len(r["cigar"]) for r in reads], dtype=np.uint32),
            l_qseq=np.array([len(r["seq"]) for r in reads], dtype=np.int32),
            cigar_off=np.array(cigar_off, dtype=np.uint64), seq_off=np.array(seq_off, dtype=np.uint64),
            mapq=np.array([r.get("mapq", 60) for r in reads], dtype=np.uint8),
            isize=np.array([r.get("isize", 0) for r in reads], dtype=np.int32),
            qual_off=np.array(qual_off, dtype=np.uint64), cigar=np.array(cig, dtype=np.uint32),
            seq=np.frombuffer(bytes(seq), dtype=np.uint8), qual=np.frombuffer(bytes(qual), dtype=np.uint8))
        v = StreamView()
        v.n_contigs, v.n_reads = nc, nr
        for k, a in arrs.items():
            setattr(v, k, a.ctypes.data if a.size else 0)
        v.draft_len, v.cigar_len, v.seq_len, v.qual_len = len(draft), len(cig), len(seq), len(qual)
        names = (C.c_char_p * max(1, nc))(*[n.encode() for n, _ in contigs])
        return cls(lib().np1_stream_build(C.byref(v), names))

    def voffs(self):
        """(start, end) BGZF virtual offsets per record of a stream loaded from a file, as numpy arrays; None for in-memory streams"""
        if self.n_reads == 0:
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64)
        b, e = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        n = lib().np1_stream_voffs(self.handle, C.byref(b), C.byref(e))
        if n <= 0:
            return None
        return np.ctypeslib.as_array(b, shape=(n,)), np.ctypeslib.as_array(e, shape=(n,))

    def pin(self):
        """Page-locks the stream's arrays (asynchronous full-rate H2D copies; needs a HIP device)."""
        if lib().np1_stream_pin(self.handle) != 0:
            raise RuntimeError("np1_stream_pin: " + last_error())

    def algorithmic_bytes(self, with_qual=False):
        return int(lib().np1_stream_algorithmic_bytes(self.handle, 1 if with_qual else 0))

    def write_files(self, fasta, bam, level=1, aux=None):
        """aux: optional list of raw BAM optional-field bytes per record (e.g. b"SAZ" + value + b"\0")."""
        if aux is not None:
            L = lib()
            L.np1_stream_write_files_aux.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_uint64)]
            pool = b"".join(aux)
            off = (C.c_uint64 * (len(aux) + 1))()
            acc = 0
            for i, a in enumerate(aux):
                off[i] = acc
                acc += len(a)
            off[len(aux)] = acc
            if L.np1_stream_write_files_aux(self.handle, fasta.encode(), bam.encode(), level, pool + b"\0", off) != 0:
                raise RuntimeError("write_files: " + last_error())
            return
        if lib().np1_stream_write_files(self.handle, fasta.encode(), bam.encode(), level) != 0:
            raise RuntimeError("write_files: " + last_error())

    def contig_draft(self, i):
        return self.draft[int(self.ctg_off[i]):int(self.ctg_off[i + 1])].tobytes()

    def close(self):
        if self.handle:
            lib().np1_stream_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
