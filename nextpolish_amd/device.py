"""Device-side objects of the batch interface (np1_ctx / np1_batch in include/nextpolish1.h)."""
import ctypes as C

from . import _native as nat


class Batch(object):
    """A decoded record stream resident in HBM."""

    def __init__(self, ctx, stream):
        self.ctx = ctx
        self.stream = stream
        self.handle = nat.lib().np1_batch_upload(ctx.handle, stream.handle)
        if not self.handle:
            raise RuntimeError("np1_batch_upload: " + nat.last_error())

    def score_chain(self, cfg=None, timed=False):
        cfg = cfg or nat.default_config()
        ms = (C.c_float * nat.NP1_MAX_STAGES)() if timed else None
        if nat.lib().np1_batch_score_chain(self.handle, C.byref(cfg), ms) != 0:
            raise RuntimeError("np1_batch_score_chain: " + nat.last_error())
        if timed:
            n = nat.lib().np1_stage_count()
            return {nat.lib().np1_stage_name(i).decode(): float(ms[i]) for i in range(n)}
        return None

    def kmer_count(self, cfg):
        """Task 2 over the resident batch (stream loaded with qualities; cfg.read_tlen set, e.g. by config_init)."""
        if nat.lib().np1_batch_kmer_count(self.handle, C.byref(cfg), None) != 0:
            raise RuntimeError("np1_batch_kmer_count: " + nat.last_error())

    def enable_replay(self, bam):
        """kmer_count of this batch replays the reference's region iterator (the stream must have been read from `bam`; experimental)"""
        if nat.lib().np1_batch_enable_replay(self.handle, self.stream.handle, bam.encode()) != 0:
            raise RuntimeError("np1_batch_enable_replay: " + nat.last_error())

    def snp_valid(self, cfg):
        """Task 4 over the resident batch (reference: source/lib/snpvalid.c; stream loaded with qualities)."""
        if nat.lib().np1_batch_snp_valid(self.handle, C.byref(cfg), None) != 0:
            raise RuntimeError("np1_batch_snp_valid: " + nat.last_error())

    def snp_phase(self, long_reads, cfg):
        """Task 3 (reference: source/lib/snpphase.c): this batch holds the short reads, `long_reads` is the Batch of the long reads
        of the same contigs (both streams loaded with qualities, same context).  The result lands in this batch."""
        if nat.lib().np1_batch_snp_phase(self.handle, long_reads.handle, C.byref(cfg)) != 0:
            raise RuntimeError("np1_batch_snp_phase: " + nat.last_error())

    def results(self):
        L = nat.lib()
        out = []
        for i in range(self.stream.n_contigs):
            n = L.np1_batch_result_len(self.handle, i)
            if n < 0:
                raise RuntimeError("no result for contig %d" % i)
            buf = C.create_string_buffer(n + 1)
            if L.np1_batch_result_copy(self.handle, i, buf, n + 1) != 0:
                raise RuntimeError("np1_batch_result_copy: " + nat.last_error())
            out.append(buf.value.decode())
        return out

    def update_count(self):
        return int(nat.lib().np1_batch_update_count(self.handle))

    def device_bytes(self):
        return int(nat.lib().np1_batch_device_bytes(self.handle))

    def close(self):
        if self.handle:
            nat.lib().np1_batch_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context(object):
    """One per process per GPU.  Creating it touches the GPU: do it after any fork."""

    def __init__(self, device=0):
        n = nat.lib().np1_device_count()
        if n <= 0:
            raise RuntimeError("no HIP device available; nextpolish_amd has no CPU fallback")
        self.handle = nat.lib().np1_ctx_create(device)
        if not self.handle:
            raise RuntimeError("np1_ctx_create: " + nat.last_error())

    def upload(self, stream):
        return Batch(self, stream)

    def score_chain(self, stream, cfg=None):
        b = Batch(self, stream)
        try:
            b.score_chain(cfg)
            return b.results()
        finally:
            b.close()

    def close(self):
        if self.handle:
            nat.lib().np1_ctx_destroy(self.handle)
            self.handle = None


class Pipe(object):
    """Streamed polishing on several device lanes (np1_pipe_* in include/nextpolish1.h): pinned host arrays -> H2D -> kernels
    -> D2H with the copies of one batch behind the kernels of another."""

    def __init__(self, device=0, lanes=2):
        if nat.lib().np1_device_count() <= 0:
            raise RuntimeError("no HIP device available; nextpolish_amd has no CPU fallback")
        self.handle = nat.lib().np1_pipe_open(device, lanes)
        if not self.handle:
            raise RuntimeError("np1_pipe_open: " + nat.last_error())

    def run(self, streams, cfg=None, task=1, fetch=True):
        """Polishes the host streams (one batch each); returns [[sequence per contig] per batch] (or None when fetch=False)."""
        cfg = cfg or nat.default_config()
        arr = (C.c_void_p * len(streams))(*[s.handle for s in streams])
        if nat.lib().np1_pipe_run(self.handle, arr, len(streams), C.byref(cfg), task) != 0:
            raise RuntimeError("np1_pipe_run: " + nat.last_error())
        if not fetch:
            return None
        out = []
        n = C.c_int64(0)
        for k, st in enumerate(streams):
            row = []
            for c in range(st.n_contigs):
                p = nat.lib().np1_pipe_result(self.handle, k, c, C.byref(n))
                row.append(C.string_at(p, n.value).decode())
            out.append(row)
        return out

    def result_lengths(self, streams):
        n = C.c_int64(0)
        tot = 0
        for k, st in enumerate(streams):
            for c in range(st.n_contigs):
                nat.lib().np1_pipe_result(self.handle, k, c, C.byref(n))
                tot += n.value
        return tot

    def run_files(self, fasta, bam, names=None, batch_bp=16000000, cfg=None, task=1, sink=None, raw_sink=None):
        """BAM + FASTA on disk -> sink(name, sequence) per contig in FASTA-index order (loaders, lanes and sink overlapped).
        raw_sink(name bytes, pointer, length) skips the copy into a Python string."""
        cfg = cfg or nat.default_config()
        got = []

        def _sink(_user, name, seq, length):
            if raw_sink is not None:
                raw_sink(name, seq, length)
                return
            s = C.string_at(seq, length).decode()
            (sink or (lambda a, b: got.append((a, b))))(name.decode(), s)

        cb = nat.SINK_FN(_sink)
        names = list(names or [])
        arr = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        if nat.lib().np1_pipe_run_files(self.handle, fasta.encode(), bam.encode(), arr if names else None, len(names), batch_bp,
                                        C.byref(cfg), task, cb, None) != 0:
            raise RuntimeError("np1_pipe_run_files: " + nat.last_error())
        return got

    def run_phase_files(self, fasta, bam_sr, bam_lr, names=None, batch_bp=16000000, cfg=None, sink=None):
        """Task 3 from files: FASTA + short-read BAM + long-read BAM -> sink(name, sequence) per contig in request order
        (np1_pipe_run_phase_files: device-side ingest of the short reads, host loader for the long reads, one snp_phase pass per batch)."""
        cfg = cfg or nat.default_config()
        got = []

        def _sink(_user, name, seq, length):
            (sink or (lambda a, b: got.append((a, b))))(name.decode(), C.string_at(seq, length).decode())

        cb = nat.SINK_FN(_sink)
        names = list(names or [])
        arr = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        if nat.lib().np1_pipe_run_phase_files(self.handle, fasta.encode(), bam_sr.encode(), bam_lr.encode(), arr if names else None, len(names), batch_bp,
                                              C.byref(cfg), cb, None) != 0:
            raise RuntimeError("np1_pipe_run_phase_files: " + nat.last_error())
        return got

    def close(self):
        if self.handle:
            nat.lib().np1_pipe_close(self.handle)
            self.handle = None
