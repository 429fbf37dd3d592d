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
