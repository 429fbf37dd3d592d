// Device context, HBM-resident batches and the score_chain launch sequence
// (include/nextpolish1.h, Part 2: np1_ctx_*, np1_batch_*).  Kernels: np1_kernels.hip.
#include <hip/hip_runtime.h>

#include <sys/stat.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <array>
#include <mutex>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np1_kernels.h"
#include "np1_kmer_kernels.h"
#include "np_stream.h"
#include "np1_priv.h"
#include "np1_batch_priv.h"
#include "np1_replay.h"
#include "np_threads.h"
#include "np1_upload.h"


using namespace np1k;

using namespace np1dev;

namespace {

// Two launch sequences share everything but the middle: the default FUSED one stages pileup columns through
// LDS (k_place + k_tile); the STAGED one (NP1_PIPELINE=staged, kept for A/B measurements and mirrored by the
// host model in tests/model) materialises symbol rows in HBM (k_rowcap + scan + k_rows + k_vote).
const char* kStageNamesStaged[] = {"prep", "scan_slots", "slotinfo", "rowcap_scan", "rows", "vote", "dp", "emit"};
const char* kStageNamesFused[] = {"prep", "scan_slots", "slotinfo", "desc", "-", "tile", "dp", "emit"};
bool use_staged() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("NP1_PIPELINE");
        v = (e && strcmp(e, "staged") == 0) ? 1 : 0;
    }
    return v == 1;
}

}  // namespace

void np1_ctx_retain(np1_ctx* c) { c->refs.fetch_add(1); }
void np1_ctx_release(np1_ctx* c) {
    if (c->refs.fetch_sub(1) != 1) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < kStages; ++i) { (void)hipEventDestroy(c->ev0[i]); (void)hipEventDestroy(c->ev1[i]); }
    (void)npalloc::stream_destroy(c->stream);
    delete c;
}

extern "C" {

int np1_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

np1_ctx* np1_ctx_create(int device) {
    int n = np1_device_count();
    if (n <= 0) { np1_set_error("no HIP device available (this library has no CPU path)"); return nullptr; }
    if (device < 0 || device >= n) { np1_set_error("invalid HIP device index"); return nullptr; }
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return nullptr;
    np1_ctx* c = new np1_ctx();
    c->device = device;
    if (!hip_ok(npalloc::stream_create(&c->stream), "hipStreamCreate")) { delete c; return nullptr; }
    for (int i = 0; i < kStages; ++i) {
        (void)hipEventCreate(&c->ev0[i]);
        (void)hipEventCreate(&c->ev1[i]);
    }
    return c;
}

void np1_ctx_destroy(np1_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    np1_ctx_release(c);      // batches that are still alive keep the stream and the events until they are freed
}

// diagnostics: streams of this library with their state and the allocator caches' counters, written to `fd` (np_devalloc.h: report)
void np1_diag_report(int fd) { npalloc::report(fd, "nextpolish1.so"); }
// cache counters for tests and bench.py: {hits, misses, runtime frees, fenced, idle bytes, live bytes, peak idle bytes, bound} of the device cache
void np1_alloc_stats(uint64_t out[8]) {
    const npalloc::CacheStats st = npalloc::dev_cache().stats();
    out[0] = st.hits; out[1] = st.misses; out[2] = st.raw_frees; out[3] = st.fenced; out[4] = st.cached; out[5] = st.live; out[6] = st.peak_cached;
    out[7] = npalloc::dev_cache().cap();
}
// every idle block of both caches back to the runtime (a worker that shares its GPU and is about to sit idle)
void np1_alloc_trim(void) { npalloc::dev_cache().flush(-1); npalloc::host_cache().flush(-1); }

int np1_stage_count(void) { return kStages; }
const char* np1_stage_name(int i) {
    if (i < 0 || i >= kStages) return "";
    return use_staged() ? kStageNamesStaged[i] : kStageNamesFused[i];
}

// Every uploaded array carries 64 bytes of pad: the kernels stage pools with 16-byte loads from 16-byte aligned addresses, which reach
// up to 15 bytes past the last element (the pad is part of the allocation, not slack the allocator happens to leave: NP_EFENCE=1)
static int upload(DevBuf& b, const void* src, size_t bytes, hipStream_t st) {
    if (b.ensure(bytes + 64) != 0) return -1;
    if (bytes) HIPCHK(npcopy::h2d(b.p, src, bytes, st));
    return 0;
}

// (Re)fills a batch object from a host stream: the HBM buffers only grow, so a batch object that is reused for a sequence of
// similar batches allocates nothing in steady state.  sync = false leaves the copies in flight on the context's stream (the
// kernels that follow are ordered behind them); the host arrays must then stay alive and -- for the copies to overlap with
// other lanes' work -- pinned (np1_stream_pin).
// Longest record, and whether cigar_off / seq_off / ctg are exactly what a device rebuilds from n_cigar, l_qseq and read_begin (true
// for every stream this library makes: loaders and generators append to the pools in record order).  Found once per stream.
static void stream_facts(np1_stream* st) {
    std::lock_guard<std::mutex> hold(st->facts_mu);      // (two lanes reloading one stream, pin + reload: the const API invites it)
    if (st->facts) return;
    const np::ReadStream& s = st->s;
    const size_t n = s.n_reads();
    uint32_t mx = 0, mx_ops = 0;
    bool dense = s.read_begin.size() == s.n_contigs() + 1 && (n == 0 || (s.cigar_off[0] == 0 && s.seq_off[0] == 0));
    uint64_t c_at = 0, s_at = 0;
    size_t ct = 0;
    for (size_t i = 0; i < n; ++i) {
        const int32_t l = s.l_qseq[i];
        if (l > 0 && (uint32_t)l > mx) mx = (uint32_t)l;
        if (s.n_cigar[i] > mx_ops) mx_ops = s.n_cigar[i];
        if (dense) {
            while (ct + 1 < s.read_begin.size() && s.read_begin[ct + 1] <= i) ++ct;
            dense = s.cigar_off[i] == c_at && s.seq_off[i] == s_at && s.ctg[i] == ct && l >= 0;
            c_at += s.n_cigar[i];
            s_at += ((uint64_t)(uint32_t)l + 1) >> 1;
        }
    }
    st->max_lq = mx;
    st->ncig16.clear();
    if (mx_ops <= 0xffffu && n > 0) {
        st->ncig16.resize(n);
        for (size_t i = 0; i < n; ++i) st->ncig16[i] = (uint16_t)s.n_cigar[i];
    }
    // ---- the upload forms (np1_upload.h)
    static const bool full = getenv("NP1_UPLOAD") && strcmp(getenv("NP1_UPLOAD"), "full") == 0;
    const size_t nb = s.seq.size();
    st->seq2.clear(); st->esc_at.clear(); st->esc_val.clear();
    if (!full && nb >= 64) np1up::build_seq2(s.seq, &st->seq2, &st->esc_at, &st->esc_val);                    // 2 bits per base
    st->draft4.clear(); st->desc_at.clear(); st->desc_val.clear();
    if (!full && s.draft.size() >= 4096) np1up::build_draft4(s.draft, &st->draft4, &st->desc_at, &st->desc_val);   // 4 bits per draft character
    st->compact = np1_stream::Compact();
    static const bool no_compact = getenv("NP1_UPLOAD") && strcmp(getenv("NP1_UPLOAD"), "seq2") == 0;      // NP1_UPLOAD=seq2: 2-bit bases only
    // (small batches: the scans and kernels that undo the compact record form cost more than the bytes it saves -- measured on 13 Mb
    // batches, 2.6 M records: 3650 -> 3300 Mbp/s; NP1_COMPACT_MIN = records from which on it is used)
    static const size_t compact_min = getenv("NP1_COMPACT_MIN") ? (size_t)atoll(getenv("NP1_COMPACT_MIN")) : ((size_t)1 << 22);
    if (!full && !no_compact && dense && n >= 64 && n >= compact_min) {
        np1up::build_compact(s, &st->compact);
        const uint64_t as_is = 4 * n + (st->ncig16.empty() ? 4 * n : 2 * n) + 4 * n + 4 * s.cigar.size();
        st->compact.on = np1up::compact_bytes(st->compact, n) * 10 < as_is * 9;
        if (!st->compact.on) st->compact = np1_stream::Compact();
    }
    {
        const bool slim_ok = !full && dense && n > 0;
        const np1_stream::Compact& C = st->compact;
        uint64_t ub = (st->draft4.empty() ? s.draft.size() : st->draft4.size() + 9 * st->desc_at.size()) + 4 * s.ctg_off.size() + 2 * n + 8 * s.read_begin.size();
        if (C.on) ub += 4 * C.plain.size() + n + 4 * C.x_pos.size() + 8 * C.x_lq.size() + 4 * C.x_cigar.size();
        else ub += 4 * n + 4 * n + 4 * s.cigar.size() + (st->ncig16.empty() ? 4 * n : 2 * n);
        ub += st->seq2.empty() ? nb : st->seq2.size() + 9 * st->esc_at.size();
        if (!slim_ok) ub += 20 * n;
        if (!s.qual.empty()) ub += 13 * n + s.qual.size();      // mapq, isize, quality offsets, qualities (tasks 2-4)
        st->upload_bytes = ub;
    }
    st->facts = dense ? 1 : 2;
}

static int fill_batch(np1_batch* b, const np1_stream* st_, bool sync) {
    np1_ctx* ctx = b->ctx;
    np1_stream* st = const_cast<np1_stream*>(st_);      // (the cached facts are not part of the stream's value)
    stream_facts(st);
    const np::ReadStream& s = st->s;
    if (s.draft.size() >= 0xfff00000ull) { np1_set_error("batch too large: draft must stay below 2^32 slots"); return -1; }
    (void)hipSetDevice(ctx->device);
    b->nc = (uint32_t)s.n_contigs();
    b->G = s.draft.size();
    b->n_reads = (int64_t)s.n_reads();
    b->h_ctg_off = s.ctg_off;
    b->max_lq = 0;
    b->replay.on = false;   // (its record view belongs to the stream of the previous fill)
    b->force_staged = false;
    b->ran = false;
    b->out_cached = false;
    b->out_pinned = false;
    b->max_lq = st->max_lq;
    hipStream_t q = ctx->stream;
    auto up = [&](DevBuf& d, const void* src, size_t bytes, hipStream_t qq) { return upload(d, st->up(src), bytes, qq); };
    size_t n = s.n_reads();
    int rc = 0;
    static const bool slim = !(getenv("NP1_UPLOAD") && strcmp(getenv("NP1_UPLOAD"), "full") == 0);   // NP1_UPLOAD=full: every array as the host holds it
    const bool rebuild = slim && st->facts == 1 && n > 0;
    if (!st->draft4.empty()) {
        const size_t ne = st->desc_at.size();
        rc |= up(b->draft4, st->draft4.data(), st->draft4.size(), q);
        if (ne) { rc |= up(b->desc_at, st->desc_at.data(), 8 * ne, q); rc |= up(b->desc_val, st->desc_val.data(), ne, q); }
        if (b->draft.ensure(s.draft.size() + 16)) return -1;
        if (rc == 0) launch_unpack_draft4(q, b->draft4.as<uint8_t>(), (uint64_t)s.draft.size(), b->draft.as<uint8_t>(), b->desc_at.as<uint64_t>(), b->desc_val.as<uint8_t>(), (uint64_t)ne);
    } else {
        rc |= up(b->draft, s.draft.data(), s.draft.size(), q);
    }
    rc |= up(b->ctg_off, s.ctg_off.data(), 4 * s.ctg_off.size(), q);
    const np1_stream::Compact& C = st->compact;
    const bool compact = rebuild && C.on;
    rc |= up(b->flag, s.flag.data(), 2 * n, q);
    if (compact) {   // one bit + one byte per record, and in full only what is not a plain read a few bases behind the last one
        rc |= up(b->up_plain, C.plain.data(), 4 * C.plain.size(), q);
        rc |= up(b->up_dpos, C.dpos.data(), n, q);
        rc |= up(b->up_xpos, C.x_pos.data(), 4 * C.x_pos.size(), q);
        rc |= up(b->up_xlq, C.x_lq.data(), 4 * C.x_lq.size(), q);
        rc |= up(b->up_xncig, C.x_ncig.data(), 4 * C.x_ncig.size(), q);
        rc |= up(b->up_xcigar, C.x_cigar.data(), 4 * C.x_cigar.size(), q);
        if (b->pos.ensure(4 * n) || b->ncig.ensure(4 * n) || b->lq.ensure(4 * n) || b->cigar.ensure(4 * (size_t)C.n_ops + 16) ||
            b->up_work.ensure(8 * (3 * (n + 1) + C.x_pos.size() + 8)))
            return -1;
    } else if (!st->ncig16.empty()) {
        rc |= up(b->pos, s.pos.data(), 4 * n, q);
        rc |= up(b->lq, s.l_qseq.data(), 4 * n, q);
        rc |= up(b->cigar, s.cigar.data(), 4 * s.cigar.size(), q);   // the usual case: 2 bytes per record cross PCIe, the device widens them
        rc |= up(b->ncig16, st->ncig16.data(), 2 * n, q);
        if (b->ncig.ensure(4 * n)) return -1;
        if (rc == 0) launch_widen_u16(q, b->ncig16.as<uint16_t>(), b->ncig.as<uint32_t>(), (uint64_t)n);
    } else {
        rc |= up(b->pos, s.pos.data(), 4 * n, q);
        rc |= up(b->lq, s.l_qseq.data(), 4 * n, q);
        rc |= up(b->cigar, s.cigar.data(), 4 * s.cigar.size(), q);
        rc |= up(b->ncig, s.n_cigar.data(), 4 * n, q);
    }
    if (!st->seq2.empty()) {   // 2 bits per base over PCIe; the device expands them and patches the exception bytes in
        const size_t n2 = st->seq2.size(), ne = st->esc_at.size();
        rc |= up(b->seq2, st->seq2.data(), n2, q);
        if (ne) { rc |= up(b->esc_at, st->esc_at.data(), 8 * ne, q); rc |= up(b->esc_val, st->esc_val.data(), ne, q); }
        if (b->seq.ensure(2 * n2 + 64)) return -1;
        if (rc == 0) launch_unpack_seq2(q, b->seq2.as<uint8_t>(), (uint64_t)n2, b->seq.as<uint8_t>(), b->esc_at.as<uint64_t>(), b->esc_val.as<uint8_t>(), (uint64_t)ne);
    } else {
        rc |= up(b->seq, s.seq.data(), s.seq.size(), q);
    }
    b->h_read_begin = s.read_begin;
    rc |= up(b->read_begin, s.read_begin.data(), 8 * s.read_begin.size(), q);
    if (rebuild) {
        // 20 of the 32 fixed bytes per record are functions of the rest: pool offsets = running sums, contig = the record's place in
        // read_begin.  The device rebuilds them (two scans and a search per record) instead of taking them over PCIe.
        if (b->cigoff.ensure(8 * (n + 2)) || b->seqoff.ensure(8 * (n + 2)) || b->ctg.ensure(4 * n) ||
            b->scan_tmp.ensure(8 * (scan_tmp_words((uint64_t)n + 1) + 8)) || b->totals.ensure(256))
            return -1;
        const CompactDev cd{b->up_plain.as<uint32_t>(), b->up_xlq.as<int32_t>(), b->up_xncig.as<uint32_t>(), b->up_xcigar.as<uint32_t>(), b->up_dpos.as<uint8_t>(),
                            b->up_xpos.as<int32_t>(), C.common_lq, (uint64_t)n, (uint64_t)C.x_lq.size(), (uint64_t)C.x_pos.size()};
        if (rc == 0 && compact)
            launch_expand_records(q, cd, b->pos.as<int32_t>(), b->ncig.as<uint32_t>(), b->lq.as<int32_t>(), b->up_work.as<uint64_t>(), b->scan_tmp.as<uint64_t>(),
                                  b->totals.as<uint64_t>() + 24);
        if (rc == 0)
            launch_record_offsets(q, b->ncig.as<uint32_t>(), b->lq.as<int32_t>(), b->read_begin.as<uint64_t>(), b->nc, (uint64_t)n, b->cigoff.as<uint64_t>(),
                                  b->seqoff.as<uint64_t>(), b->ctg.as<uint32_t>(), b->scan_tmp.as<uint64_t>(), b->totals.as<uint64_t>() + 24);
        if (rc == 0 && compact)
            launch_expand_cigars(q, cd, b->ncig.as<uint32_t>(), b->cigoff.as<uint64_t>(), b->cigar.as<uint32_t>(), b->up_work.as<uint64_t>(), b->scan_tmp.as<uint64_t>(),
                                 b->totals.as<uint64_t>() + 24);
    } else {
        rc |= up(b->ctg, s.ctg.data(), 4 * n, q);
        rc |= up(b->cigoff, s.cigar_off.data(), 8 * n, q);
        rc |= up(b->seqoff, s.seq_off.data(), 8 * n, q);
    }
    b->has_qual = !s.qual.empty() || n == 0;
    if (b->has_qual) {   // kmer_count needs mapq / isize / base qualities too (kmercount.c:365-465, contig.c:648-665)
        rc |= up(b->mapq, s.mapq.data(), n, q);
        rc |= up(b->isize, s.isize.data(), 4 * n, q);
        rc |= up(b->qualoff, s.qual_off.data(), 8 * n, q);
        rc |= up(b->qual, s.qual.data(), s.qual.size(), q);
    }
    if (rc == 0 && sync && hipStreamSynchronize(q) != hipSuccess) { np1_set_error("upload failed"); rc = -1; }
    b->input_bytes = s.draft.size() + 32 * n + 4 * s.cigar.size() + s.seq.size();
    return rc;
}

np1_batch* np1_batch_upload(np1_ctx* ctx, const np1_stream* st) {
    if (!ctx || !st) { np1_set_error("np1_batch_upload: null argument"); return nullptr; }
    np1_batch* b = new np1_batch();
    b->ctx = ctx;
    np1_ctx_retain(ctx);
    if (fill_batch(b, st, true) != 0) { np1_batch_free(b); return nullptr; }
    return b;
}

np1_batch* np1_batch_create(np1_ctx* ctx) {
    if (!ctx) { np1_set_error("np1_batch_create: null context"); return nullptr; }
    np1_batch* b = new np1_batch();
    b->ctx = ctx;
    np1_ctx_retain(ctx);
    return b;
}

int np1_batch_reload(np1_batch* b, const np1_stream* st) {
    if (!b || !st) { np1_set_error("np1_batch_reload: null argument"); return -1; }
    return fill_batch(b, st, false);
}

// Registers the arrays of a host stream with the HIP runtime (page-locked): H2D copies from them run asynchronously at
// full PCIe rate.  What an ingest stage that writes straight into pinned staging gets for free.
// bytes one reload of this stream moves host -> device (what the streamed passes of bench.py divide by their time)
uint64_t np1_stream_upload_bytes(np1_stream* st) {
    if (!st) return 0;
    stream_facts(st);
    return st->upload_bytes;
}

int np1_stream_pin(np1_stream* st) {
    if (!st) return -1;
    std::lock_guard<std::mutex> one(st->pin_mu);
    if (st->pinned.load(std::memory_order_acquire)) return 0;
    stream_facts(st);
    np::ReadStream& s = st->s;
    // every array fill_batch may upload; what is small enough for the runtime's own staging stays where it is (np_hostcopy.h)
    std::vector<std::pair<const void*, size_t>> arrays;
    auto add = [&](const void* p, size_t bytes) { if (p && bytes > npcopy::kDirectMax) arrays.emplace_back(p, bytes); };
    add(s.draft.data(), s.draft.size()); add(s.pos.data(), 4 * s.pos.size()); add(s.ctg.data(), 4 * s.ctg.size());
    add(s.flag.data(), 2 * s.flag.size()); add(s.n_cigar.data(), 4 * s.n_cigar.size()); add(st->ncig16.data(), 2 * st->ncig16.size()); add(st->seq2.data(), st->seq2.size());
    add(st->esc_at.data(), 8 * st->esc_at.size()); add(st->esc_val.data(), st->esc_val.size()); add(st->draft4.data(), st->draft4.size());
    add(st->desc_at.data(), 8 * st->desc_at.size()); add(st->desc_val.data(), st->desc_val.size()); add(st->compact.plain.data(), 4 * st->compact.plain.size());
    add(st->compact.dpos.data(), st->compact.dpos.size()); add(st->compact.x_pos.data(), 4 * st->compact.x_pos.size());
    add(st->compact.x_lq.data(), 4 * st->compact.x_lq.size()); add(st->compact.x_ncig.data(), 4 * st->compact.x_ncig.size());
    add(st->compact.x_cigar.data(), 4 * st->compact.x_cigar.size()); add(s.l_qseq.data(), 4 * s.l_qseq.size());
    add(s.cigar_off.data(), 8 * s.cigar_off.size()); add(s.seq_off.data(), 8 * s.seq_off.size());
    add(s.cigar.data(), 4 * s.cigar.size()); add(s.seq.data(), s.seq.size()); add(s.mapq.data(), s.mapq.size());
    add(s.isize.data(), 4 * s.isize.size()); add(s.qual_off.data(), 8 * s.qual_off.size()); add(s.qual.data(), s.qual.size());
    add(s.read_begin.data(), 8 * s.read_begin.size()); add(s.ctg_off.data(), 4 * s.ctg_off.size());
    // (an upload that does not use a form -- seq when seq2 exists, the per-record arrays when the compact form is on -- never asks for it)
    const bool slim = !(getenv("NP1_UPLOAD") && strcmp(getenv("NP1_UPLOAD"), "full") == 0);
    const bool rebuild = slim && st->facts == 1 && s.n_reads() > 0;
    auto unused = [&](const void* p) {
        if (!st->seq2.empty() && p == s.seq.data()) return true;
        if (!st->draft4.empty() && p == s.draft.data()) return true;
        if (rebuild && (p == s.ctg.data() || p == s.cigar_off.data() || p == s.seq_off.data())) return true;
        if (rebuild && st->compact.on && (p == s.pos.data() || p == s.l_qseq.data() || p == s.cigar.data() || p == s.n_cigar.data() || p == st->ncig16.data())) return true;
        if (!(rebuild && st->compact.on) && !st->ncig16.empty() && p == s.n_cigar.data()) return true;
        return false;
    };
    size_t total = 0;
    for (auto& a : arrays) if (!unused(a.first)) total += (a.second + 63) & ~(size_t)63;
    st->arena_map.clear();
    if (total) {
        if (npalloc::host_malloc(&st->arena, total, hipHostMallocPortable) != hipSuccess) { st->arena = nullptr; np1_set_error("hipHostMalloc failed (pinned upload arena)"); return -1; }
        st->arena_bytes = total;
        size_t at = 0;
        for (auto& a : arrays) {
            if (unused(a.first)) continue;
            st->arena_map.emplace_back(a.first, static_cast<char*>(st->arena) + at);
            at += (a.second + 63) & ~(size_t)63;
        }
        // (in pieces of 16 MiB on all helper threads: the first touch of fresh page-locked memory is most of the cost)
        struct Piece { char* dst; const char* src; size_t n; };
        std::vector<Piece> pieces;
        for (auto& m : st->arena_map) {
            size_t bytes = 0;
            for (auto& a : arrays) if (a.first == m.first) bytes = a.second;
            for (size_t off = 0; off < bytes; off += (size_t)16 << 20)
                pieces.push_back(Piece{static_cast<char*>(const_cast<void*>(m.second)) + off, static_cast<const char*>(m.first) + off, std::min<size_t>((size_t)16 << 20, bytes - off)});
        }
        np::parallel_for(pieces.size(), 1, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) memcpy(pieces[i].dst, pieces[i].src, pieces[i].n);
        });
        std::sort(st->arena_map.begin(), st->arena_map.end());
    }
    st->pinned.store(true, std::memory_order_release);
    return 0;
}

void np1_batch_free(np1_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    b->release_all();
    if (b->h_pin) (void)npalloc::host_free(b->h_pin);
    np1_ctx_release(b->ctx);
    delete b;
}

// rate = R / 2^K exactly?  (scores are then exact integers, see k_dp)
static bool rate_fixed_point(double rate, int* K, long long* R) {
    if (!std::isfinite(rate)) return false;
    for (int k = 0; k <= 10; ++k) {
        double x = std::ldexp(rate, k);
        if (x == std::floor(x) && std::fabs(x) < 1e12) { *K = k; *R = (long long)x; return true; }
    }
    return false;
}

int np1_batch_score_chain(np1_batch* b, const Configure* cfg, float* stage_ms) {
    if (!b || !cfg) { np1_set_error("np1_batch_score_chain: null argument"); return -1; }
    np1_ctx* ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    hipStream_t q = ctx->stream;
    b->ran = false;
    b->out_cached = false;
    b->out_pinned = false;
    if (npalloc::poison_runs()) b->poison_work(q);
    int K = 0;
    long long Rfix = 0;
    // rate = R / 2^K (the default 0.5, 0.25, 0.75 ...): exact integer scores, multi-state runs are independent (fast path).
    // Any other double: the reference's own fp64 arithmetic in its own order, one sequential run per contig (np1_core.h).
    const bool fp_rate = !rate_fixed_point(cfg->indel_balance_factor_sgs, &K, &Rfix);
    if (fp_rate && !std::isfinite(cfg->indel_balance_factor_sgs)) { np1_set_error("indel_balance_factor_sgs is not a finite number"); return -1; }
    if (fp_rate && (use_staged() || b->force_staged)) { np1_set_error("a general indel_balance_factor_sgs needs the default (fused) launch sequence"); return -1; }
    const uint32_t flag_single = ((1.0 < cfg->min_count_ratio_skip) ? 2u : 0u) | (fp_rate ? FLAG_ALL_RECORDS : 0u);
    const uint64_t G = b->G;
    const int64_t n = b->n_reads;
    const uint32_t nc = b->nc;
    if (nc == 0) { b->h_bounds.assign(1, 0); b->S = 0; b->ran = true; return 0; }

    ReadsDev R{b->pos.as<int32_t>(), b->ctg.as<uint32_t>(), b->flag.as<uint16_t>(), b->ncig.as<uint32_t>(),
               b->lq.as<int32_t>(), b->cigoff.as<uint64_t>(), b->seqoff.as<uint64_t>(), b->cigar.as<uint32_t>(),
               b->seq.as<uint8_t>()};
    const uint32_t* ctg_off = b->ctg_off.as<uint32_t>();
    bool timing = stage_ms != nullptr;
    auto t0 = [&](int i) { if (timing) (void)hipEventRecord(ctx->ev0[i], q); };
    auto t1 = [&](int i) { if (timing) (void)hipEventRecord(ctx->ev1[i], q); };

    size_t nn = (size_t)(n > 0 ? n : 1);
    if (b->qs.ensure(4 * nn) || b->qe.ensure(4 * nn) || b->span.ensure(4 * nn) || b->rbase.ensure(4 * nn) ||
        b->capb.ensure(4 * nn) || b->rowoff.ensure(8 * (nn + 1)) || b->meta.ensure(16 * nn) ||
        b->ins.ensure(4 * (G + 1)) || b->soff.ensure(4 * (G + 2)) || b->counters.ensure(4 * CNT_WORDS) ||
        b->totals.ensure(16 * 8) || b->bounds.ensure(4 * ((size_t)nc + 1)) ||
        b->scan_tmp.ensure(8 * (scan_tmp_words(G + G / 8 + 1024) + scan_tmp_words(nn))))
        return -1;
    uint64_t* totals = b->totals.as<uint64_t>();   // [0] slots, [1] row bytes, [2] out chars, [3] votes (staged), [8..15] vote shards (fused)
    uint32_t* counters = b->counters.as<uint32_t>();
    uint64_t* scan_tmp = b->scan_tmp.as<uint64_t>();

    // ---- stage 0: prep
    HIPCHK(hipMemsetAsync(b->ins.p, 0, 4 * (G + 1), q));
    HIPCHK(hipMemsetAsync(b->counters.p, 0, 4 * CNT_WORDS, q));
    HIPCHK(hipMemsetAsync(b->totals.p, 0, 128, q));
    t0(0);
    launch_prep(q, R, n, ctg_off, cfg->trim_len_edge, b->qs.as<int32_t>(), b->qe.as<int32_t>(), b->span.as<int32_t>(),
                b->ins.as<uint32_t>(), counters);
    t1(0);
    // ---- stage 1: slot offsets
    t0(1);
    launch_scan_slots(q, b->ins.as<uint32_t>(), G, b->soff.as<uint32_t>(), scan_tmp, &totals[0]);
    t1(1);
    uint64_t S64 = 0;
    HIPCHK(npcopy::d2h(&S64, &totals[0], 8, q));
    HIPCHK(hipStreamSynchronize(q));
    if (S64 >= 0xfffffff0ull) { np1_set_error("batch too large: more than 2^32 slots"); return -1; }
    const uint32_t S = (uint32_t)S64;
    b->S = S;
    const uint32_t n_chunks = (S + VOTE_CH - 1) / VOTE_CH + 1;
    if (b->slot_info.ensure(S + 64) || b->slot_res.ensure(2 * ((size_t)S + 64)) || b->slot_rec.ensure(4 * ((size_t)S + 64)) ||
        b->opos.ensure(4 * ((size_t)S + 2)) || b->out.ensure((size_t)S + 64) || b->chunk_first.ensure(4 * (size_t)n_chunks) ||
        b->chunk_last.ensure(4 * (size_t)n_chunks) || b->heads.ensure(4 * ((size_t)S / 2 + 64)) ||
        b->redo.ensure(4 * (size_t)n_chunks) || b->redo2.ensure(4 * (size_t)n_chunks))
        return -1;
    if (scan_tmp_words((uint64_t)S + 1) * 8 > b->scan_tmp.cap && b->scan_tmp.ensure(8 * (scan_tmp_words((uint64_t)S + 1) + scan_tmp_words(nn)))) return -1;
    scan_tmp = b->scan_tmp.as<uint64_t>();
    if (b->pool.cap == 0 && b->pool.ensure(4 * (3 * (size_t)S + (1u << 20)))) return -1;
    if (fp_rate && b->pool.ensure(std::min<size_t>(4 * (20 * (size_t)S + (1u << 20)), (size_t)0xfffffff0u * 4))) return -1;   // every slot spills a record
    // ---- stage 2: per-slot draft symbols
    t0(2);
    if (!(use_staged() || b->force_staged) && b->slot_g.ensure(4 * ((size_t)S + 64))) return -1;
    launch_slotinfo(q, b->draft.as<uint8_t>(), (uint32_t)G, ctg_off, nc, b->soff.as<uint32_t>(), b->slot_info.as<uint8_t>(),
                    (use_staged() || b->force_staged) ? nullptr : b->slot_g.as<uint32_t>());
    t1(2);
    const bool staged = use_staged() || b->force_staged;
    uint32_t hc[CNT_WORDS];
    if (staged) {
        // ---- stage 3: row placement
        t0(3);
        launch_rowcap(q, R, n, ctg_off, b->soff.as<uint32_t>(), b->qs.as<int32_t>(), b->qe.as<int32_t>(), b->span.as<int32_t>(),
                      b->rbase.as<uint32_t>(), b->capb.as<uint32_t>());
        launch_scan_rows(q, b->capb.as<uint32_t>(), (uint64_t)(n > 0 ? n : 0), b->rowoff.as<uint64_t>(), scan_tmp, &totals[1]);
        t1(3);
        uint64_t row_bytes = 0;
        HIPCHK(npcopy::d2h(&row_bytes, &totals[1], 8, q));
        HIPCHK(hipStreamSynchronize(q));
        if ((row_bytes >> 2) >= 0xffffffffull) { np1_set_error("batch too large: symbol rows exceed 16 GiB"); return -1; }
        if (b->rows.ensure(row_bytes + 64, 1.02)) return -1;
        // ---- stage 4: rows
        HIPCHK(hipMemsetAsync(b->chunk_first.p, 0xff, 4 * (size_t)n_chunks, q));
        HIPCHK(hipMemsetAsync(b->chunk_last.p, 0, 4 * (size_t)n_chunks, q));
        t0(4);
        launch_rows(q, R, n, ctg_off, b->soff.as<uint32_t>(), b->qs.as<int32_t>(), b->qe.as<int32_t>(), b->rbase.as<uint32_t>(),
                    b->rowoff.as<uint64_t>(), b->rows.as<uint8_t>(), b->meta.as<uint4>(), b->chunk_first.as<uint32_t>(),
                    b->chunk_last.as<uint32_t>(), reinterpret_cast<unsigned long long*>(&totals[3]));
        t1(4);
        // ---- stage 5: vote (+ escalation for crowded slots, + pool growth)
        for (int attempt = 0;; ++attempt) {
            uint32_t pool_cap = (uint32_t)std::min<size_t>(b->pool.cap / 4, 0xfffffff0u);
            t0(5);
            launch_vote(q, 16, b->meta.as<uint4>(), b->rows.as<uint8_t>(), b->slot_info.as<uint8_t>(), S,
                        b->chunk_first.as<uint32_t>(), b->chunk_last.as<uint32_t>(), n_chunks, nullptr, 0,
                        b->slot_res.as<uint16_t>(), b->slot_rec.as<uint32_t>(), b->pool.as<uint32_t>(), pool_cap, counters,
                        b->heads.as<uint32_t>(), b->redo.as<uint32_t>(), CNT_REDO, flag_single);
            t1(5);
            HIPCHK(npcopy::d2h(hc, counters, sizeof(hc), q));
            HIPCHK(hipStreamSynchronize(q));
            if (hc[CNT_REDO]) {
                launch_vote(q, 64, b->meta.as<uint4>(), b->rows.as<uint8_t>(), b->slot_info.as<uint8_t>(), S,
                            b->chunk_first.as<uint32_t>(), b->chunk_last.as<uint32_t>(), n_chunks, b->redo.as<uint32_t>(),
                            hc[CNT_REDO], b->slot_res.as<uint16_t>(), b->slot_rec.as<uint32_t>(), b->pool.as<uint32_t>(),
                            pool_cap, counters, b->heads.as<uint32_t>(), b->redo2.as<uint32_t>(), CNT_REDO2, flag_single);
                HIPCHK(npcopy::d2h(hc, counters, sizeof(hc), q));
                HIPCHK(hipStreamSynchronize(q));
                if (hc[CNT_REDO2]) {
                    if (b->redo3.ensure(4 * (size_t)n_chunks)) return -1;
                    launch_vote(q, 160, b->meta.as<uint4>(), b->rows.as<uint8_t>(), b->slot_info.as<uint8_t>(), S,
                                b->chunk_first.as<uint32_t>(), b->chunk_last.as<uint32_t>(), n_chunks, b->redo2.as<uint32_t>(),
                                hc[CNT_REDO2], b->slot_res.as<uint16_t>(), b->slot_rec.as<uint32_t>(), b->pool.as<uint32_t>(),
                                pool_cap, counters, b->heads.as<uint32_t>(), b->redo3.as<uint32_t>(), CNT_REDO3, flag_single);
                    HIPCHK(npcopy::d2h(hc, counters, sizeof(hc), q));
                    HIPCHK(hipStreamSynchronize(q));
                    // slots with more than 160 distinct contexts (very deep pileups with ambiguity codes): every possible context gets
                    // its own list entry, in HBM -- a slice of the list at a time, 1 MiB of scratch per chunk
                    for (uint32_t at = 0; at < hc[CNT_REDO3];) {
                        const uint32_t slice = std::min<uint32_t>(hc[CNT_REDO3] - at, 1024u);
                        if (b->ctx_lists.ensure(4 * vote_hbm_list_words() * (size_t)slice)) return -1;
                        launch_vote(q, VOTE_E_ALL, b->meta.as<uint4>(), b->rows.as<uint8_t>(), b->slot_info.as<uint8_t>(), S,
                                    b->chunk_first.as<uint32_t>(), b->chunk_last.as<uint32_t>(), n_chunks, b->redo3.as<uint32_t>() + at, slice,
                                    b->slot_res.as<uint16_t>(), b->slot_rec.as<uint32_t>(), b->pool.as<uint32_t>(), pool_cap, counters,
                                    b->heads.as<uint32_t>(), nullptr, CNT_REDO3, flag_single, b->ctx_lists.as<uint32_t>());
                        at += slice;
                    }
                    if (hc[CNT_REDO3]) {
                        HIPCHK(npcopy::d2h(hc, counters, sizeof(hc), q));
                        HIPCHK(hipStreamSynchronize(q));
                    }
                }
            }
            if (hc[CNT_ERR] & ERR_POOL_OVERFLOW) {
                if (attempt >= 3) { np1_set_error("DP record pool keeps overflowing"); return -1; }
                size_t need = (size_t)hc[CNT_POOL] * 4;
                if (b->pool.ensure(need + need / 4 + (1u << 20))) return -1;
                uint32_t zero[CNT_WORDS] = {0};
                zero[CNT_ERR] = hc[CNT_ERR] & ~ERR_POOL_OVERFLOW;
                HIPCHK(npcopy::h2d(counters, zero, sizeof(zero), q));
                HIPCHK(hipStreamSynchronize(q));
                continue;
            }
            break;
        }

    } else {
        // ---- stage 3 (fused): record descriptors (+ overflow parts) and the candidate record range of every vote chunk
        if (b->desc.ensure(4 * (size_t)DESC_WORDS * nn)) return -1;
        if (b->ovf_desc.cap == 0 && b->ovf_desc.ensure(4 * (size_t)DESC_WORDS * (nn / 64 + 4096))) return -1;
        for (int attempt = 0;; ++attempt) {
            const uint32_t ovf_cap = (uint32_t)std::min<size_t>(b->ovf_desc.cap / (4 * DESC_WORDS), 0x7fffffffu);
            HIPCHK(hipMemsetAsync(b->chunk_first.p, 0xff, 4 * (size_t)n_chunks, q));
            HIPCHK(hipMemsetAsync(b->chunk_last.p, 0, 4 * (size_t)n_chunks, q));
            HIPCHK(hipMemsetAsync(&counters[CNT_OVFDESC], 0, 4, q));
            t0(3);
            launch_desc(q, R, n, ctg_off, b->soff.as<uint32_t>(), b->qs.as<int32_t>(), b->qe.as<int32_t>(), b->desc.as<uint32_t>(),
                        b->ovf_desc.as<uint32_t>(), ovf_cap, b->chunk_first.as<uint32_t>(), b->chunk_last.as<uint32_t>(), counters);
            t1(3);
            HIPCHK(npcopy::d2h(hc, counters, sizeof(hc), q));
            HIPCHK(hipStreamSynchronize(q));
            if (!(hc[CNT_ERR] & ERR_DESC_OVERFLOW)) break;
            if (hc[CNT_OVFDESC] <= ovf_cap || attempt >= 2) {
                // not a capacity problem: a record is too long for the descriptors' 16-bit query indices.  The staged launch sequence
                // (symbol rows in HBM) has no such bound; this batch takes it from the start.
                if (fp_rate) { np1_set_error("record too long for the short-read descriptors, which a general indel_balance_factor_sgs needs"); return -1; }
                b->force_staged = true;
                return np1_batch_score_chain(b, cfg, stage_ms);
            }
            if (b->ovf_desc.ensure(4 * (size_t)DESC_WORDS * ((size_t)hc[CNT_OVFDESC] + 1024))) return -1;
            uint32_t zero = hc[CNT_ERR] & ~ERR_DESC_OVERFLOW;
            HIPCHK(npcopy::h2d(&counters[CNT_ERR], &zero, 4, q));
        }
        // ---- stage 5 (fused): votes through LDS (+ escalation for crowded slots, + pool growth)
        for (int attempt = 0;; ++attempt) {
            uint32_t pool_cap = (uint32_t)std::min<size_t>(b->pool.cap / 4, 0xfffffff0u);
            unsigned long long* votes = reinterpret_cast<unsigned long long*>(&totals[8]);   // POOL_SHARDS words
            auto tile = [&](int level, const uint32_t* redo_in, uint32_t n_redo, uint32_t* redo_out, uint32_t redo_ci) {
                return launch_tile3(q, level, R, b->soff.as<uint32_t>(), b->desc.as<uint32_t>(), b->ovf_desc.as<uint32_t>(),
                                    b->chunk_first.as<uint32_t>(),
                                    b->chunk_last.as<uint32_t>(), n_chunks, redo_in, n_redo, b->slot_info.as<uint8_t>(),
                                    b->slot_g.as<uint32_t>(), S, b->max_lq, b->slot_res.as<uint16_t>(),
                                    b->slot_rec.as<uint32_t>(), b->pool.as<uint32_t>(), pool_cap, counters,
                                    b->heads.as<uint32_t>(), (uint32_t)std::min<size_t>(b->heads.cap / 4, 0xfffffff0u), redo_out,
                                    redo_ci, flag_single, votes);
            };
            HIPCHK(hipMemsetAsync(&totals[8], 0, 8 * POOL_SHARDS, q));
            t0(5);
            // Default: k_tile3.  NP1_TILE=9: k_tile9 (np1_tile9.h; round 4: four slots per lane, agreeing records counted per window), k_tile3
            // for the chunks it hands back -- exact (GPU-tested in a subprocess: tests/test_gpu_score_chain.py) but slower on MI355X as it stands
            // (DESIGN.md section 7 has its phase clocks).  The general-rate fp64 path spills a record for every slot and stays with k_tile3.
            static const int tile_kind_env = getenv("NP1_TILE") ? atoi(getenv("NP1_TILE")) : 3;
            const int tile_kind = fp_rate ? 3 : tile_kind_env;
            const uint32_t heads_cap5 = (uint32_t)std::min<size_t>(b->heads.cap / 4, 0xfffffff0u);
            static const bool t9_phases = getenv("NP1_T9_PHASES") != nullptr;      // k_tile9's phase clocks to stderr
            unsigned long long* dbg = nullptr;
            if (t9_phases && tile_kind == 9) {
                if (b->dbg.ensure(64 * 128)) return -1;
                dbg = b->dbg.as<unsigned long long>();
                HIPCHK(hipMemsetAsync(dbg, 0, 64 * 128, q));
            }
            int rc5;
            if (tile_kind == 9)
                rc5 = launch_tile9(q, R, b->soff.as<uint32_t>(), b->desc.as<uint32_t>(), b->ovf_desc.as<uint32_t>(), b->chunk_first.as<uint32_t>(),
                                   b->chunk_last.as<uint32_t>(), n_chunks, b->slot_info.as<uint8_t>(), b->slot_g.as<uint32_t>(), S, b->max_lq,
                                   b->slot_res.as<uint16_t>(), b->slot_rec.as<uint32_t>(), b->pool.as<uint32_t>(), pool_cap, counters,
                                   b->heads.as<uint32_t>(), heads_cap5, b->redo.as<uint32_t>(), CNT_REDO, flag_single, votes, dbg);
            else
                rc5 = tile(0, nullptr, 0, b->redo.as<uint32_t>(), CNT_REDO);
            if (dbg) {
                unsigned long long hs[64 * 16], h[16] = {0};
                HIPCHK(npcopy::d2h(hs, dbg, sizeof(hs), q));
                HIPCHK(hipStreamSynchronize(q));
                for (int sh = 0; sh < 64; ++sh)
                    for (int k = 0; k < 16; ++k) h[k] += hs[16 * sh + k];
                const double nw = h[6] ? (double)h[6] : 1.0;
                fprintf(stderr, "[k_tile9 cycles/wave] setup %.0f  staging %.0f  records %.0f  evaluation %.0f  tally %.0f  epilogue %.0f | waves %llu  steps/wave %.1f  entries/wave %.1f  rounds/wave %.2f  hot lanes/wave %.2f\n",
                        h[0] / nw, h[1] / nw, h[2] / nw, h[3] / nw, h[4] / nw, h[5] / nw, h[6], h[7] / nw, h[8] / nw, h[9] / nw, h[10] / nw);
            }
            if (rc5 != 0) {   // the longest record's bases do not fit the tile's LDS plan: the staged sequence reads them from HBM rows
                if (fp_rate) { np1_set_error("records are too long for the LDS-staged path, which a general indel_balance_factor_sgs needs"); return -1; }
                b->force_staged = true;
                return np1_batch_score_chain(b, cfg, stage_ms);
            }
            t1(5);
            HIPCHK(npcopy::d2h(hc, counters, sizeof(hc), q));
            HIPCHK(hipStreamSynchronize(q));
            if (hc[CNT_REDO]) {
                if (tile(1, b->redo.as<uint32_t>(), hc[CNT_REDO], b->redo2.as<uint32_t>(), CNT_REDO2) != 0) { np1_set_error("LDS plan failed (level 1)"); return -1; }
                HIPCHK(npcopy::d2h(hc, counters, sizeof(hc), q));
                HIPCHK(hipStreamSynchronize(q));
                if (hc[CNT_REDO2]) {
                    if (tile(2, b->redo2.as<uint32_t>(), hc[CNT_REDO2], nullptr, CNT_REDO2) != 0) { np1_set_error("LDS plan failed (level 2)"); return -1; }
                    HIPCHK(npcopy::d2h(hc, counters, sizeof(hc), q));
                    HIPCHK(hipStreamSynchronize(q));
                    if ((hc[CNT_ERR] & ERR_CTX_OVERFLOW) && !fp_rate) {
                        // more than 160 distinct contexts in a slot: the staged sequence keeps such slots' lists in HBM; this batch takes
                        // it from the start (a pileup this deep and this ambiguous is not a throughput case)
                        b->force_staged = true;
                        return np1_batch_score_chain(b, cfg, stage_ms);
                    }
                }
            }
            if (hc[CNT_ERR] & ERR_POOL_OVERFLOW) {
                if (getenv("NP1_DEBUG_POOL")) {
                    fprintf(stderr, "[np1 pool] attempt %d: S %u, pool %zu words, heads %zu entries; per shard used (pool / heads):", attempt, S, b->pool.cap / 4, b->heads.cap / 4);
                    for (uint32_t sh = 0; sh < POOL_SHARDS; ++sh) fprintf(stderr, " %u/%u", hc[CNT_POOL_S0 + sh], hc[CNT_HEADS_S0 + sh]);
                    fprintf(stderr, "\n");
                }
                if (attempt >= 3) { np1_set_error("DP record pool keeps overflowing"); return -1; }
                size_t need = 2 * b->pool.cap;   // a shard ran out: double the pool (and the run-head list with it)
                if (b->heads.ensure(2 * b->heads.cap)) return -1;
                if (b->pool.ensure(need + need / 4 + (1u << 20))) return -1;
                uint32_t zero[CNT_WORDS] = {0};
                zero[CNT_ERR] = hc[CNT_ERR] & ~ERR_POOL_OVERFLOW;
                HIPCHK(npcopy::h2d(counters, zero, sizeof(zero), q));
                HIPCHK(hipStreamSynchronize(q));
                continue;
            }
            break;
        }
    }
    if (hc[CNT_ERR] & ERR_DOUBLE_INS) { np1_set_error("unsupported CIGAR: two insertion ops at one reference position"); return -1; }
    if (hc[CNT_ERR] & ERR_BAD_RECORD) { np1_set_error("alignment record extends beyond its contig"); return -1; }
    if (hc[CNT_ERR] & ERR_CTX_OVERFLOW) { np1_set_error("a slot holds more than 160 distinct 3-base contexts, which a general indel_balance_factor_sgs cannot take"); return -1; }
    if (b->keep_single) {      // intra-contig tiling: the single-state slots of the vote, before the chain overwrites the others
        if (b->single_map.ensure((size_t)S + 64)) return -1;
        launch_single_map(q, b->slot_res.as<uint16_t>(), S, b->single_map.as<uint8_t>());
    }
    // ---- stage 6: chain DP over multi-state runs
    t0(6);
    {
        uint32_t heads = hc[CNT_HEADS];
        if (!staged) {
            heads = 0;
            for (uint32_t sh = 0; sh < POOL_SHARDS; ++sh) heads += hc[CNT_HEADS_S0 + sh];
        }
        uint32_t grid = (heads + 63) / 64;
        if (grid == 0) grid = 1;
        const uint32_t heads_cap = (uint32_t)std::min<size_t>(b->heads.cap / 4, 0xfffffff0u);
        launch_dp(q, b->heads.as<uint32_t>(), counters, staged ? (uint32_t)CNT_HEADS : (uint32_t)CNT_HEADS_S0,
                  staged ? 1u : POOL_SHARDS, staged ? 0u : heads_cap / POOL_SHARDS, b->pool.as<uint32_t>(),
                  b->slot_rec.as<uint32_t>(), b->slot_res.as<uint16_t>(), K, Rfix, cfg->min_count_ratio_skip, grid, fp_rate,
                  cfg->indel_balance_factor_sgs);
    }
    t1(6);
    // ---- stage 7: emit
    t0(7);
    launch_fixfirst(q, ctg_off, nc, b->soff.as<uint32_t>(), b->slot_info.as<uint8_t>(), b->slot_res.as<uint16_t>());
    launch_scan_keep(q, b->slot_res.as<uint16_t>(), S, b->opos.as<uint32_t>(), scan_tmp, &totals[2]);
    launch_emit(q, b->slot_res.as<uint16_t>(), b->slot_info.as<uint8_t>(), b->opos.as<uint32_t>(), S, 1u | 2u, b->out.as<uint8_t>());
    launch_contig_bounds(q, ctg_off, nc, b->soff.as<uint32_t>(), b->opos.as<uint32_t>(), b->bounds.as<uint32_t>());
    t1(7);
    b->h_bounds.resize((size_t)nc + 1);
    HIPCHK(npcopy::d2h(b->h_bounds.data(), b->bounds.p, 4 * ((size_t)nc + 1), q));
    uint64_t h_votes[1 + POOL_SHARDS];
    HIPCHK(npcopy::d2h(&h_votes[0], &totals[3], 8, q));
    HIPCHK(npcopy::d2h(&h_votes[1], &totals[8], 8 * POOL_SHARDS, q));
    HIPCHK(npcopy::d2h(hc, counters, sizeof(hc), q));
    HIPCHK(hipStreamSynchronize(q));
    b->votes = 0;
    for (uint32_t i = 0; i < 1 + POOL_SHARDS; ++i) b->votes += h_votes[i];
    if (hc[CNT_ERR] & ERR_DP_INCONSISTENT) { np1_set_error("inconsistent pileup state in the chain DP"); return -1; }
    memcpy(b->last_counters, hc, sizeof(hc));
    b->votes += S;   // the draft votes once per slot
    if (timing) {
        for (int i = 0; i < NP1_MAX_STAGES; ++i) stage_ms[i] = 0.f;
        for (int i = 0; i < kStages; ++i) (void)hipEventElapsedTime(&stage_ms[i], ctx->ev0[i], ctx->ev1[i]);
    }
    b->ran = true;
    return 0;
}


// ---- kmer_count (task 2) and snp_valid (task 4): launch sequence over the same HBM-resident batch (needs qualities) ----------
// snp_valid (snpvalid.c:3-36) is kmer_count's haplotype vote without the no-depth regions, run twice: the first round leaves the
// FLAG_ZERO marks to the winners (a part that gets one loses its marks), the regions nothing spanned are cut again at the middle of
// their unmarked runs (fts_spilt_region) and voted on once more; the result is emitted without lower case.
// The votes of ss_kmer_correct (kmercount.c:175-261) on what the reference's region iterator hands out (np1_replay.h), for kmer_count,
// for both rounds of snp_valid, on every path that knows the file geometry (host loader and device-side ingest alike).  The pairs of a
// contig are one contiguous run of the part arrays, in list order (slots marked 0xffffffff are unused; contig | 1 << 31 marks an
// inverted pair of snp_valid's second round: nothing is voted there, but its end is the `nextposend` of the pair before it).
//   host: first loop of every part (record lists, buffered record)  ->  device: votes; 2 = the first loop left nothing; where the
//   loop left through the max_count_kmer break (kmercount.c:201-203) the device says after how many records  ->  host: the replay
//   again with those breaks (a re-used iterator resumes from where the break left it); only if that changes a list do the votes run
//   again, until lists and breaks agree  ->  host: passes of the second loop over the empty parts  ->  device: the votes with them.
static int replay_votes(np1_batch* b, const KcCtx& c, const uint32_t* d_ctg, const int32_t* d_se, const uint32_t* d_len, const uint32_t* d_woff, uint32_t n_parts,
                        int64_t n_all, uint8_t* d_wpool, uint8_t* d_haswin, hipStream_t q) {
    np1_batch::Replay& R = b->replay;
    if (!R.have_pos) {     // once per pass: positions and end positions of the records as the kernels see them
        if (!R.pos.ensure(4 * (size_t)(n_all + 1)) || !R.endpos.ensure(4 * (size_t)(n_all + 1))) { np1_set_error("hipHostMalloc failed"); return -1; }
        if (n_all) {
            HIPCHK(npcopy::d2h(R.pos.p, b->pos.p, 4 * (size_t)n_all, q));
            HIPCHK(npcopy::d2h(R.endpos.p, b->kc_endpos.p, 4 * (size_t)n_all, q));
        }
        R.have_pos = true;
    }
    std::vector<uint32_t> pt_ctg(n_parts);
    std::vector<int32_t> pt_se(2 * (size_t)n_parts);
    HIPCHK(npcopy::d2h(pt_ctg.data(), d_ctg, 4 * (size_t)n_parts, q));
    HIPCHK(npcopy::d2h(pt_se.data(), d_se, 8 * (size_t)n_parts, q));
    HIPCHK(hipStreamSynchronize(q));
    struct Run { uint32_t ct, p0, p1; };
    std::vector<Run> runs;
    std::vector<uint8_t> skip(n_parts, 0);
    for (uint32_t p = 0; p < n_parts;) {
        if (pt_ctg[p] == 0xffffffffu) { skip[p] = 1; ++p; continue; }
        const uint32_t ct = pt_ctg[p] & 0x7fffffffu;
        uint32_t e = p;
        while (e < n_parts && pt_ctg[e] != 0xffffffffu && (pt_ctg[e] & 0x7fffffffu) == ct) { skip[e] = (pt_ctg[e] & 0x80000000u) ? 1 : 0; ++e; }
        runs.push_back(Run{ct, p, e});
        p = e;
    }
    std::vector<int32_t> next_end(n_parts, -1);
    for (const Run& r : runs)
        for (uint32_t p = r.p0; p + 1 < r.p1; ++p) next_end[p] = pt_se[2 * (size_t)(p + 1) + 1];
    auto known = [&](const Run& r) { return R.tid[r.ct] >= 0 && (size_t)R.tid[r.ct] < R.bai->refs.size(); };   // else: a contig the BAM does not know, no records
    auto records = [&](uint32_t ct) {
        const int64_t rb = (int64_t)b->h_read_begin[ct], re = (int64_t)b->h_read_begin[ct + 1];
        return np1replay::Records{R.voff + rb, R.voff_end + rb, R.pos.as<int32_t>() + rb, R.endpos.as<int32_t>() + rb, re - rb, re < n_all,
                                  (int32_t)(b->h_ctg_off[ct + 1] - b->h_ctg_off[ct])};
    };
    struct Lists { std::vector<uint32_t> first, list; std::vector<long long> stale; };
    auto replay = [&](const std::vector<uint32_t>& limit, Lists* out) {
        std::vector<np1replay::FirstLoop> per(runs.size());
        np::parallel_for(runs.size(), 1, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const Run& r = runs[i];
                if (!known(r)) continue;
                const np1replay::RefIndex ix(R.bai->refs[(size_t)R.tid[r.ct]]);
                per[i] = np1replay::first_loop(ix, records(r.ct), pt_se.data() + 2 * (size_t)r.p0, next_end.data() + r.p0, r.p1 - r.p0, limit.data() + r.p0, skip.data() + r.p0);
            }
        });
        out->first.assign((size_t)n_parts + 1, 0);
        out->list.clear();
        out->stale.assign(n_parts, -1);
        uint32_t p = 0;
        for (size_t i = 0; i < runs.size(); ++i) {
            const Run& r = runs[i];
            for (; p < r.p0; ++p) out->first[p + 1] = (uint32_t)out->list.size();
            const uint32_t rb = (uint32_t)b->h_read_begin[r.ct];
            const np1replay::FirstLoop& fl = per[i];
            for (; p < r.p1; ++p) {
                if (!fl.first.empty()) {
                    for (uint32_t t = fl.first[p - r.p0]; t < fl.first[p - r.p0 + 1]; ++t) out->list.push_back(fl.list[t] + rb);
                    out->stale[p] = fl.stale[p - r.p0] >= 0 ? fl.stale[p - r.p0] + (long long)rb : -1;
                }
                out->first[p + 1] = (uint32_t)out->list.size();
            }
        }
        for (; p < n_parts; ++p) out->first[p + 1] = (uint32_t)out->list.size();
    };
    std::vector<uint8_t> state(n_parts);
    std::vector<uint32_t> brk(n_parts, 0), limit(n_parts, 0);
    if (R.first.ensure(4 * ((size_t)n_parts + 2)) || R.stale.ensure(8 * ((size_t)n_parts + 1)) || R.n2.ensure(4 * ((size_t)n_parts + 1)) || R.brk.ensure(4 * ((size_t)n_parts + 1))) return -1;
    auto vote = [&](const Lists& L, const int32_t* d_n2) -> int {
        if (R.list.ensure(4 * (L.list.size() + 2))) return -1;
        HIPCHK(npcopy::h2d(R.first.p, L.first.data(), 4 * ((size_t)n_parts + 1), q));
        if (!L.list.empty()) HIPCHK(npcopy::h2d(R.list.p, L.list.data(), 4 * L.list.size(), q));
        HIPCHK(npcopy::h2d(R.stale.p, L.stale.data(), 8 * (size_t)n_parts, q));
        HIPCHK(hipMemsetAsync(c.hcount, 0, 4, q));     // every launch takes the haplotype pool from its start
        kc_launch_winner_replay(q, c, d_ctg, d_se, d_len, d_woff, n_parts, n_all, d_wpool, d_haswin, R.first.as<uint32_t>(), R.list.as<uint32_t>(), R.stale.as<long long>(), d_n2,
                                d_n2 ? nullptr : R.brk.as<uint32_t>());
        if (!d_n2) {
            HIPCHK(npcopy::d2h(state.data(), d_haswin, n_parts, q));
            HIPCHK(npcopy::d2h(brk.data(), R.brk.p, 4 * (size_t)n_parts, q));
        }
        HIPCHK(hipStreamSynchronize(q));   // (the host vectors are the source of the copies above)
        return 0;
    };
    // a covering record clears FLAG_ZERO marks whether its haplotype counts or not (kmercount.c:398-437): votes that have to be redone
    // on other records start from the marks as they were
    if (!c.keep_zero_marks && b->S) {
        if (R.snap.ensure((size_t)b->S + 64)) return -1;
        HIPCHK(hipMemcpyAsync(R.snap.p, c.sflag, b->S, hipMemcpyDeviceToDevice, q));
    }
    Lists cur, nxt;
    replay(limit, &cur);
    if (vote(cur, nullptr) != 0) return -1;
    R.revotes = 0;
    for (int it = 0; brk != limit; ++it) {
        if (it > 256) { np1_set_error("kmer_count: the replay of the region iterator does not settle on this input"); return -1; }
        limit = brk;
        replay(limit, &nxt);
        bool same = true;
        for (uint32_t p = 0; p < n_parts && same; ++p) {
            const uint32_t n_old = cur.first[p + 1] - cur.first[p], n_new = nxt.first[p + 1] - nxt.first[p], n_cmp = brk[p] ? brk[p] : n_old;
            same = n_new == n_cmp && n_cmp <= n_old && std::equal(nxt.list.begin() + nxt.first[p], nxt.list.begin() + nxt.first[p + 1], cur.list.begin() + cur.first[p]) &&
                   (state[p] != 2 || nxt.stale[p] == cur.stale[p]);
        }
        cur.first.swap(nxt.first); cur.list.swap(nxt.list); cur.stale.swap(nxt.stale);
        if (same) break;                   // the votes that ran saw exactly these records: nothing to redo
        if (!c.keep_zero_marks && b->S) HIPCHK(hipMemcpyAsync(c.sflag, R.snap.p, b->S, hipMemcpyDeviceToDevice, q));
        if (vote(cur, nullptr) != 0) return -1;
        ++R.revotes;
    }
    bool any = false;
    std::vector<uint8_t> empty(n_parts, 0);
    for (uint32_t p = 0; p < n_parts; ++p) { empty[p] = state[p] == 2; any = any || empty[p]; }
    if (!any) return 0;
    std::vector<int32_t> n2(n_parts, 0);
    np::parallel_for(runs.size(), 1, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const Run& r = runs[i];
            if (!known(r)) continue;
            bool need = false;
            for (uint32_t p = r.p0; p < r.p1 && !need; ++p) need = empty[p];
            if (!need) continue;
            const np1replay::RefIndex ix(R.bai->refs[(size_t)R.tid[r.ct]]);
            const std::vector<uint32_t> k = np1replay::second_loop_passes(ix, records(r.ct), pt_se.data() + 2 * (size_t)r.p0, next_end.data() + r.p0, r.p1 - r.p0, empty.data() + r.p0);
            for (uint32_t p = r.p0; p < r.p1; ++p) n2[p] = (int32_t)k[p - r.p0];
        }
    });
    HIPCHK(npcopy::h2d(R.n2.p, n2.data(), 4 * (size_t)n_parts, q));
    return vote(cur, R.n2.as<int32_t>());
}

static int kmer_pipeline(np1_batch* b, const Configure* cfg, bool snp_valid) {
    const char* task = snp_valid ? "snp_valid" : "kmer_count";
    if (!b || !cfg) { np1_set_error(std::string(task) + ": null argument"); return -1; }
    if (!b->has_qual) { np1_set_error(std::string(task) + " needs a stream loaded with base qualities"); return -1; }
    np1_ctx* ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    hipStream_t q = ctx->stream;
    b->ran = false;
    b->out_cached = false;
    b->out_pinned = false;
    b->replay.have_pos = false;
    if (npalloc::poison_runs()) b->poison_work(q);
    int K = 0;
    long long Rfix = 0;
    if (!rate_fixed_point(cfg->indel_balance_factor_sgs, &K, &Rfix)) {   // general rate: the region DP keeps doubles (np1_kmer.h)
        if (!std::isfinite(cfg->indel_balance_factor_sgs)) { np1_set_error("indel_balance_factor_sgs is not a finite number"); return -1; }
        K = -1;
        Rfix = 0;
    }
    const uint64_t G = b->G;
    const int64_t n = b->n_reads;
    const uint32_t nc = b->nc;
    if (nc == 0) { b->h_bounds.assign(1, 0); b->S = 0; b->ran = true; return 0; }
    const size_t nn = (size_t)(n > 0 ? n : 1);
    if (b->kc_level.ensure(nn) || b->kc_endpos.ensure(4 * nn) || b->kc_code.ensure(G + 1) || b->kc_flag.ensure(G + 1) ||
        b->kc_fpos.ensure(4 * (G + 2)) || b->kc_cnt.ensure(4 * KCC_WORDS) || b->ins.ensure(4 * (G + 1)) || b->soff.ensure(4 * (G + 2)) ||
        b->totals.ensure(64) || b->counters.ensure(4 * CNT_WORDS) || b->bounds.ensure(4 * ((size_t)nc + 1)) ||
        b->scan_tmp.ensure(8 * (scan_tmp_words(G + G / 8 + 1024) + scan_tmp_words(nn))))
        return -1;
    uint64_t* totals = b->totals.as<uint64_t>();
    uint64_t* scan_tmp = b->scan_tmp.as<uint64_t>();
    uint32_t* kcnt = b->kc_cnt.as<uint32_t>();
    const uint32_t* ctg_off = b->ctg_off.as<uint32_t>();

    KcCtx c;
    memset(&c, 0, sizeof(c));
    c.R = ReadsDev{b->pos.as<int32_t>(), b->ctg.as<uint32_t>(), b->flag.as<uint16_t>(), b->ncig.as<uint32_t>(), b->lq.as<int32_t>(),
                   b->cigoff.as<uint64_t>(), b->seqoff.as<uint64_t>(), b->cigar.as<uint32_t>(), b->seq.as<uint8_t>()};
    c.mapq = b->mapq.as<uint8_t>(); c.isize = b->isize.as<int32_t>(); c.qual_off = b->qualoff.as<uint64_t>(); c.qual = b->qual.as<uint8_t>();
    c.level = b->kc_level.as<uint8_t>(); c.endpos = b->kc_endpos.as<int32_t>();
    c.ctg_off = ctg_off; c.read_begin = b->read_begin.as<uint64_t>();
    c.draft_code = b->kc_code.as<uint8_t>(); c.draft_flag = b->kc_flag.as<uint8_t>();
    c.trim = cfg->trim_len_edge; c.ext_len_edge = cfg->ext_len_edge; c.min_len_ldr = cfg->min_len_ldr;
    c.min_len_inter_kmer = cfg->min_len_inter_kmer; c.max_len_kmer = cfg->max_len_kmer; c.max_count_kmer = cfg->max_count_kmer;
    c.min_map_quality = cfg->min_map_quality; c.read_tlen = cfg->read_tlen;
    c.max_clip_ratio_sgs = cfg->max_clip_ratio_sgs; c.min_count_ratio_skip = cfg->min_count_ratio_skip;
    c.K = K; c.Rfix = Rfix; c.rate = cfg->indel_balance_factor_sgs;
    c.err = &kcnt[KCC_ERR];

    for (int attempt = 0; attempt < 5; ++attempt) {
        const size_t scale = (size_t)1 << (2 * attempt);   // pool growth on overflow: x1, x4, x16 ...
        uint32_t hk[KCC_WORDS];
        HIPCHK(hipMemsetAsync(kcnt, 0, 4 * KCC_WORDS, q));
        HIPCHK(hipMemsetAsync(b->totals.p, 0, 64, q));
        // ---- records, draft flags, compacted lowercase positions
        kc_launch_records(q, c, n, b->kc_level.as<uint8_t>(), b->kc_endpos.as<int32_t>(), &kcnt[KCC_MAXSPAN]);
        kc_launch_draft(q, b->draft.as<uint8_t>(), (uint32_t)G, b->kc_code.as<uint8_t>(), b->kc_flag.as<uint8_t>());
        launch_scan_u8(q, b->kc_flag.as<uint8_t>(), G, b->kc_fpos.as<uint32_t>(), scan_tmp, &totals[0]);
        uint64_t M = 0;
        HIPCHK(npcopy::d2h(&M, &totals[0], 8, q));
        HIPCHK(hipStreamSynchronize(q));
        const uint32_t reg_cap = (uint32_t)(M + nc + 16);
        if (b->kc_flagged.ensure(4 * (M + 4)) || b->kc_work.ensure(4 * (12 * M + 4ull * nc + 64)) || b->kc_nd_ctg.ensure(4ull * reg_cap) ||
            b->kc_nd_se.ensure(8ull * reg_cap) || b->kc_kr_ctg.ensure(4ull * reg_cap) || b->kc_kr_se.ensure(8ull * reg_cap))
            return -1;
        kc_launch_compact(q, b->kc_flag.as<uint8_t>(), b->kc_fpos.as<uint32_t>(), (uint32_t)G, b->kc_flagged.as<uint32_t>());
        kc_launch_regions(q, c, nc, b->kc_fpos.as<uint32_t>(), b->kc_flagged.as<uint32_t>(), b->kc_work.as<int32_t>(),
                          b->kc_nd_ctg.as<uint32_t>(), b->kc_nd_se.as<int32_t>(), b->kc_kr_ctg.as<uint32_t>(), b->kc_kr_se.as<int32_t>(),
                          reg_cap, kcnt);
        HIPCHK(npcopy::d2h(hk, kcnt, sizeof(hk), q));
        HIPCHK(hipStreamSynchronize(q));
        if (hk[KCC_ERR]) { np1_set_error(std::string(task) + ": region discovery failed"); return -1; }
        const uint32_t n_nd = snp_valid ? 0u : hk[KCC_NODEPTH], n_kr = hk[KCC_KREG];   // snp_valid has no no-depth regions
        const uint64_t nd_len = snp_valid ? 0ull : ((uint64_t)hk[KCC_ND_LEN] | (uint64_t)hk[KCC_ND_LEN + 1] << 32);
        c.keep_zero_marks = snp_valid ? 1 : 0;
        c.max_span = (int32_t)(hk[KCC_MAXSPAN] ? hk[KCC_MAXSPAN] : 1);
        // ---- insertion columns of the regions, slot space
        HIPCHK(hipMemsetAsync(b->ins.p, 0, 4 * (G + 1), q));
        kc_launch_inserts(q, c, b->kc_kr_ctg.as<uint32_t>(), b->kc_kr_se.as<int32_t>(), n_kr, b->kc_nd_ctg.as<uint32_t>(),
                          b->kc_nd_se.as<int32_t>(), n_nd, b->ins.as<uint32_t>());
        launch_scan_slots(q, b->ins.as<uint32_t>(), G, b->soff.as<uint32_t>(), scan_tmp, &totals[1]);
        uint64_t S64 = 0;
        HIPCHK(npcopy::d2h(&S64, &totals[1], 8, q));
        HIPCHK(hipStreamSynchronize(q));
        if (S64 >= 0xfffffff0ull) { np1_set_error("batch too large: more than 2^32 slots"); return -1; }
        const uint32_t S = (uint32_t)S64;
        b->S = S;
        const size_t lcap = std::min<size_t>(((size_t)64 * (nd_len + (S - G)) + ((size_t)1 << 20)) * scale, (size_t)0x7ffffff0u);
        const size_t stcap = std::min<size_t>((4 * nd_len + 64ull * n_nd + 4ull * (S - G) + 4096) * scale, (size_t)0x0ffffff0u);
        if (b->slot_info.ensure(S + 64) || b->slot_res.ensure(2 * ((size_t)S + 64)) || b->opos.ensure(4 * ((size_t)S + 2)) ||
            b->out.ensure((size_t)S + 64) || b->kc_sbase.ensure(S + 64) || b->kc_sflag.ensure(S + 64) ||
            b->kc_srefk.ensure(2 * ((size_t)S + 64)) || b->kc_scount.ensure(2 * ((size_t)S + 64)) || b->kc_lhead.ensure(4 * ((size_t)S + 64)) ||
            b->kc_lpool.ensure(8 * lcap) || b->kc_stsc.ensure(8 * 16 * stcap) || b->kc_stkm.ensure(2 * 16 * stcap) || b->kc_strk.ensure(16 * stcap))
            return -1;
        if (scan_tmp_words((uint64_t)S + 1) * 8 > b->scan_tmp.cap && b->scan_tmp.ensure(8 * (scan_tmp_words((uint64_t)S + 1) + scan_tmp_words(nn)))) return -1;
        scan_tmp = b->scan_tmp.as<uint64_t>();
        launch_slotinfo(q, b->draft.as<uint8_t>(), (uint32_t)G, ctg_off, nc, b->soff.as<uint32_t>(), b->slot_info.as<uint8_t>(), nullptr);
        kc_launch_slots(q, b->slot_info.as<uint8_t>(), S, b->kc_sbase.as<uint8_t>(), b->kc_sflag.as<uint8_t>(), b->kc_scount.as<uint16_t>(),
                        b->kc_lhead.as<uint32_t>());
        c.soff = b->soff.as<uint32_t>(); c.sbase = b->kc_sbase.as<uint8_t>(); c.sflag = b->kc_sflag.as<uint8_t>();
        c.srefk = b->kc_srefk.as<uint16_t>(); c.scount = b->kc_scount.as<uint16_t>(); c.lhead = b->kc_lhead.as<uint32_t>();
        c.lpool = b->kc_lpool.as<uint32_t>(); c.lcap = (uint32_t)lcap; c.lcount = &kcnt[KCC_LCOUNT];
        c.st_score = b->kc_stsc.as<long long>(); c.st_kmer = b->kc_stkm.as<uint16_t>(); c.st_rank = b->kc_strk.as<uint8_t>();
        c.st_cap = (uint32_t)stcap; c.st_count = &kcnt[KCC_STCOUNT];
        // ---- no-depth regions: level-2 / level-1 score chain
        if (n_nd) kc_launch_nodepth(q, c, b->kc_nd_ctg.as<uint32_t>(), b->kc_nd_se.as<int32_t>(), n_nd);
        // ---- split the k-mer regions into parts (host only lays out the per-region scratch rows)
        uint32_t n_parts = 0;
        uint64_t W = 0;
        if (n_kr) {
            std::vector<int32_t> se(2 * (size_t)n_kr);
            HIPCHK(npcopy::d2h(se.data(), b->kc_kr_se.p, 8 * (size_t)n_kr, q));
            HIPCHK(hipStreamSynchronize(q));
            std::vector<uint32_t> woff((size_t)n_kr + 1, 0);
            for (uint32_t i = 0; i < n_kr; ++i) woff[i + 1] = woff[i] + (uint32_t)(se[2 * i + 1] - se[2 * i]) + 8u;
            if (b->kc_workoff.ensure(4 * ((size_t)n_kr + 1)) || b->kc_work.ensure(4 * ((size_t)woff[n_kr] + 16)) ||
                b->kc_nparts.ensure(4 * ((size_t)n_kr + 2)) || b->kc_partoff.ensure(4 * ((size_t)n_kr + 2)))
                return -1;
            HIPCHK(npcopy::h2d(b->kc_workoff.p, woff.data(), 4 * ((size_t)n_kr + 1), q));
            kc_launch_split(q, c, b->kc_kr_ctg.as<uint32_t>(), b->kc_kr_se.as<int32_t>(), n_kr, b->kc_work.as<int32_t>(),
                            b->kc_workoff.as<uint32_t>(), b->kc_nparts.as<uint32_t>(), nullptr, nullptr, nullptr, nullptr);
            launch_scan_u32(q, b->kc_nparts.as<uint32_t>(), n_kr, b->kc_partoff.as<uint32_t>(), scan_tmp, &totals[2]);
            uint64_t np64 = 0;
            HIPCHK(npcopy::d2h(&np64, &totals[2], 8, q));
            HIPCHK(hipStreamSynchronize(q));   // also keeps woff alive until the copy above is done
            n_parts = (uint32_t)np64;
            if (b->kc_pt_ctg.ensure(4 * ((size_t)n_parts + 1)) || b->kc_pt_se.ensure(8 * ((size_t)n_parts + 1)) ||
                b->kc_pt_len.ensure(4 * ((size_t)n_parts + 2)) || b->kc_woff.ensure(4 * ((size_t)n_parts + 2)) ||
                b->kc_haswin.ensure((size_t)n_parts + 1))
                return -1;
            kc_launch_split(q, c, b->kc_kr_ctg.as<uint32_t>(), b->kc_kr_se.as<int32_t>(), n_kr, b->kc_work.as<int32_t>(),
                            b->kc_workoff.as<uint32_t>(), b->kc_nparts.as<uint32_t>(), b->kc_partoff.as<uint32_t>(),
                            b->kc_pt_ctg.as<uint32_t>(), b->kc_pt_se.as<int32_t>(), b->kc_pt_len.as<uint32_t>());
            launch_scan_u32(q, b->kc_pt_len.as<uint32_t>(), n_parts, b->kc_woff.as<uint32_t>(), scan_tmp, &totals[3]);
            HIPCHK(npcopy::d2h(&W, &totals[3], 8, q));
            HIPCHK(hipStreamSynchronize(q));
            const size_t hcap = std::min<size_t>(((size_t)8192 * n_parts + ((size_t)64 << 20)) * scale, (size_t)0xfffffff0u);
            if (b->kc_wpool.ensure(W + 64) || b->kc_hpool.ensure(hcap)) return -1;
            c.hpool = b->kc_hpool.as<uint8_t>(); c.hcap = (uint32_t)hcap; c.hcount = &kcnt[KCC_HCOUNT];
            // ---- spanning-read haplotype vote, then the writes in part order
            if (b->replay.on) {
                if (replay_votes(b, c, b->kc_pt_ctg.as<uint32_t>(), b->kc_pt_se.as<int32_t>(), b->kc_pt_len.as<uint32_t>(), b->kc_woff.as<uint32_t>(), n_parts, n,
                                 b->kc_wpool.as<uint8_t>(), b->kc_haswin.as<uint8_t>(), q) != 0)
                    return -1;
            } else
            kc_launch_winner(q, c, b->kc_pt_ctg.as<uint32_t>(), b->kc_pt_se.as<int32_t>(), b->kc_pt_len.as<uint32_t>(),
                             b->kc_woff.as<uint32_t>(), n_parts, n, b->kc_wpool.as<uint8_t>(), b->kc_haswin.as<uint8_t>());
            if (!snp_valid) {
                kc_launch_apply(q, c, b->kc_pt_ctg.as<uint32_t>(), b->kc_pt_se.as<int32_t>(), b->kc_pt_len.as<uint32_t>(),
                                b->kc_woff.as<uint32_t>(), n_parts, b->kc_wpool.as<uint8_t>(), b->kc_haswin.as<uint8_t>());
            } else {
                // ---- round 1 in part order per contig, then the second round on what nothing spanned
                if (b->sv_failse.ensure(8 * ((size_t)n_parts + 1)) || b->sv_failcnt.ensure(4 * ((size_t)nc + 1)) || b->sv_range.ensure(8 * ((size_t)nc + 1)) || b->sv_vsz.ensure(4 * ((size_t)n_parts + 2)) ||
                    b->sv_voff.ensure(4 * ((size_t)n_parts + 2)))
                    return -1;
                HIPCHK(hipMemsetAsync(b->sv_range.p, 0, 8 * ((size_t)nc + 1), q));
                sv_launch_ranges(q, b->kc_pt_ctg.as<uint32_t>(), n_parts, b->sv_range.as<uint32_t>());
                sv_launch_round1(q, c, nc, b->sv_range.as<uint32_t>(), b->kc_pt_se.as<int32_t>(), b->kc_pt_len.as<uint32_t>(), b->kc_woff.as<uint32_t>(),
                                 n_parts, b->kc_wpool.as<uint8_t>(), b->kc_haswin.as<uint8_t>(), b->sv_failse.as<int32_t>(), b->sv_failcnt.as<uint32_t>());
                sv_launch_val_sizes(q, b->kc_pt_len.as<uint32_t>(), n_parts, b->sv_vsz.as<uint32_t>());
                launch_scan_u32(q, b->sv_vsz.as<uint32_t>(), n_parts, b->sv_voff.as<uint32_t>(), scan_tmp, &totals[5]);
                uint64_t V = 0;
                HIPCHK(npcopy::d2h(&V, &totals[5], 8, q));
                HIPCHK(hipStreamSynchronize(q));
                if (b->sv_val.ensure(4 * (V + 4)) || b->sv_p2ctg.ensure(4 * (V + 4)) || b->sv_p2se.ensure(8 * (V + 4)) || b->sv_p2len.ensure(4 * (V + 4)) ||
                    b->sv_woff2.ensure(4 * (V + 4)) || b->sv_haswin2.ensure(V + 4))
                    return -1;
                HIPCHK(hipMemcpyAsync(b->sv_voff.as<uint32_t>() + n_parts, &totals[5], 4, hipMemcpyDeviceToDevice, q));   // voff[n_parts] = V
                HIPCHK(hipMemsetAsync(b->sv_p2ctg.p, 0xff, 4 * (V + 4), q));
                HIPCHK(hipMemsetAsync(b->sv_p2len.p, 0, 4 * (V + 4), q));
                sv_launch_round2_parts(q, c, nc, b->sv_range.as<uint32_t>(), b->kc_pt_se.as<int32_t>(), n_parts, b->sv_voff.as<uint32_t>(),
                                       b->sv_failse.as<int32_t>(), b->sv_failcnt.as<uint32_t>(), b->sv_val.as<int32_t>(), b->sv_p2ctg.as<uint32_t>(),
                                       b->sv_p2se.as<int32_t>(), b->sv_p2len.as<uint32_t>());
                launch_scan_u32(q, b->sv_p2len.as<uint32_t>(), V, b->sv_woff2.as<uint32_t>(), scan_tmp, &totals[6]);
                uint64_t W2 = 0;
                HIPCHK(npcopy::d2h(&W2, &totals[6], 8, q));
                HIPCHK(hipStreamSynchronize(q));
                if (b->kc_wpool.ensure(W2 + 64)) return -1;     // round 1's winners are in the slots by now
                if (b->replay.on) {
                    HIPCHK(hipMemsetAsync(c.hcount, 0, 4, q));
                    if (V && replay_votes(b, c, b->sv_p2ctg.as<uint32_t>(), b->sv_p2se.as<int32_t>(), b->sv_p2len.as<uint32_t>(), b->sv_woff2.as<uint32_t>(), (uint32_t)V, n,
                                          b->kc_wpool.as<uint8_t>(), b->sv_haswin2.as<uint8_t>(), q) != 0)
                        return -1;
                } else
                kc_launch_winner(q, c, b->sv_p2ctg.as<uint32_t>(), b->sv_p2se.as<int32_t>(), b->sv_p2len.as<uint32_t>(), b->sv_woff2.as<uint32_t>(),
                                 (uint32_t)V, n, b->kc_wpool.as<uint8_t>(), b->sv_haswin2.as<uint8_t>());
                sv_launch_round2_apply(q, c, nc, b->sv_range.as<uint32_t>(), n_parts, b->sv_voff.as<uint32_t>(), b->sv_p2ctg.as<uint32_t>(),
                                       b->sv_p2se.as<int32_t>(), b->sv_p2len.as<uint32_t>(), b->sv_woff2.as<uint32_t>(), b->kc_wpool.as<uint8_t>(),
                                       b->sv_haswin2.as<uint8_t>());
            }
        }
        // ---- emit with mask FLAG_ZERO (kmercount.c:121); snp_valid emits without marks (snpvalid.c:30: flag 0)
        if (snp_valid) HIPCHK(hipMemsetAsync(b->kc_sflag.p, 0, S, q));
        kc_launch_result(q, b->kc_sbase.as<uint8_t>(), b->kc_sflag.as<uint8_t>(), S, b->slot_res.as<uint16_t>());
        launch_scan_keep(q, b->slot_res.as<uint16_t>(), S, b->opos.as<uint32_t>(), scan_tmp, &totals[4]);
        launch_emit(q, b->slot_res.as<uint16_t>(), b->slot_info.as<uint8_t>(), b->opos.as<uint32_t>(), S, 1u, b->out.as<uint8_t>());
        launch_contig_bounds(q, ctg_off, nc, b->soff.as<uint32_t>(), b->opos.as<uint32_t>(), b->bounds.as<uint32_t>());
        b->h_bounds.resize((size_t)nc + 1);
        HIPCHK(npcopy::d2h(b->h_bounds.data(), b->bounds.p, 4 * ((size_t)nc + 1), q));
        HIPCHK(npcopy::d2h(hk, kcnt, sizeof(hk), q));
        HIPCHK(hipStreamSynchronize(q));
        if (hk[KCC_ERR] & ERR_KC_POOL) continue;   // a scratch pool ran out: rerun with larger pools
        if (hk[KCC_ERR] & ERR_KC_UNDEFINED) {
            np1_set_error("snp_valid: a second-round region reaches outside the insertion columns of its k-mer region; the reference dereferences a null list for this input "
                          "(snpvalid.c:24-27, kmercount.c:398,431) and has no defined result");
            return -1;
        }
        if (hk[KCC_ERR]) { np1_set_error(std::string(task) + ": inconsistent pileup or region overflow on the device"); return -1; }
        b->votes = 0;
        b->ran = true;
        return 0;
    }
    np1_set_error(std::string(task) + ": scratch pools keep overflowing");
    return -1;
}

int np1_batch_kmer_count(np1_batch* b, const Configure* cfg, float* stage_ms) {
    (void)stage_ms;
    return kmer_pipeline(b, cfg, false);
}
int np1_batch_snp_valid(np1_batch* b, const Configure* cfg, float* stage_ms) {
    (void)stage_ms;
    return kmer_pipeline(b, cfg, true);
}

int np1_batch_sync(np1_batch* b) {
    if (!b) return -1;
    (void)hipSetDevice(b->ctx->device);
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

int64_t np1_batch_result_len(np1_batch* b, int64_t c) {
    if (!b || !b->ran || c < 0 || c >= (int64_t)b->nc) return -1;
    return (int64_t)b->h_bounds[(size_t)c + 1] - (int64_t)b->h_bounds[(size_t)c];
}

// Polished strings of the whole batch -> pinned host buffer, on the batch's stream (one D2H copy), then waits for it.
int np1_batch_results_fetch(np1_batch* b) {
    if (!b || !b->ran) { np1_set_error("np1_batch_results_fetch: no completed run"); return -1; }
    (void)hipSetDevice(b->ctx->device);
    const size_t total = b->h_bounds[b->nc];
    if (total + 1 > b->h_pin_cap) {
        if (b->h_pin) (void)npalloc::host_free(b->h_pin);
        b->h_pin = nullptr;
        b->h_pin_cap = total + total / 8 + 4096;
        if (!hip_ok(npalloc::host_malloc((void**)&b->h_pin, b->h_pin_cap, hipHostMallocDefault), "hipHostMalloc")) { b->h_pin_cap = 0; return -1; }
    }
    if (total) HIPCHK(npcopy::d2h(b->h_pin, b->out.p, total, b->ctx->stream));
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
    b->out_pinned = true;
    return 0;
}
const char* np1_batch_results_ptr(np1_batch* b) { return (b && b->out_pinned) ? (const char*)b->h_pin : nullptr; }
const uint32_t* np1_batch_results_bounds(np1_batch* b) { return (b && b->ran) ? b->h_bounds.data() : nullptr; }

int np1_batch_result_copy(np1_batch* b, int64_t c, char* dst, int64_t cap) {
    int64_t len = np1_batch_result_len(b, c);
    if (len < 0 || cap < len + 1) { np1_set_error("np1_batch_result_copy: no result or buffer too small"); return -1; }
    (void)hipSetDevice(b->ctx->device);
    if (b->out_pinned) {
        memcpy(dst, b->h_pin + b->h_bounds[(size_t)c], (size_t)len);
        dst[len] = '\0';
        return 0;
    }
    if (!b->out_cached) {
        size_t total = b->h_bounds[b->nc];
        b->h_out.resize(total + 1);
        if (total) HIPCHK(npcopy::d2h_sync(b->h_out.data(), b->out.p, total));
        b->out_cached = true;
    }
    memcpy(dst, b->h_out.data() + b->h_bounds[(size_t)c], (size_t)len);
    dst[len] = '\0';
    return 0;
}

// ---- intra-contig tiling (np1_tile.cpp) -------------------------------------------------------------------------
extern "C" int np1_batch_keep_single(np1_batch* b, int on) {
    if (!b) { np1_set_error("np1_batch_keep_single: null batch"); return -1; }
    b->keep_single = on != 0;
    return 0;
}
// Join facts of a one-contig batch that holds a tile: i_* are indices into the tile's own draft (e_lo, a, b, e_hi minus the hull's start),
// skip = 2 when the halo does not start at the contig's first base (the two slots behind an artificial start have a cut context).
extern "C" int np1_batch_tile_join(np1_batch* b, uint32_t i_elo, uint32_t i_a, uint32_t i_b, uint32_t i_ehi, uint32_t skip, uint32_t out[4]) {
    if (!b || !b->ran || !b->keep_single || b->nc != 1) { np1_set_error("np1_batch_tile_join: needs a completed score_chain run of a one-contig batch with keep_single"); return -1; }
    if (!(i_elo <= i_a && i_a <= i_b && i_b <= i_ehi && i_ehi <= b->G)) { np1_set_error("np1_batch_tile_join: bad tile indices"); return -1; }
    (void)hipSetDevice(b->ctx->device);
    hipStream_t q = b->ctx->stream;
    if (b->join_out.ensure(16)) return -1;
    launch_join_info(q, b->soff.as<uint32_t>(), b->single_map.as<uint8_t>(), b->opos.as<uint32_t>(), i_elo, i_a, i_b, i_ehi, skip, b->join_out.as<uint32_t>());
    HIPCHK(npcopy::d2h(out, b->join_out.p, 16, q));
    HIPCHK(hipStreamSynchronize(q));
    return 0;
}
// polished characters [o0, o1) of the batch's output
extern "C" int np1_batch_result_range(np1_batch* b, uint32_t o0, uint32_t o1, char* dst) {
    if (!b || !b->ran || o0 > o1 || o1 > b->h_bounds[b->nc]) { np1_set_error("np1_batch_result_range: bad range"); return -1; }
    (void)hipSetDevice(b->ctx->device);
    if (o1 > o0) HIPCHK(npcopy::d2h_sync(dst, b->out.as<uint8_t>() + o0, (size_t)(o1 - o0)));
    return 0;
}

int np1_batch_debug_counters(np1_batch* b, uint32_t* out, int n) {
    if (!b) return -1;
    for (int i = 0; i < n && i < (int)CNT_WORDS; ++i) out[i] = b->last_counters[i];
    return 0;
}
// diagnostics: raw per-slot arrays of the last run.  kind 0 = slot_info (u8), 1 = slot_res (u16), 2 = slot_rec (u32);
// returns the slot count, copies min(n, S) elements
int64_t np1_batch_debug_slots(np1_batch* b, int kind, void* out, int64_t n) {
    if (!b || !b->ran) return -1;
    (void)hipSetDevice(b->ctx->device);
    const size_t m = (size_t)std::min<int64_t>(n, (int64_t)b->S);
    const void* src = kind == 0 ? b->slot_info.p : kind == 1 ? b->slot_res.p : b->slot_rec.p;
    const size_t w = kind == 0 ? 1 : kind == 1 ? 2 : 4;
    if (m && npcopy::d2h_sync(out, src, m * w) != hipSuccess) return -1;
    return (int64_t)b->S;
}
int64_t np1_batch_update_count(np1_batch* b) { return b && b->ran ? (int64_t)b->votes : -1; }
int64_t np1_batch_device_bytes(np1_batch* b) { return b ? (int64_t)b->device_bytes() : -1; }

}  // extern "C"

void np1_stream_unpin(np1_stream* st) {
    if (!st) return;
    std::lock_guard<std::mutex> one(st->pin_mu);
    if (!st->pinned.load(std::memory_order_acquire)) return;
    // copies out of the arena may still be in flight on any lane's stream, of any device this process uploads to: every stream of the
    // library's registry is waited for (np_devalloc.h), not just the calling thread's current device
    npalloc::quiesce();
    if (st->arena) (void)npalloc::host_free(st->arena);
    st->arena = nullptr;
    st->arena_bytes = 0;
    st->arena_map.clear();
    st->pinned.store(false, std::memory_order_release);
}

// ---- internal accessors used by the drop-in entry points (np1_abi.cpp) for the -debug trace list
int np1_batch_download_slots(np1_batch* b, int64_t c, std::vector<uint32_t>* soff, std::vector<uint16_t>* res) {
    if (!b || !b->ran || c < 0 || c >= (int64_t)b->nc) return -1;
    (void)hipSetDevice(b->ctx->device);
    uint32_t g0 = b->h_ctg_off[(size_t)c], g1 = b->h_ctg_off[(size_t)c + 1];
    soff->resize((size_t)(g1 - g0) + 1);
    HIPCHK(npcopy::d2h_sync(soff->data(), b->soff.as<uint32_t>() + g0, 4 * soff->size()));
    uint32_t s0 = (*soff)[0], s1 = soff->back();
    res->resize(s1 - s0);
    if (s1 > s0) HIPCHK(npcopy::d2h_sync(res->data(), b->slot_res.as<uint16_t>() + s0, 2 * (size_t)(s1 - s0)));
    return 0;
}

// the same D2H copy straight into a caller's buffer (page-locked for full rate: np1_host_alloc_pinned) -- the streamed pipe keeps the
// results of a run per batch and used to copy them once more out of the lane's own buffer
int np1_batch_results_fetch_to(np1_batch* b, char* dst, size_t cap) {
    if (!b || !b->ran || !dst) { np1_set_error("np1_batch_results_fetch_to: no completed run"); return -1; }
    (void)hipSetDevice(b->ctx->device);
    const size_t total = b->h_bounds[b->nc];
    if (total > cap) { np1_set_error("np1_batch_results_fetch_to: buffer too small"); return -1; }
    if (total) HIPCHK(npcopy::d2h(dst, b->out.p, total, b->ctx->stream));
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
    return 0;
}
size_t np1_batch_results_total(np1_batch* b) { return (b && b->ran) ? (size_t)b->h_bounds[b->nc] : 0; }
void* np1_host_alloc_pinned(size_t bytes) {
    void* p = nullptr;
    if (npalloc::host_malloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}
void np1_host_free_pinned(void* p) { if (p) (void)npalloc::host_free(p); }
void np1_batch_swap_work(np1_batch* a, np1_batch* b) { if (a && b && a != b) a->swap_work(*b); }

// Makes np1_batch_kmer_count / np1_batch_snp_valid of this batch replay the reference's region iterator (np1_replay.h).  `st` is the stream
// the batch was uploaded from, read from `bam` (it carries the records' virtual offsets) and has to stay alive until the pass is done.
extern "C" int np1_batch_enable_replay(np1_batch* b, const np1_stream* st, const char* bam) {
    if (!b || !st || !bam) { np1_set_error("np1_batch_enable_replay: null argument"); return -1; }
    const np::ReadStream& s = st->s;
    if (s.voff.size() != s.n_reads() || s.voff_end.size() != s.n_reads() || (int64_t)s.n_reads() != b->n_reads || s.n_contigs() != b->nc) {
        np1_set_error("np1_batch_enable_replay: the stream was not read from a file, or is not the one this batch holds");
        return -1;
    }
    np1_batch::Replay& R = b->replay;
    R.on = false;
    // the parsed index is kept across passes of one batch object, keyed on what the file IS, not only on its name: an index regenerated
    // at the same path between two passes (iterative polishing rounds, tests reusing temporary names) is read again
    const std::string bai_path = std::string(bam) + ".bai";
    struct stat sb;
    if (stat(bai_path.c_str(), &sb) != 0) { np1_set_error("cannot load BAM index: " + bai_path); return -1; }
    const std::string key = bai_path + "|" + std::to_string((long long)sb.st_ino) + "|" + std::to_string((long long)sb.st_size) + "|" +
                            std::to_string((long long)sb.st_mtim.tv_sec) + "." + std::to_string((long long)sb.st_mtim.tv_nsec);
    if (R.own_bai_path != key) {
        R.own_bai_path.clear();
        R.own_bai = np::BaiIndex();
        if (!R.own_bai.load(bai_path)) { np1_set_error("cannot load BAM index: " + bai_path); return -1; }
        R.own_bai_path = key;
    }
    R.bai = &R.own_bai;
    np::BamReader rd;
    if (!rd.open(bam)) { np1_set_error(std::string("cannot open BAM: ") + bam); return -1; }
    R.tid.assign(s.n_contigs(), -1);
    for (size_t c = 0; c < s.n_contigs(); ++c) R.tid[c] = rd.header().name2id(s.names[c]);
    R.voff = s.voff.data(); R.voff_end = s.voff_end.data();
    b->h_ctg_off.assign(s.ctg_off.begin(), s.ctg_off.end());
    b->h_read_begin.assign(s.read_begin.begin(), s.read_begin.end());
    R.on = true;
    return 0;
}

// The same for a batch the device-side ingest filled (np1_ingest.hip): the records' virtual offsets are in R.own_voff / own_voff_end
// already; `bai` (the pipe's index) has to stay alive until the pass is done.
int np1_batch_enable_replay_ingested(np1_batch* b, const np::BaiIndex* bai, const std::vector<int32_t>& tid) {
    np1_batch::Replay& R = b->replay;
    R.on = false;
    if (!bai || tid.size() != b->nc || R.own_voff.cap < 8 * (size_t)b->n_reads || R.own_voff_end.cap < 8 * (size_t)b->n_reads) {
        np1_set_error("np1_batch_enable_replay_ingested: the batch holds no virtual offsets");
        return -1;
    }
    R.bai = bai;
    R.tid = tid;
    R.voff = R.own_voff.as<uint64_t>(); R.voff_end = R.own_voff_end.as<uint64_t>();
    R.on = true;
    return 0;
}
