// HIP window executor of the long-read path (gfx950): alignment spans, tag streams + column statistics, link
// observations bucketed per draft column, per-column link graph, chain DP and backtrace.  Per-lane bodies are the
// ones in np2_core.h (also run by the tests' host model).  One process per GPU; device = NP2_DEVICE or pid mod n.
//
// Launch sequence per window (reference stages in brackets):
//   k2_span      lane / candidate record   clip to the window + first/last run of 8 matches   [bam2aln, clip_aln, get_align_shift]
//   (host)       500 bp rule and coverage caps over the spans (order dependent, O(records))  [ctg_cns.c:3540-3545]
//   k2_seed_tags lane / tag byte           the window against itself                          [get_align_tags on the seed]
//   k2_tags      lane / kept record        4-bit tag stream + column statistics (atomics)     [get_align_tags]
//   k2_links<0>  lane / stream             link observations per column: count               [update_msa]
//   scan         exclusive scan of the counts -> column buckets
//   k2_links<1>  lane / stream             scatter the observations
//   k2_build     lane / column             order the bucket by (stream, delta), first-seen entry lists with link counts
//   k2_cut_flags/scan/k2_cut_list          single-entry columns = cuts between independent runs
//   k2_run_ac    wave / run (first lane)   (a, c) of every entry, (A, C) of the run: x -> max(C, x + A)        [get_cns_from_align_tags]
//   k2_run_scan  one lane                  score at every cut
//   k2_run_dp    wave / run (first lane)   the reference's literal DP on the run's columns with its true entry score
//   k2_bt_runs   wave / run (first lane)   best path per run: count, scan, write               [generate_cns_from_best_score]
// Bounds: tags/links/build stream every alignment column once (HBM); the run kernels are latency bound per run and
// parallel across the thousands of runs of a window.
#include <hip/hip_runtime.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/nextpolish2.h"
#include "np2_exec.h"
#include "np2_ond_dev.h"
#include "np2_poa_dev.h"
#include "np2_lq.h"
#include "np_threads.h"
#include "np_devalloc.h"
#include "np_hostcopy.h"

namespace np { void bgzf_device_inflate_enable(int device); }   // np_bgzf_dev.hip

namespace np2 {
namespace {

using namespace np2k;

#define HIPOK(x)                                                                                        \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        /* (nothing of this process may still be moving when the caller's locals go away: a second stream may be running) */ \
        if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); (void)hipDeviceSynchronize(); return false; }  \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap && p) return true;
        if (p) { (void)npalloc::dev_free(p); p = nullptr; cap = 0; }
        const size_t want = npalloc::efence() ? bytes : bytes + bytes / 4 + 256;      // NP_EFENCE=1: the buffer ends where its mapping ends (np_devalloc.h)
        if (npalloc::dev_malloc(&p, want) != hipSuccess) { p = nullptr; return false; }
        cap = want;
        return true;
    }
    ~DevBuf() { if (p) (void)npalloc::dev_free(p); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct PinBuf {   // pinned host staging, grow-only
    void* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap && p) return true;
        if (p) { (void)npalloc::host_free(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 256;
        if (npalloc::host_malloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return false; }
        cap = want;
        return true;
    }
    ~PinBuf() { if (p) (void)npalloc::host_free(p); }
};

struct DevStat {   // column statistics as 32-bit device counters (packed to the reference's 16-bit fields afterwards)
    uint32_t *coverage, *max_size, *l_ins, *l_del;
    __device__ void add_coverage(uint32_t p) const { atomicAdd(&coverage[p], 1u); }
};
struct DevStatSink {
    DevStat d;
    __device__ void coverage(uint32_t p) { atomicAdd(&d.coverage[p], 1u); }
    __device__ void max_size(uint32_t p, uint32_t delta) { atomicMax(&d.max_size[p], delta + 1); }
    __device__ void l_ins(uint32_t p) { atomicAdd(&d.l_ins[p], 1u); }
    __device__ void l_del(uint32_t p) { atomicAdd(&d.l_del[p], 1u); }
};

struct DevRecs {   // one RecordSet in HBM
    const int32_t* pos;
    const uint32_t *n_cigar, *q0;
    const uint64_t *cigar_off, *seq_off;
    const uint32_t* cigar;
    const uint8_t* seq;
    __device__ ReadView view(uint32_t i) const { return ReadView{pos[i], n_cigar[i], cigar + cigar_off[i], seq + seq_off[i]}; }
};

// (lane-per-record kernels with long serial walks launch SERIAL_LANES records per wave: a window has ~10^4 records, 64 per
// wave would leave most SIMDs idle and every wave as slow as its longest CIGAR)
constexpr uint32_t SERIAL_LANES = 8;
__global__ void k2_span(DevRecs R, uint32_t n, const char* contig_seq, int32_t s, int32_t e, SpanOut* out) {
    const uint32_t i = blockIdx.x * SERIAL_LANES + threadIdx.x;
    if (threadIdx.x >= SERIAL_LANES || i >= n) return;
    const ReadView rv = R.view(i);
    uint32_t N, rf_len, rd_len;
    bool bad;
    cigar_totals(rv, &N, &rf_len, &rd_len, &bad);
    SpanOut o{0, 0, 0, 0, 0, bad ? 1u : 0u};
    if (!bad) {
        const AlnSpan a = align_span(rv, contig_seq, s, e, R.q0[i]);
        o.col0 = a.col0; o.aln_len = a.aln_len; o.aln_t_s = a.aln_t_s; o.aln_t_e = a.aln_t_e; o.aln_q_s = a.aln_q_s;
    }
    out[i] = o;
}

// the seed stream: window against itself, two columns per byte (its coverage +1 and max_size 1 on every column are
// added by k2_pack_stat)
__global__ void k2_seed_tags(const char* contig_seq, int32_t s, uint32_t l, uint8_t* tags) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;   // tag byte
    const uint32_t nbytes = (l + 1) / 2 + 1;
    if (b >= nbytes) return;
    uint32_t v = 0;
    for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t p = 2 * b + h;
        uint32_t nib;
        if (p < l) {
            nib = base_to_int((unsigned char)contig_seq[s + (int32_t)p]);
        } else {
            nib = 15;   // terminator (the reference sets the trailing nibble(s) of the last byte(s) to 15)
        }
        v |= h ? nib : nib << 4;
    }
    tags[b] = (uint8_t)v;
}

struct StreamDesc {   // one stream besides the seed
    uint32_t set, read;   // record set (0 window records, 1 supplementary alignments) and index into it
    uint32_t col0, aln_len, aln_t_s, pad;   // aln_t_s: contig coordinate
    uint64_t tag_off;
};

// ---- chunk-parallel tag emission -------------------------------------------------------------------------------
// k2_tag_ckpt (lane per kept record, O(#CIGAR ops)): the CIGAR cursor and the emission state at every
// TAG_CHUNK-th kept column; k2_tags_chunk (lane per chunk): get_align_tags on its columns.  TAG_CHUNK is even, so
// a chunk owns whole tag bytes.
constexpr uint32_t TAG_CHUNK = 256;
struct TagCkpt { uint32_t op_i, in_op, rfi, rdi, te_before, delta_before; };
struct TagChunk { uint32_t stream, c0, n, last; };   // stream = index into the StreamDesc array

__global__ void k2_tag_ckpt(const StreamDesc* sd, uint32_t n_streams, const uint32_t* chunk_off, DevRecs R0, DevRecs R1, int32_t s, TagCkpt* ck) {
    const uint32_t k = blockIdx.x * SERIAL_LANES + threadIdx.x;
    if (threadIdx.x >= SERIAL_LANES || k >= n_streams) return;
    const StreamDesc d = sd[k];
    const DevRecs& R = d.set ? R1 : R0;
    const uint32_t* cg = R.cigar + R.cigar_off[d.read];
    const uint32_t nc = R.n_cigar[d.read];
    const uint32_t n_chunks = (d.aln_len + TAG_CHUNK - 1) / TAG_CHUNK;
    TagCkpt* out = ck + chunk_off[k];
    uint32_t col = 0, rfi = (uint32_t)R.pos[d.read], rdi = 0, tcount = 0, run = 0, ci = 0;
    uint32_t target = d.col0;
    const uint32_t te0 = d.aln_t_s - (uint32_t)s - 1;
    for (uint32_t i = 0; i < nc && ci < n_chunks; ++i) {
        const uint32_t c = cg[i] & 0xf, n = cg[i] >> 4;
        if (c == 4 || c == 5) { rdi += n; continue; }
        if (c == 3) { rfi += n; continue; }
        if (c > 2 || n == 0) continue;
        while (ci < n_chunks && target < col + n) {
            const uint32_t o = target - col;
            TagCkpt t;
            t.op_i = i; t.in_op = o;
            t.rfi = rfi + (c != 1 ? o : 0u);
            t.rdi = rdi + (c != 2 ? o : 0u);
            const uint32_t from = col > d.col0 ? col : d.col0;   // counting starts at the first kept column
            t.te_before = te0 + tcount + (c != 1 ? target - from : 0u);
            t.delta_before = ci == 0 ? 0u : (o > 0 ? (c == 1 ? run + o : 0u) : run);
            out[ci++] = t;
            target += TAG_CHUNK;
        }
        if (c != 1 && col + n > d.col0) tcount += col + n - (col > d.col0 ? col : d.col0);
        if (c != 1) rfi += n;
        if (c != 2) rdi += n;
        run = c == 1 ? run + n : 0u;
        col += n;
    }
}

// get_align_tags on one chunk of a record's kept columns, by CIGAR op (np2_core.h emit_tags_range is the per-column
// statement of the same result).  Eight tags are packed per 32-bit store (a chunk owns whole words of the zeroed tag
// area).  Column statistics: coverage is NOT counted here -- a stream covers a contiguous range of positions, so it is
// +1/-1 into a difference array (k2_cov_diff) and a scan; max_size only moves on insertion columns (the seed already
// holds 1 everywhere); l_ins / l_del are counted where they happen.  The one exception to "coverage = streams over the
// position" is a read base with the IUPAC code M, which the reference's marker test skips (ctg_cns.c:1232): fixed up
// in the difference array; such a base also drops two link observations, so *m_seen sends the link count back to
// its exact pass (k2_chunk_links<false>) instead of tags-per-position = coverage + insertion tags.
__global__ void k2_tags_chunk(const TagChunk* tc, uint32_t n_chunks, const TagCkpt* ck, const StreamDesc* sd, DevRecs R0, DevRecs R1,
                              uint32_t gap_min_len, uint8_t* tags, DevStat st, uint32_t* cov_diff, uint32_t* m_seen, uint32_t* te_out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const TagChunk ch = tc[c];
    const StreamDesc d = sd[ch.stream];
    const TagCkpt t = ck[c];
    const ReadView rv = (d.set ? R1 : R0).view(d.read);
    // base_to_int(nt16_char(code)) for the 16 BAM base codes "=ACMGRSVTWYHKDBN": A0 T1 G2 C3 N5 M6, everything else 4
    constexpr uint64_t LUT = 0x5444444144426304ull;
    uint32_t op_i = t.op_i, in_op = t.in_op, rdi = t.rdi;
    uint32_t te = t.te_before, delta = t.delta_before, l = t.delta_before >= gap_min_len ? 1u : 0u;
    uint32_t* tw = reinterpret_cast<uint32_t*>(tags + d.tag_off) + (ch.c0 >> 3);
    uint32_t word = 0, nn = 0, left = ch.n;
    auto put = [&](uint32_t b) {
        word |= b << (((nn >> 1) << 3) + ((nn & 1u) ? 0u : 4u));
        if (++nn == 8) { *tw++ = word; word = 0; nn = 0; }
    };
    while (left) {
        const uint32_t cg = rv.cigar[op_i];
        const uint32_t op = cg & 0xfu, n = cg >> 4;
        if (op > 2 || n == 0) {   // no columns: clips move the query cursor
            if (op == 4 || op == 5) rdi += n;
            ++op_i;
            in_op = 0;
            continue;
        }
        const uint32_t take = n - in_op < left ? n - in_op : left;
        if (op == 0) {
            for (uint32_t k = 0; k < take; ++k, ++rdi) {
                const uint32_t code = (uint32_t)(rv.seq[rdi >> 1] >> ((~rdi & 1u) << 2)) & 15u;
                const uint32_t b = (uint32_t)(LUT >> (code << 2)) & 15u;
                ++te;
                if (b == 6) { atomicSub(&cov_diff[te], 1u); atomicAdd(&cov_diff[te + 1], 1u); *m_seen = 1u; }
                put(b);
            }
            l = 0;
            delta = 0;
        } else if (op == 1) {
            atomicAdd(&st.coverage[te], take);   // (windows do not count coverage here: the array holds the insertion tags per position)
            for (uint32_t k = 0; k < take; ++k, ++rdi) {
                const uint32_t code = (uint32_t)(rv.seq[rdi >> 1] >> ((~rdi & 1u) << 2)) & 15u;
                if (code == 3u) *m_seen = 1u;
                ++delta;
                atomicMax(&st.max_size[te], delta + 1);
                if (delta >= gap_min_len && !l) { atomicAdd(&st.l_ins[te], 1u); l = 1; }
                put(((uint32_t)(LUT >> (code << 2)) & 15u) | 8u);
            }
        } else {
            for (uint32_t k = 0; k < take; ++k) {
                ++te;
                atomicAdd(&st.l_del[te], 1u);
                put(4u);
            }
            l = 0;
            delta = 0;
        }
        in_op += take;
        left -= take;
        if (in_op >= n) { in_op = 0; ++op_i; }
    }
    if (ch.last) {
        put(15u);
        if (!((ch.c0 + ch.n) & 1u)) put(15u);   // the reference fills the whole next byte when the stream ends on a byte boundary
        te_out[ch.stream] = te + 1;
    }
    if (nn) *tw = word;
}
// coverage of the reads as a difference array: +1 where a stream starts, -1 behind its last position
__global__ void k2_cov_diff(const StreamDesc* sd, uint32_t n_streams, int32_t s, const uint32_t* te, uint32_t* cov_diff) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_streams) return;
    const uint32_t ts = sd[k].aln_t_s - (uint32_t)s;
    if (te[k] > ts) { atomicAdd(&cov_diff[ts], 1u); atomicSub(&cov_diff[te[k]], 1u); }
}

// ---- link observations from the tag streams, chunk-parallel ----------------------------------------------------
// A stream is cut into chunks of LINK_CHUNK tags; the position of a tag is aln_t_s + (# non-insertion tags up to it)
// - 1, so a per-chunk popcount + scan gives every chunk its starting position, and the two previous tags (pp, ppp)
// and the open insertion run are recovered by a short look-back.  Lanes = chunks (tens of thousands per window)
// instead of streams (hundreds, one of them -- the seed -- as long as the window).
constexpr uint32_t LINK_CHUNK = 512;
struct ChunkDesc { uint32_t stream, first_tag, n_tags, first_chunk_of_stream; };

__device__ __forceinline__ uint32_t tag_nib(const uint8_t* tg, uint32_t i) {
    const uint32_t t = tg[i >> 1];
    return (i & 1) ? (t & 15u) : (t >> 4);
}
// A chunk is 512 tags = 64 dwords: one wave per chunk, one coalesced dword (8 tags) per lane.  Tag t of a dword sits in byte t / 2,
// high nibble first; its insertion bit is bit 8 * (t / 2) + (t even ? 7 : 3).
__device__ __forceinline__ uint32_t dword_ins_mask(uint32_t v) {          // insertion bits of the first v (<= 8) tags of a dword
    uint32_t m = 0;
#pragma unroll
    for (uint32_t t = 0; t < 8; ++t) m |= t < v ? 1u << (8u * (t >> 1) + ((t & 1u) ? 3u : 7u)) : 0u;
    return m;
}
__device__ __forceinline__ uint32_t dword_nib(uint32_t w, uint32_t t) { return (w >> (8u * (t >> 1) + ((t & 1u) ? 0u : 4u))) & 15u; }
__global__ __launch_bounds__(256) void k2_chunk_count(const ChunkDesc* cd, uint32_t n_chunks, const uint64_t* tag_off, const uint8_t* tags, uint32_t* cnt) {
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= n_chunks) return;
    const ChunkDesc d = cd[c];
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(tags + tag_off[d.stream] + (d.first_tag >> 1));   // streams start on 4-byte boundaries
    const uint32_t v = 8u * lane < d.n_tags ? min(8u, d.n_tags - 8u * lane) : 0u;
    const uint32_t w = v ? tw[lane] : 0u;
    uint32_t n = v - (uint32_t)__popc(w & dword_ins_mask(v));
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0) cnt[c] = n;
}
// position (t_pos, delta) of tag j of a stream given T = number of non-insertion tags among tags 0..j
__device__ __forceinline__ void tag_pos(const uint8_t* tg, uint32_t j, uint32_t T, uint32_t aln_t_s, int32_t* t_pos, uint32_t* delta) {
    *t_pos = (int32_t)(aln_t_s + T) - 1;
    uint32_t d = 0;
    while ((tag_nib(tg, j - d) & 8u)) ++d;   // a stream never starts with an insertion column, so this stops at or before tag 0
    *delta = d & 0xffffu;
}
// link observation as the device keeps it: pp, ppp, (stream | delta << 32 | base << 48), column -- 32 bytes, aligned
struct __attribute__((aligned(32))) DevObs { uint64_t pp, ppp, meta, col; };

template <bool kFill>
__global__ void k2_chunk_links(const ChunkDesc* cd, uint32_t n_chunks, const uint32_t* pre, const uint64_t* tag_off, const uint32_t* aln_t_s,
                               const uint8_t* tags, uint32_t* col_cnt, const uint32_t* col_off, uint32_t* cursor, DevObs* obs) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const ChunkDesc d = cd[c];
    const uint8_t* tg = tags + tag_off[d.stream];
    const uint32_t ts = aln_t_s[d.stream];
    uint32_t T = pre[c] - pre[d.first_chunk_of_stream];   // non-insertion tags before this chunk
    uint64_t pp = KEY_HEAD, ppp = KEY_HEAD;
    uint32_t pp_base = 0;
    int32_t t_pos = (int32_t)(ts + T) - 1;
    uint32_t delta = 0;
    if (d.first_tag >= 1) {
        const uint32_t j = d.first_tag - 1;
        int32_t tp;
        uint32_t dl;
        tag_pos(tg, j, T, ts, &tp, &dl);
        const uint32_t nb = tag_nib(tg, j);
        pp = node_key(tp, dl, nb & 7u);
        pp_base = nb & 7u;
        delta = dl;   // the open insertion run continues into this chunk
        if (d.first_tag >= 2) {
            const uint32_t T2 = T - ((nb & 8u) ? 0u : 1u);
            tag_pos(tg, j - 1, T2, ts, &tp, &dl);
            ppp = node_key(tp, dl, tag_nib(tg, j - 1) & 7u);
        }
    }
    for (uint32_t i = 0; i < d.n_tags; ++i) {
        const uint32_t nb = tag_nib(tg, d.first_tag + i);
        const uint32_t base = nb & 7u;
        if (nb & 8u) delta = (delta + 1) & 0xffffu;
        else { delta = 0; ++t_pos; }
        const uint64_t key = node_key(t_pos, delta, base);
        if (base != 6 && pp_base != 6) {
            if (!kFill) {
                atomicAdd(&col_cnt[t_pos], 1u);
            } else {
                const uint32_t at = col_off[t_pos] + atomicAdd(&cursor[t_pos], 1u);
                // one aligned 32-byte record per observation, written as two 16-byte stores: a 24-byte record + a separate
                // column word cost four times their size in HBM writes (partial lines) and as much again in fetches
                uint4* dst = reinterpret_cast<uint4*>(obs + at);
                const uint64_t meta = (uint64_t)d.stream | (uint64_t)(delta & 0xffffu) << 32 | (uint64_t)(base & 0xffu) << 48;
                dst[0] = make_uint4((uint32_t)pp, (uint32_t)(pp >> 32), (uint32_t)ppp, (uint32_t)(ppp >> 32));
                dst[1] = make_uint4((uint32_t)meta, (uint32_t)(meta >> 32), (uint32_t)t_pos, 0u);
            }
        }
        ppp = pp; pp = key; pp_base = base;
    }
}

// Nodes and entries of every column from its link observations, one lane per OBSERVATION (np2_core.h build_column is
// the per-column statement of the same result).  Within a node (t_pos, delta, base) every stream appears at most once,
// so "first seen" order is stream order and nothing has to be sorted: an observation is the first of its node / of its
// (pp, ppp) pair when no observation of a smaller stream shares it, an entry's slot is the number of pair-firsts before
// it, a node's region starts after the observations of all smaller nodes, a node's index is the number of distinct
// smaller nodes.  Pass A leaves (observations in smaller nodes, pair multiplicity, the two "first" flags) per
// observation, pass B turns the flags of the column into slots and writes.  A column's bucket is a few hundred bytes and
// neighbouring lanes work on the same bucket, so the loops read through L1.
__device__ __forceinline__ uint32_t obs_nkey(uint64_t meta) { return (uint32_t)((meta >> 32) & 0xffffu) << 8 | (uint32_t)((meta >> 48) & 0xffu); }
constexpr uint64_t AUX_NODE_FIRST = 1ull << 48, AUX_PAIR_FIRST = 1ull << 49;
__global__ __launch_bounds__(256) void k2_build_a(const DevObs* __restrict__ obs, const uint32_t* __restrict__ col_off,
                                                  uint32_t total, uint64_t* __restrict__ aux) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t p = (uint32_t)obs[i].col;
    const uint32_t lo = col_off[p], hi = col_off[p + 1];
    const uint64_t* w = reinterpret_cast<const uint64_t*>(obs);
    const uint64_t pp = w[4ull * i], ppp = w[4ull * i + 1], meta = w[4ull * i + 2];
    const uint32_t nk = obs_nkey(meta), rd = (uint32_t)meta;
    uint32_t less = 0, pairs = 0;
    bool node_first = true, pair_first = true;
    for (uint32_t j = lo; j < hi; ++j) {
        const uint64_t mj = w[4ull * j + 2];
        const uint32_t nkj = obs_nkey(mj);
        less += nkj < nk ? 1u : 0u;
        if (nkj == nk) {
            const bool before = (uint32_t)mj < rd;
            node_first = node_first && !before;
            if (w[4ull * j] == pp && w[4ull * j + 1] == ppp) { ++pairs; pair_first = pair_first && !before; }
        }
    }
    aux[i] = (uint64_t)less | (uint64_t)pairs << 24 | (node_first ? AUX_NODE_FIRST : 0ull) | (pair_first ? AUX_PAIR_FIRST : 0ull);
}
__global__ __launch_bounds__(256) void k2_build_b(const DevObs* __restrict__ obs, const uint32_t* __restrict__ col_off,
                                                  uint32_t total, const uint64_t* __restrict__ aux, Entry* __restrict__ entries, Node* __restrict__ nodes,
                                                  uint32_t* __restrict__ col_nn, uint8_t* __restrict__ live) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t a = aux[i];
    if (!(a & AUX_PAIR_FIRST)) return;
    const uint32_t p = (uint32_t)obs[i].col;
    const uint32_t lo = col_off[p], hi = col_off[p + 1];
    const uint64_t* w = reinterpret_cast<const uint64_t*>(obs);
    const uint64_t meta = w[4ull * i + 2];
    const uint32_t nk = obs_nkey(meta), rd = (uint32_t)meta;
    const bool node_first = (a & AUX_NODE_FIRST) != 0;
    uint32_t slot = 0, node_len = 0, node_idx = 0, n_nodes = 0;
    for (uint32_t j = lo; j < hi; ++j) {
        const uint64_t aj = aux[j];
        if (!(aj & (AUX_NODE_FIRST | AUX_PAIR_FIRST))) continue;
        const uint64_t mj = w[4ull * j + 2];
        const uint32_t nkj = obs_nkey(mj);
        if (aj & AUX_NODE_FIRST) { ++n_nodes; node_idx += nkj < nk ? 1u : 0u; }
        if ((aj & AUX_PAIR_FIRST) && nkj == nk) { ++node_len; slot += (uint32_t)mj < rd ? 1u : 0u; }
    }
    const uint32_t start = (uint32_t)a & 0xffffffu;
    Entry e;
    e.pp = w[4ull * i];
    e.ppp = w[4ull * i + 1];
    e.score = 0;
    e.link = (uint32_t)(a >> 24) & 0xffffu;   // the reference counts in 16 bits
    e.node = nk;
    entries[lo + start + slot] = e;
    live[lo + start + slot] = 1;
    if (node_first) {
        nodes[lo + node_idx] = Node{nk, start, node_len, 0u};
        if (node_idx == 0) col_nn[p] = n_nodes;
    }
}

// ---- link graph straight from the tag streams, one wave per tile of 64 draft positions ------------------------------------
// The scatter path above writes one 32-byte observation per tag into its column's bucket and lets every observation scan its
// bucket twice (k2_chunk_links, k2_build_a/b: ~123 M observations, ~4 GB out and back several times per 5 Mb window, and the
// entry / node arrays stay as sparse as the buckets).  Here a wave owns a tile of 64 columns.  For every stream that crosses
// the tile (list built by k2_tile_list: stream + index of its first tag inside the tile; taken in ascending stream order =
// first-seen order) the LANES ARE THE STREAM'S TAGS: one coalesced nibble fetch, the position of a tag from a ballot + popcount
// (position = number of non-insertion tags so far), its insertion depth from the distance to the last non-insertion lane, its
// predecessors pp / ppp from the two lanes below.  Every tag is looked up in its column's chain of entries in LDS (own node +
// both predecessors packed into 62 bits: positions relative to the column) and either raises the link count or is appended;
// streams are taken one after the other, so a node's entries come out in first-seen order without sorting observations.
// A finished tile orders each column's few entries by node, takes its place in the COMPACT entry / node arrays with one
// atomic, and writes them.  A tile with more than TG_POOL distinct entries or a column with more than TG_COLMAX (deep
// pileups, thousand-base insertions) raises a flag and the window takes the scatter path instead.
constexpr uint32_t TG_COLMAX = 96, TG_NONE = 0xffffu;
// LDS of a tile: POOL entries, MAXS streams in the sorted list.  Two sizes are built: the work is one chain of dependent LDS
// operations per wave, so what decides the speed is how many waves fit a CU next to each other, and a 20x window needs a third of
// what a 60x window needs.
template <uint32_t POOL, uint32_t MAXS>
struct TileLds {
    unsigned long long key[POOL];
    uint32_t link[POOL];                    // bit 31 (set while the tile is finished): first entry of its node
    uint16_t next[POOL];
    uint8_t col[POOL];                      // column of the entry inside the tile
    uint32_t head[64], tail[64], cnt[64], nn[64];
    uint32_t used;
    uint32_t sid[MAXS], sstart[MAXS], tsid[MAXS], tstart[MAXS];
    uint64_t stoff[MAXS];                   // per sorted stream: tag offset, start position, number of tags
    uint32_t sts[MAXS], snt[MAXS];
};
struct TileStream { uint32_t stream, start; };

// the tiles the non-insertion tags of a stream open (position % 64 == 0, or the stream's first tag): counted with a lane per chunk
// from the chunk's position range, listed with a wave per chunk (lane = 8 tags, positions from a prefix sum over the lanes)
__global__ void k2_tile_count(const ChunkDesc* cd, uint32_t n_chunks, const uint32_t* pre, const uint32_t* aln_t_s, uint32_t n_tiles, uint32_t* tile_cnt) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const ChunkDesc d = cd[c];
    const uint32_t ts = aln_t_s[d.stream];
    const uint32_t a = ts + (pre[c] - pre[d.first_chunk_of_stream]), n = pre[c + 1] - pre[c];   // positions [a, a + n)
    if (!n) return;
    if (d.first_tag == 0 && (ts & 63u) && (ts >> 6) < n_tiles) atomicAdd(&tile_cnt[ts >> 6], 1u);
    for (uint32_t t = (a + 63u) >> 6; t <= (a + n - 1) >> 6 && t < n_tiles; ++t) atomicAdd(&tile_cnt[t], 1u);
}
__global__ __launch_bounds__(256) void k2_tile_list(const ChunkDesc* cd, uint32_t n_chunks, const uint32_t* pre, const uint64_t* tag_off, const uint32_t* aln_t_s,
                                                    const uint8_t* tags, uint32_t n_tiles, const uint32_t* tile_off, uint32_t* cursor, TileStream* list) {
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= n_chunks) return;
    const ChunkDesc d = cd[c];
    const uint32_t a = aln_t_s[d.stream] + (pre[c] - pre[d.first_chunk_of_stream]);
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(tags + tag_off[d.stream] + (d.first_tag >> 1));
    const uint32_t v = 8u * lane < d.n_tags ? min(8u, d.n_tags - 8u * lane) : 0u;
    const uint32_t w = v ? tw[lane] : 0u;
    const uint32_t n = v - (uint32_t)__popc(w & dword_ins_mask(v));
    uint32_t inc = n;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += y; }
    uint32_t p = a + inc - n;
    for (uint32_t t = 0; t < v; ++t) {
        if (dword_nib(w, t) & 8u) continue;
        const uint32_t i = d.first_tag + 8u * lane + t;
        if ((!(p & 63u) || i == 0) && (p >> 6) < n_tiles) list[tile_off[p >> 6] + atomicAdd(&cursor[p >> 6], 1u)] = TileStream{d.stream, i};
        ++p;
    }
}

struct TileArgs {
    const uint8_t* tags; const uint64_t* tag_off; const uint32_t* aln_t_s; const uint32_t* n_tags;
    const uint32_t* tile_off; const TileStream* list; uint32_t n_tiles, n_cols;
    Entry* entries; Node* nodes; uint32_t* col_off; uint32_t* col_nn; uint32_t* col_ne; uint32_t* counter;   // counter[0] entries placed, [1] overflow flag, [2] tiles in redo[]
    uint32_t* redo;            // tiles that did not fit this launch's pool (nullptr: flag the window instead)
    const uint32_t* todo;      // nullptr: tile = block; else tile = todo[block] for block < counter[2]
};

// own node + predecessors of an entry in 62 bits; predecessor positions relative to the column (pp: 0..1 back, ppp: 0..2 back)
__device__ __forceinline__ unsigned long long tg_pack(uint32_t delta, uint32_t base, int32_t tp, unsigned long long pp, unsigned long long ppp) {
    unsigned long long k = (unsigned long long)(delta & 0xffffu) | (unsigned long long)(base & 7u) << 16;
    if (key_tpos(pp) == -1) k |= 1ull << 19;
    else k |= (unsigned long long)((uint32_t)(tp - key_tpos(pp)) & 1u) << 20 | (unsigned long long)key_delta(pp) << 21 | (unsigned long long)(key_base(pp) & 7u) << 37;
    if (key_tpos(ppp) == -1) k |= 1ull << 40;
    else k |= (unsigned long long)((uint32_t)(tp - key_tpos(ppp)) & 3u) << 41 | (unsigned long long)key_delta(ppp) << 43 | (unsigned long long)(key_base(ppp) & 7u) << 59;
    return k;
}
__device__ __forceinline__ uint32_t tg_node(unsigned long long k) { return (uint32_t)(k & 0xffffu) << 8 | (uint32_t)((k >> 16) & 7u); }
__device__ __forceinline__ void tg_unpack(unsigned long long k, int32_t tp, uint64_t* pp, uint64_t* ppp) {
    *pp = ((k >> 19) & 1ull) ? KEY_HEAD : node_key(tp - (int32_t)((k >> 20) & 1ull), (uint32_t)((k >> 21) & 0xffffu), (uint32_t)((k >> 37) & 7u));
    *ppp = ((k >> 40) & 1ull) ? KEY_HEAD : node_key(tp - (int32_t)((k >> 41) & 3ull), (uint32_t)((k >> 43) & 0xffffu), (uint32_t)((k >> 59) & 7u));
}
// lanes of one wave hand LDS data to each other: orders the LDS operations only (a fence over all address spaces would also wait
// for the global loads that are in flight on purpose)
__device__ __forceinline__ void tg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

template <bool kProf, uint32_t TG_POOL, uint32_t TG_MAXS>
__global__ __launch_bounds__(64) void k2_tile_graph(TileArgs A, unsigned long long* prof) {
    long long tk = kProf ? clock64() : 0;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
#define TG_TICK(i) do { if (kProf) { const long long now = clock64(); ph[i] += (unsigned long long)(now - tk); tk = now; } } while (0)
    __shared__ TileLds<TG_POOL, TG_MAXS> L;
    const uint32_t lane = threadIdx.x;
    if (A.todo && blockIdx.x >= A.counter[2]) return;
    const uint32_t t = A.todo ? A.todo[blockIdx.x] : blockIdx.x;
    const uint32_t c0 = t << 6, c1 = c0 + 63;
    const unsigned long long lane_le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    L.head[lane] = TG_NONE; L.tail[lane] = TG_NONE; L.cnt[lane] = 0; L.nn[lane] = 0;
    if (lane == 0) L.used = 0;
    // ---- the tile's streams in ascending order
    const uint32_t l0 = A.tile_off[t], ns = A.tile_off[t + 1] - l0;
    const bool cached = ns <= TG_MAXS;
    if (cached) {
        for (uint32_t j = lane; j < ns; j += 64) { const TileStream e = A.list[l0 + j]; L.tsid[j] = e.stream; L.tstart[j] = e.start; }
        tg_sync();
        for (uint32_t j = lane; j < ns; j += 64) {      // rank sort: the list comes out of atomics in any order
            const uint32_t s = L.tsid[j];
            uint32_t r = 0;
            for (uint32_t q = 0; q < ns; ++q) r += L.tsid[q] < s ? 1u : 0u;
            L.sid[r] = s; L.sstart[r] = L.tstart[j];
        }
    }
    tg_sync();
    if (cached) {
        for (uint32_t j = lane; j < ns; j += 64) { const uint32_t s = L.sid[j]; L.stoff[j] = A.tag_off[s]; L.sts[j] = A.aln_t_s[s]; L.snt[j] = A.n_tags[s]; }
        tg_sync();
    }
    // The tags of a stream inside the tile are fetched one stream ahead: lane k holds the nibbles of tags i0 + k and i0 + 64 + k
    // (64 columns hold at least 64 tags of a stream that crosses them, so the second fetch is the rule) and, for the state in
    // front of tag i0, of tag i0 - 1 - k.
    uint32_t pf_back = 0, pf_a = 0, pf_b = 0;          // raw bytes (the loads are unconditional so that they stay in flight; indices are clamped)
    auto prefetch = [&](uint32_t j) {
        const uint8_t* tg = A.tags + L.stoff[j];
        const uint32_t i0 = L.sstart[j], nt = L.snt[j];
        pf_back = tg[(lane < i0 ? i0 - 1 - lane : 0u) >> 1];
        pf_a = tg[min(i0 + lane, nt - 1) >> 1];
        pf_b = tg[min(i0 + 64 + lane, nt - 1) >> 1];
    };
    auto nib_of = [](uint32_t byte, uint32_t i) { return (i & 1u) ? (byte & 15u) : (byte >> 4); };
    if (cached && ns) prefetch(0);
    TG_TICK(0);
    uint32_t prev_s = 0;
    for (uint32_t j = 0; j < ns; ++j) {
        uint32_t s, i0;
        const uint32_t raw_back = pf_back, raw_a = pf_a, raw_b = pf_b;
        if (cached) {
            s = L.sid[j]; i0 = L.sstart[j];
            if (j + 1 < ns) prefetch(j + 1);
        } else {            // a very crowded tile: the next stream by selection over the list
            unsigned long long best = ~0ull;
            for (uint32_t q = lane; q < ns; q += 64) {
                const TileStream e = A.list[l0 + q];
                const unsigned long long v = (unsigned long long)e.stream << 32 | e.start;
                if ((j == 0 || e.stream > prev_s) && v < best) best = v;
            }
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(best, o, 64); best = y < best ? y : best; }
            s = (uint32_t)(best >> 32); i0 = (uint32_t)best;
        }
        s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
        i0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i0);
        prev_s = s;
        if (!cached) {      // through the same LDS slots as the cached lists (one kind of load below: no generic pointers)
            if (lane == 0) { L.stoff[0] = A.tag_off[s]; L.sts[0] = A.aln_t_s[s]; L.snt[0] = A.n_tags[s]; }
            tg_sync();
        }
        const uint32_t jj = cached ? j : 0u;
        const uint8_t* tg = A.tags + L.stoff[jj];
        const uint32_t ts = L.sts[jj], nt = L.snt[jj];
        const uint32_t nb_back = lane < i0 ? nib_of(raw_back, i0 - 1 - lane) : 0u;          // in front of the stream: like a non-insertion tag
        const uint32_t nb_a = nib_of(raw_a, i0 + lane), nb_b = nib_of(raw_b, i0 + 64 + lane);
        // state in front of tag i0: position of the last non-insertion tag, the two tags before it (as node keys), the open insertion run
        int32_t T0 = (int32_t)max(c0, ts) - 1;
        unsigned long long P1 = KEY_HEAD, P2 = KEY_HEAD;
        uint32_t D0 = 0;
        if (i0 >= 1) {
            int32_t tp; uint32_t dl;
            const unsigned long long stop = cached ? __ballot(!(nb_back & 8u)) : 0ull;     // look-back lanes that end an insertion run
            const uint32_t nb1 = cached ? (uint32_t)__builtin_amdgcn_readlane((int)nb_back, 0) : tag_nib(tg, i0 - 1);
            if (stop) { tp = T0; dl = (uint32_t)__ffsll((long long)stop) - 1u; }
            else tag_pos(tg, i0 - 1, (uint32_t)(T0 + 1) - ts, ts, &tp, &dl);          // tags 0 .. i0-1 hold T0 + 1 - ts non-insertion tags
            P1 = node_key(tp, dl, nb1 & 7u);
            D0 = dl;
            if (i0 >= 2) {
                const uint32_t nb2 = cached ? (uint32_t)__builtin_amdgcn_readlane((int)nb_back, 1) : tag_nib(tg, i0 - 2);
                if (stop >> 1) { tp = T0 - ((nb1 & 8u) ? 0 : 1); dl = (uint32_t)__ffsll((long long)(stop >> 1)) - 1u; }
                else tag_pos(tg, i0 - 2, (uint32_t)(T0 + 1) - ts - ((nb1 & 8u) ? 0u : 1u), ts, &tp, &dl);
                P2 = node_key(tp, dl, nb2 & 7u);
            }
        }
        TG_TICK(1);
        for (uint32_t b = i0; b < nt; b += 64) {
            const uint32_t i = b + lane;
            const bool valid = i < nt;
            const uint32_t nb = !valid ? 8u : (cached && b == i0) ? nb_a : (cached && b == i0 + 64) ? nb_b : tag_nib(tg, i);
            const uint32_t base = nb & 7u;
            const unsigned long long M = __ballot(valid && !(nb & 8u));
            const unsigned long long Mle = M & lane_le;
            const uint32_t below = (uint32_t)__popcll(Mle);
            const int32_t tp = T0 + (int32_t)below;
            uint32_t delta = below ? lane - (63u - (uint32_t)__clzll((long long)Mle)) : D0 + lane + 1u;
            delta &= 0xffffu;
            const unsigned long long own = node_key(tp, delta, base);
            unsigned long long pp = __shfl_up(own, 1, 64), ppp = __shfl_up(own, 2, 64);
            if (lane == 0) { pp = P1; ppp = P2; }
            if (lane == 1) ppp = P1;
            const unsigned long long beyond = __ballot(valid && tp > (int32_t)c1);
            const bool obs = valid && tp <= (int32_t)c1 && base != 6u && key_base(pp) != 6u;
            const uint32_t col = (uint32_t)(tp - (int32_t)c0) & 63u;
            const unsigned long long k = tg_pack(delta, base, tp, pp, ppp);
            // the entries the column held before this fetch (a stream visits a node once: no two lanes carry the same entry)
            uint32_t hit = TG_NONE;
            if (obs) {
                uint32_t e = L.head[col];
                while (e != TG_NONE) {
                    if (L.key[e] == k) { hit = e; break; }
                    e = L.next[e];
                }
            }
            tg_sync();
            TG_TICK(2);
            if (obs) {
                if (hit != TG_NONE) atomicAdd(&L.link[hit], 1u);
                else {
                    const uint32_t e = atomicAdd(&L.used, 1u);
                    if (e < TG_POOL) {
                        L.key[e] = k; L.link[e] = 1u; L.next[e] = (uint16_t)TG_NONE; L.col[e] = (uint8_t)col;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                        const uint32_t prev = atomicExch(&L.tail[col], e);
                        if (prev == TG_NONE) L.head[col] = e; else L.next[prev] = (uint16_t)e;
                        atomicAdd(&L.cnt[col], 1u);
                    }
                }
            }
            tg_sync();
            TG_TICK(3);
            if (beyond) break;
            T0 += (int32_t)__popcll(M);
            D0 = (uint32_t)__builtin_amdgcn_readlane((int)delta, 63);
            P2 = __shfl(own, 62, 64);
            P1 = __shfl(own, 63, 64);
        }
    }
    tg_sync();
    TG_TICK(1);
    // ---- finish the tile: lane = column
    const uint32_t p = c0 + lane;
    uint32_t n = p < A.n_cols ? L.cnt[lane] : 0u;
    const bool over = L.used > TG_POOL || __ballot(n > TG_COLMAX) != 0ull;
    if (over) {
        if (lane == 0) {
            if (A.redo) A.redo[atomicAdd(&A.counter[2], 1u)] = t;
            else atomicOr(&A.counter[1], 1u);
        }
        if (p < A.n_cols) { A.col_off[p] = 0; A.col_nn[p] = 0; A.col_ne[p] = 0; }
        return;
    }
    uint32_t inc = n;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += v; }
    const uint32_t tile_total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    uint32_t basep = 0;
    if (lane == 0 && tile_total) basep = atomicAdd(&A.counter[0], tile_total);
    basep = (uint32_t)__builtin_amdgcn_readfirstlane((int)basep);
    // entries of the pool in any order (balanced over the lanes): position inside the column from one walk along its chain
    L.tail[lane] = basep + inc - n;          // the column's place in the graph arrays
    const uint32_t used = L.used;
    tg_sync();
    for (uint32_t e = lane; e < used; e += 64) {          // the first entry of every node
        const uint32_t c = L.col[e];
        if (c0 + c >= A.n_cols) continue;
        const uint32_t nk = tg_node(L.key[e]);
        bool first = true;
        for (uint32_t f = L.head[c]; f != e; f = L.next[f])
            if (tg_node(L.key[f]) == nk) { first = false; break; }
        if (first) { L.link[e] |= 0x80000000u; atomicAdd(&L.nn[c], 1u); }
    }
    tg_sync();
    for (uint32_t e = lane; e < used; e += 64) {
        const uint32_t c = L.col[e];
        if (c0 + c >= A.n_cols) continue;
        const unsigned long long k = L.key[e];
        const uint32_t nk = tg_node(k);
        uint32_t node_start = 0, node_idx = 0, node_len = 0, pos_in = 0;
        bool passed = false;
        for (uint32_t f = L.head[c]; f != TG_NONE; f = L.next[f]) {
            const uint32_t nkf = tg_node(L.key[f]);
            if (f == e) passed = true;
            if (nkf < nk) { ++node_start; node_idx += L.link[f] >> 31; }
            else if (nkf == nk) { ++node_len; if (!passed) ++pos_in; }
        }
        const uint32_t off = L.tail[c];
        Entry en;
        tg_unpack(k, (int32_t)(c0 + c), &en.pp, &en.ppp);
        en.score = 0;
        en.link = L.link[e] & 0xffffu;           // the reference counts in 16 bits
        en.node = nk;
        A.entries[off + node_start + pos_in] = en;
        if (L.link[e] >> 31) A.nodes[off + node_idx] = Node{nk, node_start, node_len, 0u};
    }
    if (p < A.n_cols) {
        A.col_off[p] = basep + inc - n;
        A.col_nn[p] = L.nn[lane];
        A.col_ne[p] = n;
    }
    TG_TICK(4);
    if (kProf && lane == 0) for (int i = 0; i < 6; ++i) atomicAdd(&prof[i], ph[i]);
#undef TG_TICK
}

// seed_len > 0 (a window): coverage = seed + scanned difference array of the reads (cov_pre exclusive prefix, cov_diff the
// array itself), max_size at least the seed's 1; seed_len == 0 (concatenated low-quality regions): counted directly
// predecessor entries of every entry, resolved once for both DP passes (np2_core.h EMatch): lane per entry slot
__global__ __launch_bounds__(256) void k2_match(MsaView mv, const uint32_t* __restrict__ slot_col, const uint8_t* __restrict__ live, uint32_t total,
                                                EMatch* __restrict__ match) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total || !live[g]) return;
    const Entry em = mv.entries[g];
    EMatch mt;
    mt.pe0 = 0;
    mt.n = 0;
    mt.ps0 = 0;
    mt.m4 = 0;
    (void)slot_col;
    if (key_tpos(em.pp) != -1) {
        const int32_t tp = key_tpos(em.pp);
        const Node* first = mv.nodes + mv.col_off[tp];
        const Node* ppn = find_node(mv, tp, key_delta(em.pp) << 8 | key_base(em.pp));
        if (ppn) {
            mt.pe0 = mv.col_off[tp] + ppn->start;
            uint32_t before = 0;
            for (const Node* qn = first; qn < ppn; ++qn) before += qn->len;
            mt.ps0 = (uint16_t)(before < 0xffffu ? before : 0xffffu);
            uint32_t n_found = 0;
            for (uint32_t n = 0; n < ppn->len; ++n)
                if (mv.entries[mt.pe0 + n].pp == em.ppp) {
                    if (n_found < MATCH_INLINE && n < 65536u) mt.m4 |= (uint64_t)n << (16 * n_found);
                    else n_found = MATCH_INLINE;   // does not fit: the consumers scan (n = MATCH_INLINE + 1 below)
                    ++n_found;
                }
            mt.n = (uint16_t)(n_found > MATCH_INLINE ? MATCH_INLINE + 1 : n_found);
        }
    }
    match[g] = mt;
}
// Where the low-quality scans of the window consensus have anything to look at (ctg_cns.c:1562-1725: the loop heads of get_l_del_regions
// and of the insertion scan), one bit per consensus base, written a 64-bit word per wave.  The comparisons are the reference's own:
// int against double for the deletion rule, float against float for the insertion rule (no contraction: -ffp-contract=off).
__global__ __launch_bounds__(256) void k2_lq_triggers(const ConsBase* __restrict__ cb, uint32_t len, const ColStat* __restrict__ st, float ratio1,
                                                      unsigned long long* __restrict__ trig_del, unsigned long long* __restrict__ trig_ins) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool del = false, ins = false;
    if (i < len) {
        const uint32_t pos = cb[i].pos;
        const ColStat c = st[pos];
        if (i >= 1) del = !((double)c.l_del < (double)c.coverage * 0.3 && pos < cb[i - 1].pos + 20u);
        ins = !((float)c.l_ins < (float)c.coverage * ratio1);
    }
    const unsigned long long md = __ballot(del), mi = __ballot(ins);
    if ((threadIdx.x & 63) == 0 && i < ((len + 63u) & ~63u)) { trig_del[i >> 6] = md; trig_ins[i >> 6] = mi; }
}


__global__ void k2_pack_stat(const uint32_t* coverage, const uint32_t* max_size, const uint32_t* l_ins, const uint32_t* l_del,
                             uint32_t n, ColStat* st, uint32_t seed_len, const uint32_t* cov_pre, const uint32_t* cov_diff, uint32_t* col_cnt) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    ColStat c;
    uint32_t cov = coverage[p], ms = max_size[p];
    if (seed_len) {
        cov = cov_pre[p] + cov_diff[p] + (p < seed_len ? 1u : 0u);
        if (p < seed_len && ms < 1u) ms = 1u;
        col_cnt[p] = cov + coverage[p];   // link observations of the column = its tags: one per covering stream + the insertion tags
    }
    c.coverage = (uint16_t)cov;
    c.max_size = (uint16_t)ms;
    c.l_ins = (uint16_t)l_ins[p];
    c.l_del = (uint16_t)l_del[p];
    st[p] = c;
}

struct DpResult {
    long long gbest;
    uint64_t gkey;
    uint32_t cons_len, status;   // status: 0 ok, 1 no end column, 2 backtrace left the graph, 3 zero coverage
};

// tag streams of gapped string pairs (the concatenated low-quality regions), chunk-parallel: the target position of
// a column is the number of non-gap target characters before it (count per chunk + scan), the open insertion run is
// found by a short look-back
// ------------------------------------------------------------------------------------------------
// Candidate strings of the low-quality regions straight from the tag streams in HBM (one lane per (stream, region)
// request): the tag holding window position `start` is found through the per-chunk counts of non-insertion tags the
// link stage already scanned (binary search over the stream's chunks, then 8 tags per word inside the chunk), then the
// bases up to position `end` are counted (kWrite = false) or written (kWrite = true).
struct SubReqDev { uint32_t stream, start, end, first_chunk, n_chunks; };
template <bool kWrite>
__global__ void k2_extract(const SubReqDev* rq, uint32_t n, const uint32_t* pre, const uint64_t* tag_off, const uint32_t* aln_t_s,
                           const uint8_t* tags, uint32_t* first_tag, uint32_t* len, const uint32_t* out_off, char* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SubReqDev r = rq[i];
    const uint8_t* tg = tags + tag_off[r.stream];
    const uint32_t ts = aln_t_s[r.stream];
    uint32_t j;
    if (!kWrite) {
        const uint32_t need = r.start - ts + 1;   // tag j is the need-th non-insertion tag of the stream
        uint32_t lo = 0, hi = r.n_chunks;         // last chunk whose prefix count is < need
        const uint32_t base = pre[r.first_chunk];
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pre[r.first_chunk + mid] - base < need) lo = mid;
            else hi = mid;
        }
        uint32_t have = pre[r.first_chunk + lo] - base;
        j = lo * LINK_CHUNK;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(tg) + (j >> 3);   // streams start 4-byte aligned, chunks are 256 bytes
        for (;;) {   // whole words while the target lies beyond them (a terminator nibble has bit 3 set: it never counts)
            const uint32_t c = 8u - (uint32_t)__popc(*w & 0x88888888u);
            if (have + c >= need) break;
            have += c;
            ++w;
            j += 8;
        }
        for (;; ++j)
            if (!(tag_nib(tg, j) & 8u) && ++have == need) break;
        first_tag[i] = j;
    } else {
        j = first_tag[i];
    }
    uint32_t t_pos = r.start, n_out = 0;
    char* o = kWrite ? out + out_off[i] : nullptr;
    const uint32_t j0 = j;
    for (;; ++j) {
        const uint32_t nb = tag_nib(tg, j);
        if (nb == 15u) break;
        if (!(nb & 8u) && j != j0 && ++t_pos > r.end) break;
        if ((nb & 7u) != 4u) {
            if (kWrite) o[n_out] = int_to_base(nb & 7u);
            ++n_out;
        }
    }
    if (!kWrite) len[i] = n_out;
}

// read coordinate of a stream at a window column (Exec::read_coords): a lane per request walks its stream's tag nibbles from the front --
// eight at a time while the column lies beyond the word -- counting the tags that carry a base; the column's own tag is its need-th
// non-insertion tag.  A few hundred requests per window with split reads; replaces the download of every tag stream of the window
// (~40 MB) that the host walk needed (round 5).
struct CoordReqDev { uint32_t stream, col, through_col, pad; };
__global__ void k2_read_coord(const CoordReqDev* rq, uint32_t n, const uint64_t* tag_off, const uint32_t* aln_t_s, const uint8_t* tags, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CoordReqDev r = rq[i];
    const uint8_t* tg = tags + tag_off[r.stream];
    const uint32_t need = r.col - aln_t_s[r.stream] + 1;      // the column's first tag is the need-th non-insertion tag of the stream
    const uint32_t* w = reinterpret_cast<const uint32_t*>(tg);   // streams start 4-byte aligned
    uint32_t have = 0, q = 0, j = 0;
    for (;;) {      // whole words while the target lies beyond them (a terminator nibble has bit 3 set: it never counts as a column)
        const uint32_t x = *w;
        const uint32_t cols = 8u - (uint32_t)__popc(x & 0x88888888u);
        if (have + cols >= need) break;
        if (x & (x >> 1) & (x >> 2) & (x >> 3) & 0x11111111u) break;      // the stream's terminator (15) is in this word: a column behind the
                                                                           // stream's last one was asked for -- everything up to the end counts
        const uint32_t y = (x & 0x77777777u) ^ 0x44444444u;      // a nibble is zero where the tag's base code is 4 (a gap)
        q += (uint32_t)__popc((y | (y >> 1) | (y >> 2)) & 0x11111111u);
        have += cols;
        ++w;
        j += 8;
    }
    for (;; ++j) {
        const uint32_t nb = tag_nib(tg, j);
        if (nb == 15u) break;                                     // (a column behind the stream's last one: the walk ends with the stream, like the host's)
        if (!(nb & 8u) && ++have == need) { if (r.through_col && (nb & 7u) != 4u) ++q; break; }
        if ((nb & 7u) != 4u) ++q;
    }
    out[i] = q;
}

struct StrChunk { uint32_t stream, c0, n, first_chunk_of_stream, last, pad0, pad1, pad2; };
__global__ void k2_str_count(const StrChunk* sc, uint32_t n_chunks, const char* pool, const uint64_t* str_off, uint32_t* cnt) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const StrChunk d = sc[c];
    const char* t = pool + str_off[2 * d.stream] + d.c0;
    uint32_t n = 0;
    for (uint32_t i = 0; i < d.n; ++i) n += t[i] != '-' ? 1u : 0u;
    cnt[c] = n;
}
__global__ void k2_tags_str_chunk(const StrChunk* sc, uint32_t n_chunks, const uint32_t* pre, const char* pool, const uint64_t* str_off,
                                  uint32_t gap_min_len, const uint64_t* tag_off, uint8_t* tags, DevStat st, uint32_t* te_out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const StrChunk d = sc[c];
    const char* t = pool + str_off[2 * d.stream];
    const char* qs = pool + str_off[2 * d.stream + 1];
    uint32_t delta = 0;
    if (d.c0) while (delta < d.c0 && t[d.c0 - 1 - delta] == '-') ++delta;
    EmitState es{pre[c] - pre[d.first_chunk_of_stream] - 1u, delta, delta >= gap_min_len ? 1u : 0u};
    StrColIter f{t, qs, d.c0};
    DevStatSink sink{st};
    emit_tags_range(f, d.c0, d.n, &es, d.last != 0, gap_min_len, tags + tag_off[d.stream], sink);
    if (d.last) te_out[d.stream] = es.te + 1;
}

// ---- run-decomposed chain DP ---------------------------------------------------------------------------------
// Everything to the right of a column depends on the left part of the window only through the scores of that
// column's entries.  Scores obey v = max(0, max_n(v_n) + w) (entries start at 0 and are only raised; stream heads
// inject constants), so every entry of a run is a function f(x) = max(c, max_i(x_i + a_i)) of the scores x of the
// entries of the run's left cut column, and a run maps x -> max(C, A (x) x) in the (max, +) semiring.
// Cuts are columns with at most K live entries (K = 8, or 32 where deep pileups leave too few narrow columns; at most one
// per block of CUT_BLOCK columns):
//   k2_cut_flags / scan / k2_cut_list   the cuts, in order
//   k2_run_ac     K + 1 lanes per run, lane i = input i (lane K = constants): (a, c) of every entry in a two-column ring, (A, C) of the run
//   k2_run_scan   x at every cut from the (A, C) chain, written into the cut columns' entries   [one lane, #runs steps]
//   k2_run_dp_a   wave per run (first lane): the reference's literal DP on the run's interior columns
//   k2_run_dp_b   the same on the cut columns themselves (their scores are recomputed to the same values; the
//                 best-index rules need the literal pass)
// Best-index rules compare real scores, which is why the literal passes run after the scan.
constexpr int RULE_LQ = 5, RULE_LQ_HIFI = 6;   // DP rules of the low-quality re-consensus, next to READS_ONT..READS_RS
constexpr long long AC_NEG = INT64_MIN / 4;   // "-infinity" that survives adding a column weight
constexpr uint32_t CUT_K_SMALL = 8, CUT_K_LARGE = 32, CUT_BLOCK = 32;   // cut width: 8, or 32 where deep pileups leave few narrow columns

__device__ __forceinline__ uint32_t live_entries(const MsaView& mv, uint32_t p) {
    const Node* nd = mv.nodes + mv.col_off[p];
    uint32_t n = 0;
    for (uint32_t j = 0; j < mv.col_nn[p]; ++j) n += nd[j].len;
    return n;
}
// one candidate per block of CUT_BLOCK columns: the first column with 1..K live entries among the block's first
// CUT_BLOCK - 2 columns (so two cuts are never adjacent: k2_run_dp_b relies on the column before a cut being interior)
__global__ void k2_cut_flags(MsaView mv, uint32_t n_cols, int32_t l, uint32_t* flag, uint32_t K) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t p0 = b * CUT_BLOCK;
    if (p0 > n_cols) return;
    bool found = false;
    for (uint32_t o = 0; o < CUT_BLOCK && p0 + o <= n_cols; ++o) {
        const uint32_t p = p0 + o;
        uint32_t f = 0;
        if (!found && o < CUT_BLOCK - 2 && (int32_t)p < l) {
            const uint32_t e = live_entries(mv, p);
            if (e >= 1 && e <= K) { f = 1; found = true; }
        }
        flag[p] = f;
    }
}
__global__ void k2_cut_list(const uint32_t* flag, const uint32_t* pos, uint32_t n_cols, uint32_t* cuts) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_cols && flag[p]) cuts[pos[p]] = p;
}

template <uint32_t K>
struct RunT {
    long long A[K][K];   // A[j][i]: output j from input i
    long long C[K];
    uint32_t n_out, pad;         // live entries of the right cut column
};

// run r covers columns (lo, hi]: lo = cuts[r-1] (or -1), hi = cuts[r] (or l-1 for the last, open run)
__device__ __forceinline__ void run_bounds(const uint32_t* cuts, uint32_t n_cuts, uint32_t r, int32_t l, int32_t* lo, int32_t* hi) {
    *lo = r == 0 ? -1 : (int32_t)cuts[r - 1];
    *hi = r < n_cuts ? (int32_t)cuts[r] : l - 1;
}
// state index of the entry with global index ge inside its column: live entries of the nodes before its node + its offset
// inside the node
__device__ __forceinline__ uint32_t col_state_index(const MsaView& mv, int32_t p, uint32_t ge) {
    const Node* nd = mv.nodes + mv.col_off[p];
    const uint32_t rel = ge - mv.col_off[p];
    uint32_t idx = 0;
    for (uint32_t j = 0; j < mv.col_nn[p]; ++j) {
        if (rel >= nd[j].start && rel < nd[j].start + nd[j].len) return idx + (rel - nd[j].start);
        idx += nd[j].len;
    }
    return 0xffffffffu;
}
// the coefficient vectors of a run live in a ring of two columns (an entry only ever looks at its own column and the one
// before it): size of the ring of every run = 2 * (most live entries of one of its columns), scanned into offsets
__global__ void k2_run_size(MsaView mv, const uint32_t* cuts, uint32_t n_cuts, uint32_t n_runs, int32_t l, uint32_t* size, const uint32_t* col_ne = nullptr) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_runs) return;
    if (r == n_runs) { size[r] = 0; return; }
    int32_t lo, hi;
    run_bounds(cuts, n_cuts, r, l, &lo, &hi);
    uint32_t mx = 0;
    for (int32_t p = lo + 1; p <= hi; ++p) {
        const uint32_t e = col_ne ? col_ne[p] : live_entries(mv, (uint32_t)p);     // a graph built by tiles keeps the count per column
        if (e > mx) mx = e;
    }
    size[r] = 2 * mx;
}

template <uint32_t K>
__global__ void k2_run_ac(MsaView mv, const uint32_t* cuts, uint32_t n_cuts, uint32_t n_runs, int32_t l, long long C, long long* ring,
                          const uint32_t* ring_off, RunT<K>* out, uint32_t rpw, uint32_t* status, const uint32_t* list = nullptr,
                          const uint32_t* n_list = nullptr) {
    // `rpw` runs per wave, K + 1 lanes each: lanes 0..K-1 carry the coefficient of one input (= one live entry of the
    // run's left cut column), lane K the constant.  A run is a serial chain of dependent loads, so what counts is how
    // many runs are in flight: small graphs (the low-quality re-consensus) take one run per wave, whole windows pack
    // 64 / (K + 1) runs per wave and let them diverge.
    constexpr uint32_t W = K + 1;
    uint32_t r = blockIdx.x * rpw + threadIdx.x / W;
    const uint32_t lane = threadIdx.x % W;
    if (r >= n_runs || threadIdx.x >= rpw * W) return;
    if (list) {      // only the runs k2_run_ac_lds left
        if (r >= *n_list) return;
        r = list[r];
    }
    int32_t lo, hi;
    run_bounds(cuts, n_cuts, r, l, &lo, &hi);
    long long* R = ring + (uint64_t)ring_off[r] * W;
    const uint64_t half = (uint64_t)((ring_off[r + 1] - ring_off[r]) / 2) * W;
    uint32_t jout = 0;
    for (int32_t p = lo + 1; p <= hi; ++p) {
        const long long cov = mv.stat[p].coverage;
        const Node* nd = mv.nodes + mv.col_off[p];
        const uint32_t nn = mv.col_nn[p];
        long long* cur = R + (uint64_t)(p & 1) * half;
        const long long* prev = R + (uint64_t)((p & 1) ^ 1) * half;
        uint32_t s = 0;   // state index of the entry inside its column
        for (uint32_t j = 0; j < nn; ++j) {
            for (uint32_t m = 0; m < nd[j].len; ++m, ++s) {
                const uint32_t g = mv.col_off[p] + nd[j].start + m;
                const Entry& em = mv.entries[g];
                const long long w = 10 * (long long)em.link - C * cov;
                long long v;   // this lane's component of the entry's function
                if (key_tpos(em.pp) == -1) {
                    v = lane == K ? w : AC_NEG;   // assigned directly, may be negative
                } else {
                    const int32_t tp = key_tpos(em.pp);
                    if (tp != p && tp != p - 1) { if (lane == 0) atomicMax(status, 4u); return; }   // a tag's predecessor is its neighbour
                    long long best = AC_NEG;
                    PredList pr;
                    pr.open(mv, em, g);
                    if (pr.cnt) {
                        const uint32_t g0 = (uint32_t)(pr.PE - mv.entries);
                        const bool fast = pr.listed && pr.mt.ps0 != 0xffffu;
                        for (uint32_t it = 0; it < pr.cnt; ++it) {
                            const uint32_t n = pr.listed ? pr.mt.at(it) : it;
                            if (!pr.listed && mv.entries[g0 + n].pp != em.ppp) continue;
                            const uint32_t sidx = fast ? pr.mt.ps0 + n : col_state_index(mv, tp, g0 + n);
                            long long vn;
                            if (tp == lo) vn = sidx == lane ? 0 : AC_NEG;   // the left cut's entries are the inputs
                            else vn = (tp == p ? cur : prev)[(uint64_t)sidx * W + lane];
                            if (vn > best) best = vn;
                        }
                    }
                    v = best > AC_NEG ? best + w : AC_NEG;
                    if (lane == K && v < 0) v = 0;   // clamp lives in the constant
                }
                if (p == hi && r < n_cuts) {   // an entry of the right cut: a row of the run's transfer
                    if (lane < K) out[r].A[jout][lane] = v;
                    else out[r].C[jout] = v;
                    ++jout;
                }
                cur[(uint64_t)s * W + lane] = v;
            }
        }
    }
    if (lane == 0) out[r].n_out = jout;
}

// ---- scores at the cuts: scan of the run transfers, grouped ------------------------------------------------------
// T2 o T1 in the (max, +) semiring with constants: A = A2 (x) A1, C = max(C2, A2 (x) C1).  Groups of SCAN_G runs are
// composed in parallel (one wave per group, the K x K outputs spread over the lanes), a wave chains the groups (two
// levels), then every group replays its runs from its true entry vector and writes the cut columns' scores.
constexpr uint32_t SCAN_G = 32;
__device__ __forceinline__ long long mp_add(long long a, long long b) { return (a > AC_NEG && b > AC_NEG) ? a + b : AC_NEG; }

template <uint32_t K>
__global__ __launch_bounds__(64) void k2_scan_groups(const RunT<K>* rt, uint32_t n_cuts, RunT<K>* gt) {
    __shared__ long long sA[K][K], sC[K], tA[K][K], tC[K], nA[K][K], nC[K];
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    const uint32_t r0 = g * SCAN_G, r1 = r0 + SCAN_G < n_cuts ? r0 + SCAN_G : n_cuts;
    // running transfer = run r0
    for (uint32_t x = lane; x < K * K; x += 64) sA[x / K][x % K] = rt[r0].A[x / K][x % K];
    for (uint32_t x = lane; x < K; x += 64) sC[x] = rt[r0].C[x];
    uint32_t n_mid = rt[r0].n_out;   // outputs of the running transfer = inputs of the next run
    __syncthreads();
    for (uint32_t r = r0 + 1; r < r1; ++r) {
        for (uint32_t x = lane; x < K * K; x += 64) tA[x / K][x % K] = rt[r].A[x / K][x % K];
        for (uint32_t x = lane; x < K; x += 64) tC[x] = rt[r].C[x];
        const uint32_t n_out = rt[r].n_out;
        __syncthreads();
        for (uint32_t x = lane; x < K * K; x += 64) {
            const uint32_t j = x / K, i = x % K;
            long long a = AC_NEG;
            if (j < n_out)
                for (uint32_t k = 0; k < n_mid; ++k) {
                    const long long v = mp_add(tA[j][k], sA[k][i]);
                    if (v > a) a = v;
                }
            nA[j][i] = a;
        }
        for (uint32_t j = lane; j < K; j += 64) {
            long long c = AC_NEG;
            if (j < n_out) {
                c = tC[j];
                for (uint32_t k = 0; k < n_mid; ++k) {
                    const long long v = mp_add(tA[j][k], sC[k]);
                    if (v > c) c = v;
                }
            }
            nC[j] = c;
        }
        __syncthreads();
        for (uint32_t x = lane; x < K * K; x += 64) sA[x / K][x % K] = nA[x / K][x % K];
        for (uint32_t x = lane; x < K; x += 64) sC[x] = nC[x];
        n_mid = n_out;
        __syncthreads();
    }
    for (uint32_t x = lane; x < K * K; x += 64) gt[g].A[x / K][x % K] = sA[x / K][x % K];
    for (uint32_t x = lane; x < K; x += 64) gt[g].C[x] = sC[x];
    if (lane == 0) gt[g].n_out = n_mid;
}
// entry vector of every group (x before its first run).  x_in == nullptr: one block walks all groups, group 0 starts
// from nothing.  Otherwise block s walks groups [s * SCAN_G, (s + 1) * SCAN_G) starting from x_in[s] (two-level scan:
// the groups are themselves composed in groups by k2_scan_groups and chained once at the top).  Lane j owns x[j].
template <uint32_t K>
__global__ __launch_bounds__(64) void k2_scan_chain(const RunT<K>* gt, uint32_t n_groups, const long long* x_in, const uint32_t* n_in_arr, long long* gx,
                                                    uint32_t* gn) {
    __shared__ long long xs[K];
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    const uint32_t g0 = x_in ? s * SCAN_G : 0u, g1 = x_in ? (g0 + SCAN_G < n_groups ? g0 + SCAN_G : n_groups) : n_groups;
    uint32_t n_in = x_in ? n_in_arr[s] : 0u;
    if (lane < K) xs[lane] = (x_in && lane < n_in) ? x_in[(uint64_t)s * K + lane] : AC_NEG;
    __syncthreads();
    for (uint32_t g = g0; g < g1; ++g) {
        if (lane < K) gx[(uint64_t)g * K + lane] = lane < n_in ? xs[lane] : AC_NEG;
        if (lane == 0) gn[g] = n_in;
        const uint32_t n_out = gt[g].n_out;
        long long v = AC_NEG;
        if (lane < n_out) {
            v = gt[g].C[lane];
            for (uint32_t i = 0; i < n_in; ++i) {
                const long long t = mp_add(xs[i], gt[g].A[lane][i]);
                if (t > v) v = t;
            }
        }
        __syncthreads();
        if (lane < K) xs[lane] = v;
        n_in = n_out;
        __syncthreads();
    }
}
// every group replays its runs; the cut columns' entries get their true scores so the literal passes can start from them
template <uint32_t K>
__global__ __launch_bounds__(64) void k2_scan_apply(MsaView mv, const uint32_t* cuts, uint32_t n_cuts, const RunT<K>* rt, const long long* gx, const uint32_t* gn) {
    __shared__ long long xs[K];
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    const uint32_t r0 = g * SCAN_G, r1 = r0 + SCAN_G < n_cuts ? r0 + SCAN_G : n_cuts;
    uint32_t n_in = gn[g];
    if (lane < K) xs[lane] = lane < n_in ? gx[(uint64_t)g * K + lane] : AC_NEG;
    __syncthreads();
    for (uint32_t r = r0; r < r1; ++r) {
        const uint32_t p = cuts[r];
        const uint32_t n_out = rt[r].n_out;
        long long v = AC_NEG;
        if (lane < n_out) {   // lane = state index of the entry inside the cut column
            v = rt[r].C[lane];
            for (uint32_t i = 0; i < n_in; ++i) {
                const long long t = mp_add(xs[i], rt[r].A[lane][i]);
                if (t > v) v = t;
            }
            // the entry with this state index
            const Node* nd = mv.nodes + mv.col_off[p];
            uint32_t left = lane;
            for (uint32_t j = 0; j < mv.col_nn[p]; ++j) {
                if (left < nd[j].len) { mv.entries[mv.col_off[p] + nd[j].start + left].score = v; break; }
                left -= nd[j].len;
            }
        }
        __syncthreads();
        if (lane < K) xs[lane] = v;
        n_in = n_out;
        __syncthreads();
    }
}

template <int TYPE>
__device__ __forceinline__ void dp_any(const MsaView& mv, int32_t p, int32_t l, long long* gbest, uint64_t* gkey) {
    if (TYPE == RULE_LQ) dp_column_lq<false>(mv, p);
    else if (TYPE == RULE_LQ_HIFI) dp_column_lq<true>(mv, p);
    else dp_column<TYPE == RULE_LQ || TYPE == RULE_LQ_HIFI ? READS_ONT : TYPE>(mv, p, l, gbest, gkey);
}
template <int TYPE>
__device__ __forceinline__ void publish_best(const MsaView& mv, int32_t l, long long gbest, uint64_t gkey, DpResult* res) {
    if (TYPE == RULE_LQ || TYPE == RULE_LQ_HIFI) gkey = node_key(l - 1, (uint32_t)mv.stat[l - 1].max_size - 1, 5);   // last node the reference's loops visit (ctg_cns.c:1036-1038,1090-1092)
    res->gbest = gbest;
    res->gkey = gkey;
    res->status = key_base(gkey) == 0xff ? 1u : 0u;
}

// interior columns of every run (and the last column of the open run)
template <int TYPE>
__global__ void k2_run_dp_a(MsaView mv, const uint32_t* cuts, uint32_t n_cuts, uint32_t n_runs, int32_t l, DpResult* res, uint32_t rpw,
                            const uint32_t* list = nullptr, const uint32_t* n_list = nullptr) {
    uint32_t r = blockIdx.x * rpw + threadIdx.x;   // rpw runs per wave, one lane each (see k2_run_ac)
    if (r >= n_runs || threadIdx.x >= rpw) return;
    if (list) {      // only the runs k2_run_dp_lds left
        if (r >= *n_list) return;
        r = list[r];
    }
    int32_t lo, hi;
    run_bounds(cuts, n_cuts, r, l, &lo, &hi);
    long long gbest = INT64_MIN;
    uint64_t gkey = node_key(0, 0, 0xff);
    const int32_t last = r < n_cuts ? hi - 1 : hi;
    for (int32_t p = lo + 1; p <= last; ++p) dp_any<TYPE>(mv, p, l, &gbest, &gkey);
    if (r >= n_cuts && hi == l - 1) publish_best<TYPE>(mv, l, gbest, gkey, res);
}
// the cut columns: scores back to the state update_msa left them in, then the literal column
template <int TYPE>
__global__ void k2_run_dp_b(MsaView mv, const uint32_t* cuts, uint32_t n_cuts, int32_t l, DpResult* res, uint32_t rpw) {
    const uint32_t r = blockIdx.x * rpw + threadIdx.x;
    if (r >= n_cuts || threadIdx.x >= rpw) return;
    const int32_t p = (int32_t)cuts[r];
    const Node* nd = mv.nodes + mv.col_off[p];
    for (uint32_t j = 0; j < mv.col_nn[p]; ++j)
        for (uint32_t m = 0; m < nd[j].len; ++m) mv.entries[mv.col_off[p] + nd[j].start + m].score = 0;
    long long gbest = INT64_MIN;
    uint64_t gkey = node_key(0, 0, 0xff);
    dp_any<TYPE>(mv, p, l, &gbest, &gkey);
    if (p == l - 1) publish_best<TYPE>(mv, l, gbest, gkey, res);
}

// ---- the same two passes over a run with its columns staged in LDS ---------------------------------------------------------
// k2_run_ac / k2_run_dp_a walk a run as one serial chain of dependent loads from HBM (entry -> resolved predecessors -> their
// coefficients / scores), 7 or 64 runs per wave; what they wait for is latency, and every lane of a load instruction touches
// its own cache line.  When the graph was built by tiles the entries of a column are live, contiguous and in state order (state
// index = offset inside the column) and consecutive columns follow each other inside a tile, so a wave can own ONE run and
//   * fetch up to 64 columns / RL_CAP entries of it (entries, resolved predecessors, nodes) into LDS with a few coalesced
//     copies (a chunk crosses at most one tile boundary = two contiguous pieces),
//   * keep the two-column ring of coefficient vectors / scores in LDS,
//   * work the columns from LDS: entries of one insertion depth are independent (a tag's predecessor is in the column before,
//     or in its own column one insertion level down), so they are taken 64 / (K + 1) at a time by k2_run_ac_lds (lanes =
//     entry x input) and a node per lane by k2_run_dp_lds (the best-predecessor rules are sequential inside a node).
// Runs these kernels cannot take (a column with more than RL_COLMAX entries) are listed and left to the kernels above.
constexpr uint32_t RL_CAP = 96, RL_COLMAX = 32;
struct RunStage {
    uint32_t c_off[64], c_pre[65], c_nn[64];
    uint16_t c_cov[64];
    Entry e[RL_CAP];
    EMatch m[RL_CAP];
};
// Fetches columns pf .. of the run (at most 64, at most RL_CAP entries; the first one may be the column in front of the part to
// work on, staged for its entries' keys) and returns how many were taken, 0 = the run is not for these kernels.  kNodes: the node
// records too (staged at the same offsets as the column's entries).
template <bool kNodes>
__device__ __forceinline__ uint32_t run_stage(const MsaView& mv, const uint32_t* __restrict__ col_ne, int32_t pf, int32_t p_end, RunStage& S, Node* nd_lds,
                                              uint32_t lane) {
    const int32_t p = pf + (int32_t)lane;
    const bool in = p <= p_end;
    const uint32_t ne = in ? col_ne[p] : 0u, off = in ? mv.col_off[p] : 0u;
    if (__ballot(ne > RL_COLMAX)) return 0;
    uint32_t inc = ne;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += v; }
    const uint32_t nc = (uint32_t)__popcll(__ballot(in && inc <= RL_CAP));      // a prefix of the lanes
    S.c_off[lane] = off;
    S.c_pre[lane] = inc - ne;
    if (lane + 1 == nc) S.c_pre[nc] = inc;
    if (in) { S.c_cov[lane] = mv.stat[p].coverage; S.c_nn[lane] = kNodes ? mv.col_nn[p] : 0u; }
    const uint32_t poff = __shfl_up(off, 1, 64), pne = __shfl_up(ne, 1, 64);
    unsigned long long brk = __ballot(lane < nc && (lane == 0 || off != poff + pne));
    while (brk) {          // contiguous pieces: columns ca .. cb
        const uint32_t ca = (uint32_t)__ffsll((long long)brk) - 1u;
        brk &= brk - 1;
        const uint32_t cb = brk ? (uint32_t)__ffsll((long long)brk) - 2u : nc - 1;
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)ca);
        const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)(inc - ne), (int)ca), f1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)cb);
        const uint4* ge = reinterpret_cast<const uint4*>(mv.entries + g0);
        uint4* le = reinterpret_cast<uint4*>(S.e + f0);
        for (uint32_t x = lane; x < 2 * (f1 - f0); x += 64) le[x] = ge[x];
        const uint4* gm = reinterpret_cast<const uint4*>(mv.match + g0);
        uint4* lm = reinterpret_cast<uint4*>(S.m + f0);
        for (uint32_t x = lane; x < f1 - f0; x += 64) lm[x] = gm[x];
        if (kNodes) {       // a column's nodes sit at the start of its range: copy the range, the tail is never read
            const uint4* gn = reinterpret_cast<const uint4*>(mv.nodes + g0);
            uint4* ln = reinterpret_cast<uint4*>(nd_lds + f0);
            for (uint32_t x = lane; x < f1 - f0; x += 64) ln[x] = gn[x];
        }
    }
    tg_sync();
    return nc;
}
// The predecessor entries of `em` when their list is not inline (more than MATCH_INLINE): the entries of the predecessor node
// (column staged at pbase, pcount entries; the node starts at state index ps0) whose own predecessor is em.ppp, in list order.
template <class F>
__device__ __forceinline__ void pred_scan(const RunStage& S, uint32_t pbase, uint32_t pcount, const Entry& em, uint32_t ps0, F&& f) {
    const uint32_t ppk = key_delta(em.pp) << 8 | key_base(em.pp);
    for (uint32_t s = ps0; s < pcount && S.e[pbase + s].node == ppk; ++s)
        if (S.e[pbase + s].pp == em.ppp) f(s);
}

template <uint32_t K>
__global__ __launch_bounds__(64) void k2_run_ac_lds(MsaView mv, const uint32_t* __restrict__ col_ne, const uint32_t* cuts, uint32_t n_cuts, uint32_t n_runs,
                                                    int32_t l, long long C, RunT<K>* out, uint8_t* __restrict__ runflag, uint32_t* __restrict__ runlist,
                                                    uint32_t* __restrict__ runctr, uint32_t* status) {
    constexpr uint32_t W = K + 1, EPP = 64 / W;
    __shared__ RunStage S;
    __shared__ long long ring[2][RL_COLMAX][W];
    const uint32_t r = blockIdx.x, lane = threadIdx.x;
    const uint32_t sub = lane / W, li = lane % W;
    int32_t lo, hi;
    run_bounds(cuts, n_cuts, r, l, &lo, &hi);
    uint32_t n_last = 0;
    for (int32_t p0 = lo + 1; p0 <= hi;) {
        const uint32_t ctx = p0 > 0 ? 1u : 0u;          // the column in front comes along
        const int32_t pf = p0 - (int32_t)ctx;
        const uint32_t nc = run_stage<false>(mv, col_ne, pf, hi, S, nullptr, lane);
        if (nc <= ctx) {       // left to k2_run_ac / k2_run_dp_a: runlist[0 .. runctr[0])
            if (lane == 0) { runflag[r] = 1; runlist[atomicAdd(&runctr[0], 1u)] = r; }
            return;
        }
        for (uint32_t c = ctx; c < nc; ++c) {
            const int32_t p = pf + (int32_t)c;
            const uint32_t base = S.c_pre[c], n = S.c_pre[c + 1] - base;
            const uint32_t pbase = c ? S.c_pre[c - 1] : 0u, pn = c ? base - pbase : 0u;
            const long long cov = S.c_cov[c];
            long long (*cur)[W] = ring[p & 1];
            long long (*prev)[W] = ring[(p & 1) ^ 1];
            n_last = n;
            for (uint32_t s0 = 0; s0 < n;) {
                // entries of one insertion depth
                const uint32_t d0 = S.e[base + s0].node >> 8;
                const unsigned long long same = __ballot(s0 + lane < n && (S.e[base + s0 + lane].node >> 8) == d0);
                const uint32_t s1 = s0 + (~same ? (uint32_t)__ffsll((long long)~same) - 1u : 64u);
                for (uint32_t q = s0; q < s1; q += EPP) {
                    const uint32_t s = q + sub;
                    if (sub < EPP && s < s1) {
                        const Entry& em = S.e[base + s];
                        const long long w = 10 * (long long)em.link - C * cov;
                        long long v;
                        if (key_tpos(em.pp) == -1) v = li == K ? w : AC_NEG;
                        else {
                            const int32_t tp = key_tpos(em.pp);
                            if (tp != p && tp != p - 1) { atomicMax(status, 4u); v = AC_NEG; }
                            else {
                                const EMatch mt = S.m[base + s];
                                long long best = AC_NEG;
                                auto take = [&](uint32_t sidx) {
                                    long long vn;
                                    if (tp == lo) vn = sidx == li ? 0 : AC_NEG;        // the left cut's entries are the inputs
                                    else vn = (tp == p ? cur : prev)[sidx][li];
                                    if (vn > best) best = vn;
                                };
                                if (mt.n <= MATCH_INLINE) for (uint32_t it = 0; it < mt.n; ++it) take((uint32_t)mt.ps0 + mt.at(it));
                                else pred_scan(S, tp == p ? base : pbase, tp == p ? n : pn, em, mt.ps0, take);
                                v = best > AC_NEG ? best + w : AC_NEG;
                                if (li == K && v < 0) v = 0;
                            }
                        }
                        if (p == hi && r < n_cuts) {
                            if (li < K) out[r].A[s][li] = v;
                            else out[r].C[s] = v;
                        }
                        cur[s][li] = v;
                    }
                }
                tg_sync();
                s0 = s1;
            }
        }
        p0 = pf + (int32_t)nc;
    }
    if (lane == 0) { out[r].n_out = r < n_cuts ? n_last : 0u; runflag[r] = 0; }
}

template <int TYPE>
__global__ __launch_bounds__(64) void k2_run_dp_lds(MsaView mv, const uint32_t* __restrict__ col_ne, const uint32_t* cuts, uint32_t n_cuts, uint32_t n_runs,
                                                    int32_t l, DpResult* res, const uint8_t* __restrict__ runflag) {
    __shared__ RunStage S;
    __shared__ Node nd_lds[RL_CAP];
    __shared__ long long ring[2][RL_COLMAX];
    const uint32_t r = blockIdx.x, lane = threadIdx.x;
    if (runflag[r]) return;
    constexpr long long C = (TYPE == RULE_LQ ? 2 : (TYPE == READS_HIFI || TYPE == RULE_LQ_HIFI) ? 4 : 3);
    int32_t lo, hi;
    run_bounds(cuts, n_cuts, r, l, &lo, &hi);
    const int32_t last = r < n_cuts ? hi - 1 : hi;
    if (lo >= 0) {           // the left cut's scores (k2_scan_apply left them in its entries)
        const uint32_t n = col_ne[lo];
        if (lane < n && lane < RL_COLMAX) ring[lo & 1][lane] = mv.entries[mv.col_off[lo] + lane].score;
    }
    tg_sync();
    long long gbest = INT64_MIN;
    uint64_t gkey = node_key(0, 0, 0xff);
    for (int32_t p0 = lo + 1; p0 <= last;) {
        const uint32_t ctx = p0 > 0 ? 1u : 0u;
        const int32_t pf = p0 - (int32_t)ctx;
        const uint32_t nc = run_stage<true>(mv, col_ne, pf, last, S, nd_lds, lane);
        if (nc <= ctx) return;          // cannot happen: k2_run_ac_lds took the run (same columns and one more)
        for (uint32_t c = ctx; c < nc; ++c) {
            const int32_t p = pf + (int32_t)c;
            const uint32_t base = S.c_pre[c], n = S.c_pre[c + 1] - base, nn = S.c_nn[c];
            const uint32_t pbase = c ? S.c_pre[c - 1] : 0u, pn = c ? base - pbase : 0u;
            const long long cov = S.c_cov[c];
            long long* cur = ring[p & 1];
            const long long* prev = ring[(p & 1) ^ 1];
            for (uint32_t k0 = 0; k0 < nn;) {
                const uint32_t d0 = nd_lds[base + k0].key >> 8;
                const unsigned long long same = __ballot(k0 + lane < nn && (nd_lds[base + k0 + lane].key >> 8) == d0);
                const uint32_t k1 = k0 + (~same ? (uint32_t)__ffsll((long long)~same) - 1u : 64u);
                if (k0 + lane < k1) {         // lane = node: np2_core.h dp_column / dp_column_lq on staged data
                    Node& pb = nd_lds[base + k0 + lane];
                    Entry* E = S.e + base + pb.start;
                    const EMatch* M = S.m + base + pb.start;
                    const uint32_t b = pb.key & 0xffu;
                    uint32_t best = 0;
                    long long p_pp_score_ = INT64_MIN, p_pp_score = INT64_MIN;
                    int tmp = 0;
                    if (TYPE == READS_ONT)
                        for (uint32_t mi = 0; mi < pb.len; ++mi)
                            if ((int)E[mi].link > tmp) tmp = (int)E[mi].link;
                    for (uint32_t mi = 0; mi < pb.len; ++mi) {
                        Entry& em = E[mi];
                        long long sc = 0;
                        const long long w = 10 * (long long)em.link - C * cov;
                        if (key_tpos(em.pp) == -1) sc = w;
                        else {
                            const int32_t tp = key_tpos(em.pp);
                            const long long* src = tp == p ? cur : prev;
                            const EMatch mt = M[mi];
                            const uint32_t ppb = key_base(em.pp), pppb = key_base(em.ppp);
                            auto take = [&](uint32_t sidx) {
                                const long long en_score = src[sidx];
                                const long long cand = en_score + w;
                                if (cand > sc) { sc = cand; p_pp_score_ = en_score; }
                                if (TYPE == READS_CLR || TYPE == READS_HIFI || TYPE == RULE_LQ_HIFI) {
                                    if (en_score > p_pp_score || (en_score == p_pp_score && ppb != 4)) { best = mi; p_pp_score = en_score; }
                                } else if (TYPE == READS_ONT) {
                                    if (((key_delta(em.ppp) > 1 || key_delta(em.pp) > 0) && ((double)em.link > (double)cov * 0.2 || (int)em.link > tmp / 2)) ||
                                        ((int)em.link > (int)E[best].link / 2 && en_score > p_pp_score && (ppb == 4 || ppb == b || pppb == b || ppb == pppb))) {
                                        best = mi;
                                        p_pp_score = en_score;
                                    }
                                } else if (TYPE == RULE_LQ) {
                                    if ((int)em.link > (int)E[best].link / 2 && en_score > p_pp_score && (ppb == 4 || ppb == b || pppb == b || ppb == pppb)) {
                                        best = mi;
                                        p_pp_score = en_score;
                                    }
                                }
                            };
                            if (mt.n <= MATCH_INLINE) for (uint32_t it = 0; it < mt.n; ++it) take((uint32_t)mt.ps0 + mt.at(it));
                            else pred_scan(S, tp == p ? base : pbase, tp == p ? n : pn, em, mt.ps0, take);
                        }
                        em.score = sc;
                        cur[pb.start + mi] = sc;
                        if (TYPE == READS_RS) {
                            if (sc >= E[best].score) { best = mi; p_pp_score = p_pp_score_; }
                        } else if (sc > E[best].score || (sc == E[best].score && key_base(em.pp) != 4)) {
                            best = mi;
                            p_pp_score = p_pp_score_;
                        }
                    }
                    pb.best = best;
                }
                tg_sync();
                k0 = k1;
            }
            if ((TYPE != RULE_LQ && TYPE != RULE_LQ_HIFI) && p == l - 1) {     // global best node: nodes in order, the last of equals wins
                for (uint32_t k = 0; k < nn; ++k) {
                    const Node& pb = nd_lds[base + k];
                    if (!pb.len) continue;
                    const long long sc = S.e[base + pb.start + pb.best].score;
                    if (sc >= gbest) {
                        gkey = node_key(p, pb.key >> 8, pb.key & 0xffu);
                        if (sc > gbest) gbest = sc;
                    }
                }
            }
        }
        // scores and best indices back to the graph (not the column in front: its staged scores are not the real ones)
        const uint32_t n_total = S.c_pre[nc];
        for (uint32_t f = S.c_pre[ctx] + lane; f < n_total; f += 64) {
            uint32_t clo = 0, chi = nc;          // column of staged entry f
            while (chi - clo > 1) { const uint32_t mid = (clo + chi) >> 1; if (S.c_pre[mid] <= f) clo = mid; else chi = mid; }
            const uint32_t g = S.c_off[clo] + (f - S.c_pre[clo]);
            mv.entries[g].score = S.e[f].score;
            if (f - S.c_pre[clo] < S.c_nn[clo]) mv.nodes[g].best = nd_lds[f].best;
        }
        tg_sync();
        p0 = pf + (int32_t)nc;
    }
    if (r >= n_cuts && hi == l - 1 && lane == 0) publish_best<TYPE>(mv, l, gbest, gkey, res);
}

// ---- parallel backtrace over the runs --------------------------------------------------------------------------
// The walk of a run (from a node of its right cut column, or from the global best node for the open run, down to
// the first node it reaches in its left cut column) does not depend on the other runs.  Pass 0 walks from every
// node of the right cut column (lanes = start nodes) and records exit node, base count and whether the walk ended on
// a stream head; a one-lane chain picks the start of every run from right to left and places the runs; pass 1 walks
// again from the chosen starts and writes.  kLq: character output of the low-quality re-consensus
// (ctg_cns.c:1104-1143) instead of consensus bases (:1836-1858).
struct BtWalk { uint32_t exit_idx; uint32_t count; uint32_t ended; uint32_t pad; };   // per (run, start node); exit_idx = node index in the left cut column
struct BtPick { uint32_t start; uint32_t count; uint32_t off; uint32_t used; };   // per run

template <bool kLq, bool kWrite>
__global__ void k2_bt_runs(MsaView mv, const uint32_t* cuts, uint32_t n_cuts, uint32_t n_runs, int32_t l, const DpResult* res, BtWalk* walks,
                           const BtPick* pick, ConsBase* cons, char* chars, uint32_t* status, uint32_t K, uint32_t rpw) {
    // a walk is a chain of dependent loads: `rpw` runs share a wave (K start nodes each on the first pass, the one chosen start
    // on the second) and diverge
    const uint32_t lanes_per_run = kWrite ? 1u : K;
    const uint32_t r = blockIdx.x * rpw + threadIdx.x / lanes_per_run;
    if (r >= n_runs || threadIdx.x >= rpw * lanes_per_run) return;
    int32_t lo, hi;
    run_bounds(cuts, n_cuts, r, l, &lo, &hi);
    uint32_t s = threadIdx.x % lanes_per_run;
    uint64_t cur;
    if (kWrite) {
        if (!pick[r].used) return;
        s = pick[r].start;
    }
    if (r < n_cuts) {
        if (s >= mv.col_nn[hi]) return;
        const Node& nd = mv.nodes[mv.col_off[hi] + s];
        cur = node_key(hi, nd.key >> 8, nd.key & 0xffu);
    } else {
        if (s != 0) return;
        cur = res->gkey;
    }
    uint32_t n = 0;
    const uint32_t base_off = kWrite ? pick[r].off : 0u, total = kWrite ? pick[r].count : 0u;
    bool ended = false;
    for (;;) {
        const int32_t tp = key_tpos(cur);
        Node* nd = find_node(mv, tp, key_delta(cur) << 8 | key_base(cur));
        if (!nd || nd->len == 0) { if (kWrite) atomicMax(status, 2u); else { ended = true; n = 0xffffffffu; } break; }
        const Entry& be = mv.entries[mv.col_off[tp] + nd->start + nd->best];
        if (key_base(cur) != 4) {
            if (kWrite) {
                const uint32_t at = base_off + (total - 1 - n);   // forward order inside the run
                const uint32_t cov = mv.stat[tp].coverage;
                const char up = int_to_base(key_base(cur));
                const char low = (char)(up >= 'A' && up <= 'Z' ? up + 32 : up);
                if (kLq) {
                    chars[at] = ((be.link & 0xffffu) * 5 > cov || up == 'N') ? up : low;
                } else {
                    if (cov == 0) { atomicMax(status, 3u); break; }
                    ConsBase cb;
                    cb.qv = (char)(100 * be.link / cov);
                    cb.base = (cov > 4u && cb.qv > 20) ? up : low;
                    cb.pos = (uint32_t)tp;
                    cons[at] = cb;
                }
            }
            ++n;
        }
        cur = be.pp;
        if (key_tpos(cur) == -1) { ended = true; break; }
        if (key_tpos(cur) <= lo) break;   // reached the left cut column: that node starts the previous run's walk
    }
    if (!kWrite) {
        uint32_t xi = 0xffffffffu;
        if (!ended && lo >= 0) {   // index of the exit node among the nodes of the left cut column
            const Node* nl = mv.nodes + mv.col_off[lo];
            const uint32_t key = key_delta(cur) << 8 | key_base(cur), nn = mv.col_nn[lo];
            for (uint32_t j = 0; j < nn && j < K; ++j)
                if (nl[j].key == key && key_tpos(cur) == lo) xi = j;
            if (xi == 0xffffffffu) n = 0xffffffffu;   // cannot happen on a consistent graph
        }
        walks[(uint64_t)r * K + s] = BtWalk{xi, n, ended ? 1u : 0u, 0};
    }
}

// The chain "start node of run r -> exit node = start node of run r - 1" is a composition of maps on at most K (cut width)
// states, so it is grouped like the score scan: groups of BT_G runs are composed for every possible start (lanes =
// starts), one lane chains the groups, then every group replays its runs and places them.
constexpr uint32_t BT_G = 32;
struct BtGroup { uint32_t exit_idx, count, ended, invalid; };   // per (group, start at the group's right-most run)
struct BtGroupPick { uint32_t start, off, used, pad; };

__global__ void k2_bt_groups(const BtWalk* walks, uint32_t n_runs, BtGroup* grp, uint32_t K) {
    const uint32_t g = blockIdx.x, s = threadIdx.x;
    if (s >= K) return;
    const uint32_t r0 = g * BT_G, r1 = r0 + BT_G < n_runs ? r0 + BT_G : n_runs;
    uint32_t cur = s, cnt = 0, ended = 0, invalid = 0;
    for (uint32_t r = r1; r-- > r0;) {
        const BtWalk w = walks[(uint64_t)r * K + cur];
        if (w.count == 0xffffffffu) { invalid = 1; break; }
        cnt += w.count;
        if (w.ended) { ended = 1; break; }
        cur = w.exit_idx;
        if (cur >= K) { invalid = 1; break; }
    }
    grp[(uint64_t)g * K + s] = BtGroup{cur, cnt, ended, invalid};
}
__global__ __launch_bounds__(64) void k2_bt_chain(MsaView mv, const uint32_t* cuts, uint32_t n_cuts, uint32_t n_runs, int32_t l, const DpResult* res, const BtGroup* grp,
                                                   uint32_t n_groups, BtGroupPick* gp, uint32_t* total_out, uint32_t* status, uint32_t K) {
    if (blockIdx.x) return;
    const uint32_t lane = threadIdx.x;
    // start of the right-most run: the open run has a single start (the global best node); a window ending on a cut
    // column starts from that node's index among the cut's nodes
    uint32_t s = 0;
    if (n_runs - 1 < n_cuts) {
        const uint64_t cur = res->gkey;
        const int32_t hi = (int32_t)cuts[n_runs - 1];
        const Node* nd = mv.nodes + mv.col_off[hi];
        const uint32_t key = key_delta(cur) << 8 | key_base(cur), nn = mv.col_nn[hi];
        while (s < nn && nd[s].key != key) ++s;
        if (key_tpos(cur) != hi || s >= nn || s >= K) { if (lane == 0) { *status = 2; *total_out = 0; } return; }
    }
    // Right to left.  The map of a group does not depend on where the chain enters it, so the maps of 64 / K groups are fetched at
    // once (lanes = group x start) and the chain is threaded through them with lane reads; the fetch of the next batch is in
    // flight meanwhile.
    const uint32_t gpb = 64u / K;                      // groups per batch
    const uint32_t sub = lane / K, st = lane % K;      // lane = (group gi - sub, start st)
    int64_t first_used = (int64_t)n_groups;
    bool stop = false;
    BtGroup nxt{0, 0, 0, 0};
    {
        const int64_t g = (int64_t)n_groups - 1 - (int64_t)sub;
        if (sub < gpb && g >= 0) nxt = grp[(uint64_t)g * K + st];
    }
    for (int64_t gi = (int64_t)n_groups - 1; gi >= 0 && !stop; gi -= gpb) {
        const BtGroup e = nxt;
        {
            const int64_t g = gi - (int64_t)gpb - (int64_t)sub;
            if (sub < gpb && g >= 0) nxt = grp[(uint64_t)g * K + st];
        }
        for (uint32_t k = 0; k < gpb && gi - (int64_t)k >= 0; ++k) {
            const int src = (int)(k * K + s);
            const uint32_t exit_idx = (uint32_t)__builtin_amdgcn_readlane((int)e.exit_idx, src), cnt = (uint32_t)__builtin_amdgcn_readlane((int)e.count, src);
            const uint32_t ended = (uint32_t)__builtin_amdgcn_readlane((int)e.ended, src), invalid = (uint32_t)__builtin_amdgcn_readlane((int)e.invalid, src);
            if (invalid) { if (lane == 0) *status = 2; stop = true; break; }
            if (lane == 0) gp[gi - k] = BtGroupPick{s, cnt, 1, 0};   // off holds the count until the prefix pass below
            first_used = gi - (int64_t)k;
            if (ended) { stop = true; break; }
            s = exit_idx;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __builtin_amdgcn_wave_barrier();
    for (int64_t q = lane; q < first_used; q += 64) gp[q] = BtGroupPick{0, 0, 0, 0};
    uint32_t carry = 0;
    for (int64_t q0 = first_used; q0 < (int64_t)n_groups; q0 += 64) {
        const int64_t q = q0 + lane;
        const uint32_t c = q < (int64_t)n_groups ? gp[q].off : 0u;
        uint32_t inc = c;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += v; }
        if (q < (int64_t)n_groups) gp[q].off = carry + inc - c;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (lane == 0) *total_out = carry;
}
__global__ void k2_bt_place(const BtWalk* walks, uint32_t n_runs, const BtGroupPick* gp, BtPick* pick, uint32_t K) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r0 = g * BT_G;
    if (r0 >= n_runs) return;
    const uint32_t r1 = r0 + BT_G < n_runs ? r0 + BT_G : n_runs;
    for (uint32_t r = r0; r < r1; ++r) pick[r] = BtPick{0, 0, 0, 0};
    if (!gp[g].used) return;
    uint32_t cur = gp[g].start, first = r1;
    for (uint32_t r = r1; r-- > r0;) {
        const BtWalk w = walks[(uint64_t)r * K + cur];
        pick[r] = BtPick{cur, w.count, 0, 1};
        first = r;
        if (w.ended) break;
        cur = w.exit_idx;
    }
    uint32_t off = gp[g].off;
    for (uint32_t r = first; r < r1; ++r) { pick[r].off = off; off += pick[r].count; }
}


// ---- pseudo-seeds of the low-quality regions: one resident wave per job slot (np2_poa_dev.h) --------------------------------
// A launch works through a list of job numbers (`order`, longest first) that its waves take off ONE atomic counter (round 4 dealt jobs in a
// fixed stride: the slot that drew the long regions finished last with the others idle).  Three launches per window: the Small class (graph
// indices in bytes, score table in LDS, 9 waves a CU) on the jobs the host expects to fit it, at the same time the Big class (table in HBM
// scratch, the last two rows in LDS) on the rest, then the Big class again on what the Small one gave back (redo_only: status 1); what Big
// gives back goes to the host version.
template <class C>
__global__ __launch_bounds__(64) void k2_poa(const char* __restrict__ pool, const uint32_t* __restrict__ str_off, const uint32_t* __restrict__ str_len,
                                             const np2poa::Job* __restrict__ jobs, const uint32_t* __restrict__ order, uint32_t n_order, uint32_t redo_only,
                                             int32_t* tabS, uint32_t* tabF, uint32_t tab_cap, char* out_pool, uint32_t* out_len, uint32_t* status, uint32_t* queue, uint32_t* dbg) {
    __shared__ np2poa::PoaLdsT<C> L;
    if (dbg) dbg += 4 * blockIdx.x;
    int32_t* TS = C::TAB_LDS ? nullptr : tabS + (size_t)blockIdx.x * tab_cap;
    uint32_t* TF = C::TAB_LDS ? nullptr : tabF + (size_t)blockIdx.x * tab_cap;
    // (a wave takes n_order jobs at most, plus the one fetch that tells it the list is empty: the bound costs nothing and makes the loop end
    // whatever a future compiler does with the wave-uniform idiom below -- a job nobody reached keeps status "not done" and goes to the host)
    for (uint32_t taken = 0; taken <= n_order; ++taken) {
        uint32_t at = 0;
        if (threadIdx.x == 0) at = atomicAdd(queue, 1u);
        at = np2poa::uni(at);
        if (at >= n_order) break;
        const uint32_t j = order[at];
        if (dbg && threadIdx.x == 0) { __atomic_store_n(dbg + 2, j, __ATOMIC_RELAXED); __atomic_store_n(dbg, 0u, __ATOMIC_RELAXED); }
        const np2poa::Job J = jobs[j];
        // Every path through the body ends in the same lane-0 store + wave barrier: a `continue` behind a store of lane 0 alone let the
        // compiler run the other 63 lanes ahead into the next iteration, where readfirstlane then broadcast THEIR counter value (0)
        // instead of lane 0's fresh one -- an endless loop (round 5, found with NP2_POA_DEBUG; the wave-uniform idiom needs the wave to
        // reconverge before the back edge).
        const bool run = redo_only ? status[j] != 0 : true;
        bool ok = false;
        if (run) ok = np2poa::poa_region<C>(pool, str_off, str_len, J, TS, TF, tab_cap, out_pool, &out_len[j], &L, dbg);
        if (threadIdx.x == 0 && run) status[j] = ok ? 0u : 1u;
        if (dbg && threadIdx.x == 0) __atomic_store_n(dbg, ok ? 100u : 101u, __ATOMIC_RELAXED);
        np2poa::lds_sync();
    }
}

// ---- exclusive scan of uint32 counts (three launches: block sums, scan of the sums, final)
constexpr uint32_t SCAN_T = 256, SCAN_PER = 16, SCAN_TILE = SCAN_T * SCAN_PER;
__global__ __launch_bounds__(SCAN_T) void k2_scan_sums(const uint32_t* v, uint32_t n, uint32_t* sums) {
    __shared__ uint32_t sh[SCAN_T];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER;
    uint32_t s = 0;
    for (uint32_t i = 0; i < SCAN_PER; ++i)
        if (base + i < n) s += v[base + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = SCAN_T / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = sh[0];
}
__global__ void k2_scan_top(uint32_t* sums, uint32_t nb) {   // one lane: a few thousand block sums at most
    if (blockIdx.x || threadIdx.x) return;
    uint32_t run = 0;
    for (uint32_t i = 0; i < nb; ++i) { const uint32_t t = sums[i]; sums[i] = run; run += t; }
    sums[nb] = run;
}
__global__ __launch_bounds__(SCAN_T) void k2_scan_final(const uint32_t* v, uint32_t n, const uint32_t* sums, uint32_t* out) {
    __shared__ uint32_t sh[SCAN_T];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER;
    uint32_t loc[SCAN_PER];
    uint32_t s = 0;
    for (uint32_t i = 0; i < SCAN_PER; ++i) { loc[i] = base + i < n ? v[base + i] : 0u; s += loc[i]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = sums[blockIdx.x];
        for (uint32_t i = 0; i < SCAN_T; ++i) { const uint32_t t = sh[i]; sh[i] = run; run += t; }
    }
    __syncthreads();
    uint32_t run = sh[threadIdx.x];
    for (uint32_t i = 0; i < SCAN_PER; ++i)
        if (base + i < n) { out[base + i] = run; run += loc[i]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = sums[gridDim.x];
}

inline uint32_t nblk(uint64_t n, uint32_t t) { return (uint32_t)((n + t - 1) / t); }


// ---------------------------------------------------------------------------------------------------------------------
// Low-quality regions: candidate-to-seed alignment and assembly of the concatenated gapped strings on the device
// (generate_consensus_trimed, ctg_cns.c:1287-1414; align.c:39-177 -> np2_ond_dev.h)
struct OndRegion { uint64_t seed_off; uint32_t seed_len, first_cand, n_cand, first_pair; };   // first_pair: pair index of round 0 (pair_of below)
constexpr uint32_t PIECE_SEED = 0, PIECE_LQSEQ = 1, PIECE_ALN = 2;

__global__ __launch_bounds__(256) void k2_ond_align(const uint8_t* __restrict__ pool, const np2ond::Pair* __restrict__ pairs, uint32_t n_pairs,
                                                    uint8_t* out_pool, np2ond::PairResult* __restrict__ res, int32_t* vbuf, int32_t* lobuf, uint64_t* chbuf,
                                                    uint32_t max_d_cap, uint32_t row_words) {
    __shared__ uint8_t sq[4][np2ond::STR_CAP];
    __shared__ uint8_t st[4][np2ond::STR_CAP];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slot = blockIdx.x * 4 + wave, n_slots = gridDim.x * 4;
    np2ond::WaveScratch W{vbuf + (size_t)slot * (2 * (size_t)max_d_cap + 4), lobuf + (size_t)slot * ((size_t)max_d_cap + 1),
                          chbuf + (size_t)slot * (size_t)(max_d_cap + 1) * row_words, max_d_cap, row_words};
    for (uint32_t p = slot; p < n_pairs; p += n_slots) {
        const np2ond::Pair P = pairs[p];
        np2ond::align_pair_wave(pool, P, out_pool, &res[p], W, sq[wave], st[wave]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// one lane per region: which piece every round contributes (the fill rules and the region's fallback counter are sequential
// over the rounds) and how long it is
__global__ void k2_ond_pieces(const OndRegion* __restrict__ regs, uint32_t n_regs, const uint32_t* __restrict__ cand_len,
                              const int32_t* __restrict__ pair_of, const np2ond::PairResult* __restrict__ res,
                              uint32_t* __restrict__ piece_len, uint8_t* __restrict__ piece_kind) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_regs) return;
    const OndRegion R = regs[j];
    const int32_t seed_len = (int32_t)R.seed_len;
    int32_t lqcount = 0;
    const int32_t lq0_len = (int32_t)cand_len[R.first_cand];
    for (int i = 0; i < LQ_ROUNDS; ++i) {
        const bool beyond = (uint32_t)i >= R.n_cand;
        if (beyond) lqcount = 0;
        bool fallback = true;
        uint32_t len = 0, kind = PIECE_SEED;
        const int32_t pi = pair_of[(size_t)j * LQ_ROUNDS + i];
        if (pi >= 0) {
            const np2ond::PairResult r = res[pi];
            if (r.aln_len > 2) {
                const int32_t cl = (int32_t)cand_len[R.first_cand + (uint32_t)i];
                int32_t tail_t = seed_len - r.aln_t_len; if (tail_t < 0) tail_t = 0;
                int32_t tail_q = cl - r.aln_q_len; if (tail_q < 0) tail_q = 0; if (tail_q > 250) tail_q = 250;
                len = (uint32_t)(r.aln_len + tail_t + tail_q);
                kind = PIECE_ALN;
                fallback = false;
            }
        }
        if (fallback) {
            if (lqcount++ < (int32_t)R.n_cand - 1) { kind = PIECE_SEED; len = (uint32_t)seed_len; }
            else { kind = PIECE_LQSEQ; len = (uint32_t)lq0_len; }
        }
        piece_len[(size_t)i * n_regs + j] = len;
        piece_kind[(size_t)i * n_regs + j] = (uint8_t)kind;
    }
}

// one workgroup per round: position of every region's piece in the round's string ('N' + piece per region, a final 'N')
__global__ __launch_bounds__(256) void k2_ond_scan(const uint32_t* __restrict__ piece_len, uint32_t n_regs, uint32_t* __restrict__ piece_pos,
                                                   uint32_t* __restrict__ total) {
    __shared__ uint32_t sh[256];
    __shared__ uint32_t carry;
    const uint32_t i = blockIdx.x;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_regs; base += 256) {
        const uint32_t j = base + threadIdx.x;
        const uint32_t v = j < n_regs ? piece_len[(size_t)i * n_regs + j] + 1u : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 256; o <<= 1) {
            const uint32_t a = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += a;
            __syncthreads();
        }
        if (j < n_regs) piece_pos[(size_t)i * n_regs + j] = carry + sh[threadIdx.x] - v + 1u;   // after this region's 'N'
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[i] = carry + 1u;   // + the closing 'N'
}

// one wave per (round, region) piece: 'N' + the gapped characters into the round's t and q strings
__global__ __launch_bounds__(256) void k2_ond_emit(const OndRegion* __restrict__ regs, uint32_t n_regs, const uint8_t* __restrict__ pool,
                                                   const uint64_t* __restrict__ cand_off, const uint32_t* __restrict__ cand_len,
                                                   const int32_t* __restrict__ pair_of, const np2ond::Pair* __restrict__ pairs,
                                                   const np2ond::PairResult* __restrict__ res, const uint8_t* __restrict__ out_pool,
                                                   const uint32_t* __restrict__ piece_len, const uint8_t* __restrict__ piece_kind,
                                                   const uint32_t* __restrict__ piece_pos, const uint32_t* __restrict__ total,
                                                   const uint64_t* __restrict__ str_off, char* __restrict__ strpool) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (w >= (uint32_t)LQ_ROUNDS * n_regs) return;
    const uint32_t i = w / n_regs, j = w % n_regs;
    const OndRegion R = regs[j];
    char* T = strpool + str_off[2 * i];
    char* Q = strpool + str_off[2 * i + 1];
    const uint32_t pos = piece_pos[(size_t)i * n_regs + j], len = piece_len[(size_t)i * n_regs + j], kind = piece_kind[(size_t)i * n_regs + j];
    if (lane == 0) {
        T[pos - 1] = 'N'; Q[pos - 1] = 'N';
        if (j == n_regs - 1) { const uint32_t e = total[i]; T[e - 1] = 'N'; Q[e - 1] = 'N'; T[e] = 0; Q[e] = 0; }
    }
    const uint8_t* seed = pool + R.seed_off;
    if (kind == PIECE_SEED) {
        for (uint32_t p = lane; p < len; p += 64) { T[pos + p] = 'M'; Q[pos + p] = 'M'; }
    } else if (kind == PIECE_LQSEQ) {
        const uint8_t* lq0 = pool + cand_off[R.first_cand];
        for (uint32_t p = lane; p < len; p += 64) { T[pos + p] = p < R.seed_len ? (char)seed[p] : '-'; Q[pos + p] = (char)lq0[p]; }
    } else {
        const int32_t pi = pair_of[(size_t)j * LQ_ROUNDS + i];
        const np2ond::PairResult r = res[pi];
        const np2ond::Pair P = pairs[pi];
        const uint8_t* rt = out_pool + P.out_off;       // traceback order: column c of the alignment is at aln_len - 1 - c
        const uint8_t* rq = rt + P.out_cap;
        const uint8_t* cand = pool + cand_off[R.first_cand + i];
        const uint32_t a = (uint32_t)r.aln_len;
        const uint32_t tail_t = (int32_t)R.seed_len > r.aln_t_len ? R.seed_len - (uint32_t)r.aln_t_len : 0u;
        for (uint32_t p = lane; p < len; p += 64) {
            char tc, qc;
            if (p < a) { tc = (char)rt[a - 1 - p]; qc = (char)rq[a - 1 - p]; }
            else if (p < a + tail_t) { tc = (char)seed[(uint32_t)r.aln_t_len + (p - a)]; qc = '-'; }
            else { tc = '-'; qc = (char)cand[(uint32_t)r.aln_q_len + (p - a - tail_t)]; }
            T[pos + p] = tc; Q[pos + p] = qc;
        }
    }
}

// NP2_TIMING=1: per-stage wall time (with stream syncs) of every window on stderr
struct StageClock {
    bool on;
    hipStream_t q;
    double t0;
    std::string line;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    StageClock(hipStream_t s) : on(getenv("NP2_TIMING") != nullptr), q(s), t0(0) { if (on) t0 = now(); }
    void mark(const char* name) {
        if (!on) return;
        (void)hipStreamSynchronize(q);
        const double t = now();
        char b[64];
        snprintf(b, sizeof(b), " %s %.2f", name, t - t0);
        line += b;
        t0 = t;
    }
    void flush(const char* what, long cols, long streams, long entries) {
        if (on) fprintf(stderr, "[np2 %s] cols %ld streams %ld entries %ld | ms:%s\n", what, cols, streams, entries, line.c_str());
    }
};

int pick_device(std::string* err) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { *err = "no HIP device (the long-read consensus has no CPU fallback)"; return -1; }
    const char* e = getenv("NP2_DEVICE");
    int d = e ? atoi(e) : (int)(getpid() % n);
    if (d < 0 || d >= n) d = 0;
    return d;
}

class HipExec : public Exec {
  public:
    explicit HipExec(int device) : device_(device) {}
    ~HipExec() override {
        if (stream_) (void)hipStreamSynchronize(stream_);
        if (stream2_) { (void)hipStreamSynchronize(stream2_); (void)npalloc::stream_destroy(stream2_); }
        if (ev_up_) (void)hipEventDestroy(ev_up_);
        if (ev_big_) (void)hipEventDestroy(ev_big_);
        if (stream_) (void)npalloc::stream_destroy(stream_);
    }
    bool init(std::string* err) {
        HIPOK(hipSetDevice(device_));
        // workers share the host cores of a GPU (the reference's -p model): waiting for the GPU must not spin on one
        if (!getenv("NP2_SPIN_SYNC")) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        HIPOK(npalloc::stream_create(&stream_));
        np::bgzf_device_inflate_enable(device_);   // the worker's BAM readers inflate their windows on the device from now on
        return true;
    }
    bool compute_spans(const WindowInput& in, int set, std::vector<SpanOut>* spans, std::string* err) override;
    bool run_window(const WindowInput& in, WindowOutput* out, std::string* err) override;
    bool run_lq(const LqInput& in, std::string* cons_rev, std::string* err) override;
    bool run_lq_aligned(const LqAlignInput& in, std::string* cons_rev, std::string* err) override;
    bool extract(const std::vector<SubReq>& req, std::vector<uint32_t>* off, std::string* bases, std::string* err) override;
    bool run_poa(const PoaBatch& in, std::vector<std::string>* out, std::string* err) override;
    bool read_coords(const std::vector<CoordReq>& req, std::vector<uint32_t>* bases, std::string* err) override;

  private:
    bool lq_from_pool(const std::vector<uint32_t>& str_len, uint32_t t_len, uint32_t gap_min_len, bool hifi, std::string* cons_rev, std::string* err);
    // link observations -> column buckets -> nodes/entries for n_streams tag streams over n_cols columns; *total = entries
    // m_seen != nullptr: colcnt_ already holds the tags per column (valid unless *m_seen, a device flag, is set)
    int build_graph_tiles(const std::vector<uint32_t>& n_tags, uint32_t n_chunks, uint32_t n_cols, uint32_t* total, std::string* err, struct StageClock* clk);
    bool build_graph(const std::vector<uint32_t>& n_tags, uint32_t n_cols, uint32_t* total, std::string* err, struct StageClock* clk = nullptr,
                     const uint32_t* m_seen = nullptr);
    bool solve(const MsaView& mv, int32_t l, uint32_t n_cols, uint32_t total, int rule, uint32_t* cons_len, struct StageClock* clk, std::string* err);
    template <uint32_t K>
    bool solve_runs(const MsaView& mv, int32_t l, uint32_t n_cols, uint32_t total, int rule, uint32_t n_cuts, uint32_t n_runs, uint32_t* cons_len,
                    struct StageClock* clk, std::string* err);
    int device_;
    hipStream_t stream_ = nullptr;
    bool upload_contig(const WindowInput& in, std::string* err);
    bool upload_set(const RecordSet& rs, int set, std::string* err);
    DevRecs dev_set(int set) const {
        const DevBuf* b = rb_[set];
        return DevRecs{b[0].as<int32_t>(), b[1].as<uint32_t>(), b[2].as<uint32_t>(), b[3].as<uint64_t>(), b[4].as<uint64_t>(), b[5].as<uint32_t>(), b[6].as<uint8_t>()};
    }
    DevBuf rb_[2][7];   // pos, n_cigar, q0, cigar_off, seq_off, cigar, seq of the two record sets
    DevBuf contig_, spans_, sd_, tags_, tagoff_, alnts_, te_, cnt4_, stat_, colcnt_,
        coloff_, cursor_, sums_, obs_, entries_, nodes_, colnn_, res_, cons_, strpool_, stroff_, strlen_, flag_, tchunks_, tckpt_, tchoff_, chunks_, chcnt_, chpre_, sums2_, cutflag_, cutpos_, cuts_, eav_, runt_, grpt_, grpx_, grpn_, btwalk_, btpick_, btgrp_, btgpick_;
    DevBuf runsz_, runoff_, live_, ematch_, xreq_, xfirst_, xlen_, xoff_, xout_, obscol_, obsaux_, grpt2_, grpx2_, grpn2_, covdiff_, covpre_;
    DevBuf tilecnt_, tileoff_, tilecur_, tilelist_, tilectr_, ntags_, colne_, runflag_, runlist_, runctr_, tileredo_;
    DevBuf poapool_, poaoff_, poalen_, poajobs_, poatabs_, poatabf_, poaout_, poaolen_, poastat_, poadbg_, poaord_, trig_;
    hipStream_t stream2_ = nullptr;      // the Big class of the pseudo-seed jobs runs beside the Small one (run_poa)
    hipEvent_t ev_up_ = nullptr, ev_big_ = nullptr;
    bool graph_compact_ = false;   // the graph in HBM was built by tiles: columns are contiguous, every entry slot is live
    DevBuf ondpool_, ondregs_, ondcoff_, ondclen_, ondpairof_, ondpairs_, ondres_, ondout_, ondv_, ondlo_, ondch_, ondplen_, ondpkind_, ondppos_, ondtot_;
    PinBuf pin_;
    std::vector<uint32_t> win_first_chunk_, win_n_chunks_;   // chunk range of every stream of the last run_window
    bool win_tags_live_ = false;
    uint64_t contig_serial_ = ~0ull;
    size_t contig_len_ = 0;
};

bool HipExec::upload_contig(const WindowInput& in, std::string* err) {
    // contig characters: uploaded once per contig (the pipeline passes the same serial for every window of it)
    const size_t clen = strlen(in.contig_seq);
    if (contig_serial_ != in.contig_serial || contig_len_ != clen) {
        if (!contig_.ensure(clen + 16)) { *err = "out of device memory (contig)"; return false; }
        HIPOK(npcopy::h2d(contig_.p, in.contig_seq, clen + 1, stream_));
        contig_serial_ = in.contig_serial;
        contig_len_ = clen;
    }
    return true;
}

bool HipExec::upload_set(const RecordSet& rs, int set, std::string* err) {
    hipStream_t q = stream_;
    const size_t n = rs.size();
    DevBuf* b = rb_[set];
    auto up = [&](DevBuf& d, const void* src, size_t bytes) -> bool {
        if (!d.ensure(bytes + 16)) return false;
        return bytes == 0 || npcopy::h2d(d.p, src, bytes, q) == hipSuccess;
    };
    if (!up(b[0], rs.pos.data(), 4 * n) || !up(b[1], rs.n_cigar.data(), 4 * n) || !up(b[2], rs.q0.data(), 4 * n) ||
        !up(b[3], rs.cigar_off.data(), 8 * n) || !up(b[4], rs.seq_off.data(), 8 * n) || !up(b[5], rs.cigar.data(), 4 * rs.cigar.size()) ||
        !up(b[6], rs.seq.data(), rs.seq.size())) { *err = "out of device memory (records)"; return false; }
    return true;
}

// spans of a record set; the set stays in HBM for the run_window call of the same window
bool HipExec::compute_spans(const WindowInput& in, int set, std::vector<SpanOut>* spans, std::string* err) {
    HIPOK(hipSetDevice(device_));
    hipStream_t q = stream_;
    const RecordSet& rs = set ? in.sup : in.recs;
    const uint32_t n = (uint32_t)rs.size();
    spans->assign(n, SpanOut{0, 0, 0, 0, 0, 0});
    if (!upload_contig(in, err) || !upload_set(rs, set, err)) return false;
    if (n) {
        if (!spans_.ensure(sizeof(SpanOut) * (size_t)n)) { *err = "out of device memory (spans)"; return false; }
        k2_span<<<nblk(n, SERIAL_LANES), 64, 0, q>>>(dev_set(set), n, contig_.as<char>(), in.s, in.e, spans_.as<SpanOut>());
        HIPOK(npcopy::d2h(spans->data(), spans_.p, sizeof(SpanOut) * (size_t)n, q));
    }
    HIPOK(hipStreamSynchronize(q));
    return true;
}

// the record sets of `in` were uploaded by the compute_spans calls of this window
bool HipExec::run_window(const WindowInput& in, WindowOutput* out, std::string* err) {
    HIPOK(hipSetDevice(device_));
    hipStream_t q = stream_;
    const int32_t s = in.s, e = in.e, l = e - s;
    StageClock clk(q);
    if (!upload_contig(in, err)) return false;
    // ---- stream layout: seed + the given streams
    std::vector<StreamDesc> sd;
    out->tag_off.clear(); out->aln_t_s.clear(); out->aln_t_e.clear();
    uint64_t tag_bytes = 0;
    out->tag_off.push_back(0);
    out->aln_t_s.push_back(0);
    tag_bytes += ((uint64_t)l + 1) / 2 + 1;
    tag_bytes = (tag_bytes + 3) & ~3ull;
    for (const StreamRef& sr : in.streams) {
        StreamDesc d;
        d.set = sr.set; d.read = sr.rec; d.col0 = sr.span.col0; d.aln_len = sr.span.aln_len; d.aln_t_s = sr.span.aln_t_s; d.pad = 0; d.tag_off = tag_bytes;
        sd.push_back(d);
        out->tag_off.push_back(tag_bytes);
        out->aln_t_s.push_back(sr.span.aln_t_s - (uint32_t)s);
        tag_bytes += ((uint64_t)sr.span.aln_len + 1) / 2 + 1;
        tag_bytes = (tag_bytes + 3) & ~3ull;
    }
    const uint32_t n_streams = (uint32_t)out->tag_off.size();
    out->seq_count = n_streams;
    const uint32_t n_cols = (uint32_t)l + 1;
    if (!tags_.ensure(tag_bytes + 16) || !cnt4_.ensure(16ull * n_cols + 64) || !stat_.ensure(sizeof(ColStat) * (size_t)n_cols) ||
        !tagoff_.ensure(8ull * n_streams) || !alnts_.ensure(4ull * n_streams) || !te_.ensure(4ull * n_streams + 16) ||
        !sd_.ensure(sizeof(StreamDesc) * sd.size() + 16) || !colcnt_.ensure(4ull * (n_cols + 2)) || !coloff_.ensure(4ull * (n_cols + 2)) ||
        !cursor_.ensure(4ull * (n_cols + 2)) || !sums_.ensure(4ull * (nblk(n_cols + 1, SCAN_TILE) + 2)) || !colnn_.ensure(4ull * n_cols) ||
        !res_.ensure(sizeof(DpResult))) { *err = "out of device memory (window)"; return false; }
    HIPOK(hipMemsetAsync(tags_.p, 0, tag_bytes + 16, q));
    HIPOK(hipMemsetAsync(cnt4_.p, 0, 16ull * n_cols + 64, q));
    HIPOK(hipMemsetAsync(colcnt_.p, 0, 4ull * (n_cols + 2), q));
    HIPOK(hipMemsetAsync(cursor_.p, 0, 4ull * (n_cols + 2), q));
    DevStat st{cnt4_.as<uint32_t>(), cnt4_.as<uint32_t>() + n_cols, cnt4_.as<uint32_t>() + 2ull * n_cols, cnt4_.as<uint32_t>() + 3ull * n_cols};
    if (!covdiff_.ensure(4ull * (n_cols + 4)) || !covpre_.ensure(4ull * (n_cols + 4))) { *err = "out of device memory (window)"; return false; }
    HIPOK(hipMemsetAsync(covdiff_.p, 0, 4ull * (n_cols + 4), q));
    k2_seed_tags<<<nblk(((uint64_t)l + 1) / 2 + 1, 256), 256, 0, q>>>(contig_.as<char>(), s, (uint32_t)l, tags_.as<uint8_t>());
    if (!sd.empty()) {
        HIPOK(npcopy::h2d(sd_.p, sd.data(), sizeof(StreamDesc) * sd.size(), q));
        // chunk list (host, O(records)); empty streams still need their terminator and end position
        std::vector<TagChunk> tcs;
        std::vector<uint32_t> choff(sd.size());
        for (uint32_t k = 0; k < (uint32_t)sd.size(); ++k) {
            choff[k] = (uint32_t)tcs.size();
            const uint32_t nch = (sd[k].aln_len + TAG_CHUNK - 1) / TAG_CHUNK;
            for (uint32_t c = 0; c < nch; ++c)
                tcs.push_back(TagChunk{k, c * TAG_CHUNK, std::min(TAG_CHUNK, sd[k].aln_len - c * TAG_CHUNK), c + 1 == nch ? 1u : 0u});
        }
        const uint32_t n_tchunks = (uint32_t)tcs.size();
        if (!tchunks_.ensure(sizeof(TagChunk) * (size_t)n_tchunks + 64) || !tckpt_.ensure(sizeof(TagCkpt) * (size_t)n_tchunks + 64) ||
            !tchoff_.ensure(4ull * sd.size() + 64)) { *err = "out of device memory (tag chunks)"; return false; }
        // streams without columns (aln_len 0): terminator byte 0xff and end = start, written from the host
        for (uint32_t k = 0; k < (uint32_t)sd.size(); ++k)
            if (sd[k].aln_len == 0) {
                const uint8_t ff = 0xff;
                const uint32_t te = sd[k].aln_t_s - (uint32_t)s;
                HIPOK(npcopy::h2d(tags_.as<uint8_t>() + sd[k].tag_off, &ff, 1, q));
                HIPOK(npcopy::h2d(te_.as<uint32_t>() + 1 + k, &te, 4, q));
            }
        if (n_tchunks) {
            HIPOK(npcopy::h2d(tchunks_.p, tcs.data(), sizeof(TagChunk) * (size_t)n_tchunks, q));
            HIPOK(npcopy::h2d(tchoff_.p, choff.data(), 4ull * sd.size(), q));
            k2_tag_ckpt<<<nblk(sd.size(), SERIAL_LANES), 64, 0, q>>>(sd_.as<StreamDesc>(), (uint32_t)sd.size(), tchoff_.as<uint32_t>(), dev_set(0), dev_set(1), s,
                                                           tckpt_.as<TagCkpt>());
            k2_tags_chunk<<<nblk(n_tchunks, 64), 64, 0, q>>>(tchunks_.as<TagChunk>(), n_tchunks, tckpt_.as<TagCkpt>(), sd_.as<StreamDesc>(), dev_set(0), dev_set(1),
                                                             in.gap_min_len, tags_.as<uint8_t>(), st, covdiff_.as<uint32_t>(), covdiff_.as<uint32_t>() + n_cols + 3,
                                                             te_.as<uint32_t>() + 1);
        }
        k2_cov_diff<<<nblk(sd.size(), 256), 256, 0, q>>>(sd_.as<StreamDesc>(), (uint32_t)sd.size(), s, te_.as<uint32_t>() + 1, covdiff_.as<uint32_t>());
    }
    {
        const uint32_t nsb = nblk(n_cols + 1, SCAN_TILE);
        k2_scan_sums<<<nsb, SCAN_T, 0, q>>>(covdiff_.as<uint32_t>(), n_cols + 1, sums_.as<uint32_t>());
        k2_scan_top<<<1, 1, 0, q>>>(sums_.as<uint32_t>(), nsb);
        k2_scan_final<<<nsb, SCAN_T, 0, q>>>(covdiff_.as<uint32_t>(), n_cols + 1, sums_.as<uint32_t>(), covpre_.as<uint32_t>());
    }
    k2_pack_stat<<<nblk(n_cols, 256), 256, 0, q>>>(st.coverage, st.max_size, st.l_ins, st.l_del, n_cols, stat_.as<ColStat>(), (uint32_t)l, covpre_.as<uint32_t>(),
                                                    covdiff_.as<uint32_t>(), colcnt_.as<uint32_t>());
    HIPOK(npcopy::h2d(tagoff_.p, out->tag_off.data(), 8ull * n_streams, q));
    HIPOK(npcopy::h2d(alnts_.p, out->aln_t_s.data(), 4ull * n_streams, q));
    clk.mark("tags");
    uint32_t total = 0;
    std::vector<uint32_t> n_tags;
    n_tags.push_back((uint32_t)l);
    for (const StreamDesc& d : sd) n_tags.push_back(d.aln_len);
    if (!build_graph(n_tags, n_cols, &total, err, &clk, covdiff_.as<uint32_t>() + n_cols + 3)) return false;
    if (!cons_.ensure(sizeof(ConsBase) * ((size_t)total + 16))) { *err = "out of device memory (consensus)"; return false; }
    clk.mark("build");
    // ---- chain DP + backtrace
    MsaView mv{coloff_.as<uint32_t>(), colnn_.as<uint32_t>(), nodes_.as<Node>(), entries_.as<Entry>(), stat_.as<ColStat>(),
               getenv("NP2_NO_MATCH") ? nullptr : ematch_.as<EMatch>()};
    uint32_t cons_len = 0;
    if (!solve(mv, l, n_cols, total, in.read_type, &cons_len, &clk, err)) return false;
    out->cons.resize(cons_len);
    out->stat.resize(n_cols);
    out->tags.resize(in.want_tags ? tag_bytes : 0);
    out->aln_t_e.assign(n_streams, 0);
    // consensus + column statistics (8 B per draft base each) come back through a pinned staging buffer (DMA at link
    // speed) and are copied out by the host thread pool; pageable targets would cost ~4x the time on one core
    const size_t cons_bytes = sizeof(ConsBase) * (size_t)cons_len, stat_bytes = sizeof(ColStat) * (size_t)n_cols;
    static const bool lq_triggers = !(getenv("NP2_LQ_TRIGGERS") && getenv("NP2_LQ_TRIGGERS")[0] == '0');
    const size_t trig_words = (in.lq_ratio1 > 0.f && lq_triggers && cons_len) ? ((size_t)cons_len + 63) / 64 : 0;
    out->trig_del.assign(trig_words, 0);
    out->trig_ins.assign(trig_words, 0);
    if (trig_words) {      // the loop heads of the low-quality scans, evaluated for every consensus base where the data lies
        if (!trig_.ensure(16 * trig_words + 64)) { *err = "out of device memory (low-quality triggers)"; return false; }
        k2_lq_triggers<<<nblk((cons_len + 63u) & ~63u, 256), 256, 0, q>>>(cons_.as<ConsBase>(), cons_len, stat_.as<ColStat>(), in.lq_ratio1,
                                                                          trig_.as<unsigned long long>(), trig_.as<unsigned long long>() + trig_words);
        HIPOK(npcopy::d2h(out->trig_del.data(), trig_.p, 8 * trig_words, q));
        HIPOK(npcopy::d2h(out->trig_ins.data(), trig_.as<unsigned long long>() + trig_words, 8 * trig_words, q));
    }
    if (!pin_.ensure(cons_bytes + stat_bytes + 64)) { *err = "out of pinned host memory (window download)"; return false; }
    uint8_t* pin = static_cast<uint8_t*>(pin_.p);
    if (cons_len) HIPOK(npcopy::d2h(pin, cons_.p, cons_bytes, q));
    HIPOK(npcopy::d2h(pin + cons_bytes, stat_.p, stat_bytes, q));
    if (in.want_tags) HIPOK(npcopy::d2h(out->tags.data(), tags_.p, tag_bytes, q));
    if (n_streams > 1) HIPOK(npcopy::d2h(out->aln_t_e.data() + 1, te_.as<uint32_t>() + 1, 4ull * (n_streams - 1), q));
    HIPOK(hipStreamSynchronize(q));
    {
        uint8_t* dst_c = reinterpret_cast<uint8_t*>(out->cons.data());
        uint8_t* dst_s = reinterpret_cast<uint8_t*>(out->stat.data());
        const size_t total_b = cons_bytes + stat_bytes, blk = 1u << 20;
        np::parallel_for((total_b + blk - 1) / blk, 1, [&](size_t lo, size_t hi) {
            for (size_t b = lo; b < hi; ++b) {
                size_t o = b * blk, e = std::min(total_b, o + blk);
                if (o < cons_bytes) { const size_t e1 = std::min(e, cons_bytes); memcpy(dst_c + o, pin + o, e1 - o); o = e1; }
                if (o < e) memcpy(dst_s + (o - cons_bytes), pin + o, e - o);
            }
        });
    }
    clk.mark("download");
    clk.flush("window", l, n_streams, total);
    out->aln_t_e[0] = (uint32_t)l;
    win_tags_live_ = true;
    return true;
}

// Chain DP + backtrace over a built link graph.  rule: READS_ONT..READS_RS for a window, RULE_LQ / RULE_LQ_HIFI for the
// concatenated low-quality regions.  Leaves the consensus in cons_ in FORWARD order (ConsBase items, or characters
// for the LQ rules) and returns its length.
bool HipExec::solve(const MsaView& mv, int32_t l, uint32_t n_cols, uint32_t total, int rule, uint32_t* cons_len, StageClock* clk,
                    std::string* err) {
    hipStream_t q = stream_;
    if (!cutflag_.ensure(4ull * (n_cols + CUT_BLOCK + 2)) || !cutpos_.ensure(4ull * (n_cols + 2)) || !cuts_.ensure(4ull * (n_cols + 2)) ||
        !flag_.ensure(32) || !cons_.ensure(sizeof(ConsBase) * ((size_t)total + 16))) {
        *err = "out of device memory (dp runs)";
        return false;
    }
    if (clk) clk->mark("dp.alloc");
    // ---- the cuts: narrow columns (<= 8 live entries) where every block of 32 columns has one; deep pileups rarely do,
    // and runs that span many blocks are serial, so the cut width goes up to 32 when more than ~a third of the blocks
    // come up empty
    uint32_t K = getenv("NP2_CUT_K") ? (uint32_t)atoi(getenv("NP2_CUT_K")) : CUT_K_SMALL;
    if (K != CUT_K_SMALL && K != CUT_K_LARGE) K = CUT_K_SMALL;
    uint32_t n_cuts = 0;
    const uint32_t nsb2 = nblk(n_cols + 1, SCAN_TILE);
    for (;;) {
        k2_cut_flags<<<nblk(n_cols / CUT_BLOCK + 1, 64), 64, 0, q>>>(mv, n_cols, l, cutflag_.as<uint32_t>(), K);
        k2_scan_sums<<<nsb2, SCAN_T, 0, q>>>(cutflag_.as<uint32_t>(), n_cols + 1, sums_.as<uint32_t>());
        k2_scan_top<<<1, 1, 0, q>>>(sums_.as<uint32_t>(), nsb2);
        k2_scan_final<<<nsb2, SCAN_T, 0, q>>>(cutflag_.as<uint32_t>(), n_cols + 1, sums_.as<uint32_t>(), cutpos_.as<uint32_t>());
        HIPOK(npcopy::d2h(&n_cuts, cutpos_.as<uint32_t>() + n_cols, 4, q));
        HIPOK(hipStreamSynchronize(q));
        if (K == CUT_K_LARGE || (uint64_t)n_cuts * 3 >= (uint64_t)(n_cols / CUT_BLOCK) * 2 || getenv("NP2_CUT_K")) break;
        K = CUT_K_LARGE;
    }
    if (clk) clk->mark("dp.cutscan");
    k2_cut_list<<<nblk(n_cols, 256), 256, 0, q>>>(cutflag_.as<uint32_t>(), cutpos_.as<uint32_t>(), n_cols, cuts_.as<uint32_t>());
    uint32_t last_cut = 0xffffffffu;
    if (n_cuts) HIPOK(npcopy::d2h(&last_cut, cuts_.as<uint32_t>() + (n_cuts - 1), 4, q));
    HIPOK(hipStreamSynchronize(q));
    if (clk) clk->mark("dp.cutlist");
    // a window ending on a cut column has no open run behind it
    const uint32_t n_runs = (n_cuts && (int32_t)last_cut == l - 1) ? n_cuts : n_cuts + 1;
    if (clk && clk->on) {
        std::vector<uint32_t> hc(n_cuts);
        if (n_cuts) (void)npcopy::d2h_sync(hc.data(), cuts_.p, 4ull * n_cuts);
        uint32_t mx = n_cuts ? hc[0] + 1 : (uint32_t)l, over1k = 0;
        for (uint32_t i = 1; i < n_cuts; ++i) { mx = std::max(mx, hc[i] - hc[i - 1]); over1k += hc[i] - hc[i - 1] > 1000; }
        fprintf(stderr, "[np2 dp] cut width %u: %u cuts, %u runs over %d columns; longest run %u columns, %u runs > 1000\n", K, n_cuts, n_runs, l, mx, over1k);
    }
    return K == CUT_K_LARGE ? solve_runs<CUT_K_LARGE>(mv, l, n_cols, total, rule, n_cuts, n_runs, cons_len, clk, err)
                            : solve_runs<CUT_K_SMALL>(mv, l, n_cols, total, rule, n_cuts, n_runs, cons_len, clk, err);
}

template <uint32_t K>
bool HipExec::solve_runs(const MsaView& mv, int32_t l, uint32_t n_cols, uint32_t total, int rule, uint32_t n_cuts, uint32_t n_runs, uint32_t* cons_len,
                         StageClock* clk, std::string* err) {
    hipStream_t q = stream_;
    (void)n_cols; (void)total;
    const bool lq = rule == RULE_LQ || rule == RULE_LQ_HIFI;
    typedef RunT<K> RT;
    if (!runt_.ensure(sizeof(RT) * (size_t)n_runs + 64) || !btwalk_.ensure(sizeof(BtWalk) * K * (size_t)n_runs + 64) ||
        !btpick_.ensure(sizeof(BtPick) * (size_t)n_runs + 64) || !runsz_.ensure(4ull * (n_runs + 2)) || !runoff_.ensure(4ull * (n_runs + 2)) ||
        !sums2_.ensure(4ull * (nblk(n_runs + 1, SCAN_TILE) + 2))) { *err = "out of device memory (dp runs)"; return false; }
    HIPOK(hipMemsetAsync(flag_.p, 0, 32, q));
    HIPOK(hipMemsetAsync(res_.p, 0, sizeof(DpResult), q));
    uint32_t* status = flag_.as<uint32_t>() + 4;
    uint32_t* total_dev = flag_.as<uint32_t>() + 5;
    // ring of two columns of coefficient vectors per run
    k2_run_size<<<nblk(n_runs + 1, 256), 256, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, n_runs, l, runsz_.as<uint32_t>(), graph_compact_ ? colne_.as<uint32_t>() : nullptr);
    {
        const uint32_t nsr = nblk(n_runs + 1, SCAN_TILE);
        k2_scan_sums<<<nsr, SCAN_T, 0, q>>>(runsz_.as<uint32_t>(), n_runs + 1, sums2_.as<uint32_t>());
        k2_scan_top<<<1, 1, 0, q>>>(sums2_.as<uint32_t>(), nsr);
        k2_scan_final<<<nsr, SCAN_T, 0, q>>>(runsz_.as<uint32_t>(), n_runs + 1, sums2_.as<uint32_t>(), runoff_.as<uint32_t>());
    }
    uint32_t ring_units = 0;
    HIPOK(npcopy::d2h(&ring_units, runoff_.as<uint32_t>() + n_runs, 4, q));
    HIPOK(hipStreamSynchronize(q));
    if (!eav_.ensure(8ull * (K + 1) * (size_t)ring_units + 64)) { *err = "out of device memory (dp runs)"; return false; }
    if (clk) clk->mark("dp.cuts");
    const long long C = rule == RULE_LQ ? 2 : (rule == READS_HIFI || rule == RULE_LQ_HIFI) ? 4 : 3;
    const bool dense = n_runs >= 16384 && !getenv("NP2_RUN_PER_WAVE");   // enough runs to fill the chip with several per wave
    uint32_t rpw_ac = dense ? 64u / (K + 1) : 1u, rpw_dp = dense ? 64u : 1u;
    if (getenv("NP2_RPW_DP")) rpw_dp = std::min(64u, std::max(1u, (uint32_t)atoi(getenv("NP2_RPW_DP"))));   // tuning experiments
    if (getenv("NP2_RPW_AC")) rpw_ac = std::min(64u / (K + 1), std::max(1u, (uint32_t)atoi(getenv("NP2_RPW_AC"))));
    // graphs built by tiles: one wave per run, columns staged in LDS; the runs those kernels flag take the kernels that walk HBM
    const bool staged = graph_compact_ && mv.match && !getenv("NP2_RUN_GLOBAL");
    const uint8_t* only = nullptr;
    const uint32_t* list = nullptr;
    const uint32_t* n_list = nullptr;
    if (staged) {
        if (!runflag_.ensure((size_t)n_runs + 64) || !runlist_.ensure(4ull * n_runs + 64) || !runctr_.ensure(64)) { *err = "out of device memory (dp runs)"; return false; }
        only = runflag_.as<uint8_t>();
        list = runlist_.as<uint32_t>();
        n_list = runctr_.as<uint32_t>();
        HIPOK(hipMemsetAsync(runctr_.p, 0, 64, q));
        k2_run_ac_lds<K><<<n_runs, 64, 0, q>>>(mv, colne_.as<uint32_t>(), cuts_.as<uint32_t>(), n_cuts, n_runs, l, C, runt_.as<RT>(), runflag_.as<uint8_t>(),
                                                runlist_.as<uint32_t>(), runctr_.as<uint32_t>(), status);
        if (clk && clk->on) {
            uint32_t c[3] = {0, 0, 0};
            (void)npcopy::d2h(c, runctr_.p, 12, q);
            (void)hipStreamSynchronize(q);
            fprintf(stderr, "[np2 dp] %u of %u runs left to the kernels that walk HBM (a column over %u entries)\n", c[0], n_runs, RL_COLMAX);
        }
    }
    k2_run_ac<K><<<nblk(n_runs, rpw_ac), 64, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, n_runs, l, C, eav_.as<long long>(), runoff_.as<uint32_t>(), runt_.as<RT>(),
                                                      rpw_ac, status, list, n_list);
    if (clk) clk->mark("dp.ac");
    if (n_cuts) {
        const uint32_t n_groups = nblk(n_cuts, SCAN_G);
        if (!grpt_.ensure(sizeof(RT) * (size_t)n_groups + 64) || !grpx_.ensure(8ull * K * n_groups + 64) || !grpn_.ensure(4ull * n_groups + 64)) {
            *err = "out of device memory (dp scan)";
            return false;
        }
        k2_scan_groups<K><<<n_groups, 64, 0, q>>>(runt_.as<RT>(), n_cuts, grpt_.as<RT>());
        if (n_groups > 2 * SCAN_G) {   // second level: groups of groups, one short chain on top
            const uint32_t n_super = nblk(n_groups, SCAN_G);
            if (!grpt2_.ensure(sizeof(RT) * (size_t)n_super + 64) || !grpx2_.ensure(8ull * K * n_super + 64) || !grpn2_.ensure(4ull * n_super + 64)) {
                *err = "out of device memory (dp scan)";
                return false;
            }
            k2_scan_groups<K><<<n_super, 64, 0, q>>>(grpt_.as<RT>(), n_groups, grpt2_.as<RT>());
            k2_scan_chain<K><<<1, 64, 0, q>>>(grpt2_.as<RT>(), n_super, nullptr, nullptr, grpx2_.as<long long>(), grpn2_.as<uint32_t>());
            k2_scan_chain<K><<<n_super, 64, 0, q>>>(grpt_.as<RT>(), n_groups, grpx2_.as<long long>(), grpn2_.as<uint32_t>(), grpx_.as<long long>(), grpn_.as<uint32_t>());
        } else {
            k2_scan_chain<K><<<1, 64, 0, q>>>(grpt_.as<RT>(), n_groups, nullptr, nullptr, grpx_.as<long long>(), grpn_.as<uint32_t>());
        }
        k2_scan_apply<K><<<n_groups, 64, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, runt_.as<RT>(), grpx_.as<long long>(), grpn_.as<uint32_t>());
    }
    if (clk) clk->mark("dp.scan");
#define NP2_RUN_DP(T)                                                                                                         \
    do {                                                                                                                      \
        if (staged) k2_run_dp_lds<T><<<n_runs, 64, 0, q>>>(mv, colne_.as<uint32_t>(), cuts_.as<uint32_t>(), n_cuts, n_runs, l, res_.as<DpResult>(), only); \
        k2_run_dp_a<T><<<nblk(n_runs, rpw_dp), 64, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, n_runs, l, res_.as<DpResult>(), rpw_dp, list, n_list); \
        if (n_cuts) k2_run_dp_b<T><<<nblk(n_cuts, rpw_dp), 64, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, l, res_.as<DpResult>(), rpw_dp);          \
    } while (0)
    switch (rule) {
        case READS_CLR: NP2_RUN_DP(READS_CLR); break;
        case READS_HIFI: NP2_RUN_DP(READS_HIFI); break;
        case READS_RS: NP2_RUN_DP(READS_RS); break;
        case RULE_LQ: NP2_RUN_DP(RULE_LQ); break;
        case RULE_LQ_HIFI: NP2_RUN_DP(RULE_LQ_HIFI); break;
        default: NP2_RUN_DP(READS_ONT); break;
    }
#undef NP2_RUN_DP
    if (clk) clk->mark("dp.replay");
    // ---- backtrace: walk every start, chain the runs, write
    const uint32_t rpw_bt0 = dense ? 64u / K : 1u, rpw_bt1 = dense ? 64u : 1u;
    if (lq) k2_bt_runs<true, false><<<nblk(n_runs, rpw_bt0), 64, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, n_runs, l, res_.as<DpResult>(), btwalk_.as<BtWalk>(), nullptr, nullptr, nullptr, status, K, rpw_bt0);
    else k2_bt_runs<false, false><<<nblk(n_runs, rpw_bt0), 64, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, n_runs, l, res_.as<DpResult>(), btwalk_.as<BtWalk>(), nullptr, nullptr, nullptr, status, K, rpw_bt0);
    {
        const uint32_t n_bgroups = nblk(n_runs, BT_G);
        if (!btgrp_.ensure(sizeof(BtGroup) * K * (size_t)n_bgroups + 64) || !btgpick_.ensure(sizeof(BtGroupPick) * (size_t)n_bgroups + 64)) {
            *err = "out of device memory (backtrace)";
            return false;
        }
        k2_bt_groups<<<n_bgroups, 64, 0, q>>>(btwalk_.as<BtWalk>(), n_runs, btgrp_.as<BtGroup>(), K);
        k2_bt_chain<<<1, 64, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, n_runs, l, res_.as<DpResult>(), btgrp_.as<BtGroup>(), n_bgroups,
                                     btgpick_.as<BtGroupPick>(), total_dev, status, K);
        k2_bt_place<<<nblk(n_bgroups, 64), 64, 0, q>>>(btwalk_.as<BtWalk>(), n_runs, btgpick_.as<BtGroupPick>(), btpick_.as<BtPick>(), K);
    }
    if (lq) k2_bt_runs<true, true><<<nblk(n_runs, rpw_bt1), 64, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, n_runs, l, res_.as<DpResult>(), nullptr, btpick_.as<BtPick>(), nullptr, cons_.as<char>(), status, K, rpw_bt1);
    else k2_bt_runs<false, true><<<nblk(n_runs, rpw_bt1), 64, 0, q>>>(mv, cuts_.as<uint32_t>(), n_cuts, n_runs, l, res_.as<DpResult>(), nullptr, btpick_.as<BtPick>(), cons_.as<ConsBase>(), nullptr, status, K, rpw_bt1);
    DpResult res;
    uint32_t st2[2] = {0, 0};
    HIPOK(npcopy::d2h(&res, res_.p, sizeof(res), q));
    HIPOK(npcopy::d2h(st2, status, 8, q));
    HIPOK(hipStreamSynchronize(q));
    if (clk) clk->mark("backtrace");
    if (st2[0] == 4) { *err = "a link reaches further back than the previous column"; return false; }
    if (!lq && res.status == 1) { *err = "no alignment column reaches the end of the window"; return false; }
    if (st2[0] == 2) { *err = lq ? "low-quality backtrace left the graph" : "backtrace left the graph"; return false; }
    if (st2[0] == 3) { *err = "zero coverage on the consensus path"; return false; }
    *cons_len = st2[1];
    return true;
}

// The link graph by tiles (k2_tile_graph).  Returns 1 = built, 2 = a tile overflowed (the caller takes the scatter path), 0 = error.
int HipExec::build_graph_tiles(const std::vector<uint32_t>& n_tags, uint32_t n_chunks, uint32_t n_cols, uint32_t* total_out, std::string* err, StageClock* clk) {
    hipStream_t q = stream_;
    const uint32_t n_tiles = (n_cols + 63) / 64, n_streams = (uint32_t)n_tags.size();
    uint64_t list_cap = 0, tag_total = 0;
    for (uint32_t v : n_tags) { list_cap += v / 64 + 2; tag_total += v; }
    const uint32_t nst = nblk(n_tiles + 1, SCAN_TILE);
#define TILEOK(x) do { if ((x) != hipSuccess) { *err = std::string("HIP error in ") + #x; return 0; } } while (0)
    if (!tilecnt_.ensure(4ull * (n_tiles + 2)) || !tileoff_.ensure(4ull * (n_tiles + 2)) || !tilecur_.ensure(4ull * (n_tiles + 2)) ||
        !tilelist_.ensure(sizeof(TileStream) * list_cap + 64) || !tilectr_.ensure(64) || !colne_.ensure(4ull * (n_cols + 2)) || !ntags_.ensure(4ull * n_streams + 16) ||
        !sums_.ensure(4ull * (nst + 2)) || !entries_.ensure(sizeof(Entry) * tag_total + 64) || !nodes_.ensure(sizeof(Node) * tag_total + 64)) {
        *err = "out of device memory (link graph)";
        return 0;
    }
    TILEOK(npcopy::h2d(ntags_.p, n_tags.data(), 4ull * n_streams, q));
    TILEOK(hipMemsetAsync(tilecnt_.p, 0, 4ull * (n_tiles + 2), q));
    TILEOK(hipMemsetAsync(tilecur_.p, 0, 4ull * (n_tiles + 2), q));
    TILEOK(hipMemsetAsync(tilectr_.p, 0, 64, q));
    k2_tile_count<<<nblk(n_chunks, 64), 64, 0, q>>>(chunks_.as<ChunkDesc>(), n_chunks, chpre_.as<uint32_t>(), alnts_.as<uint32_t>(), n_tiles, tilecnt_.as<uint32_t>());
    k2_scan_sums<<<nst, SCAN_T, 0, q>>>(tilecnt_.as<uint32_t>(), n_tiles + 1, sums_.as<uint32_t>());
    k2_scan_top<<<1, 1, 0, q>>>(sums_.as<uint32_t>(), nst);
    k2_scan_final<<<nst, SCAN_T, 0, q>>>(tilecnt_.as<uint32_t>(), n_tiles + 1, sums_.as<uint32_t>(), tileoff_.as<uint32_t>());
    k2_tile_list<<<nblk(n_chunks, 4), 256, 0, q>>>(chunks_.as<ChunkDesc>(), n_chunks, chpre_.as<uint32_t>(), tagoff_.as<uint64_t>(), alnts_.as<uint32_t>(),
                                                    tags_.as<uint8_t>(), n_tiles, tileoff_.as<uint32_t>(), tilecur_.as<uint32_t>(), tilelist_.as<TileStream>());
    if (clk) clk->mark("tiles.list");
    TileArgs A{tags_.as<uint8_t>(), tagoff_.as<uint64_t>(), alnts_.as<uint32_t>(), ntags_.as<uint32_t>(), tileoff_.as<uint32_t>(), tilelist_.as<TileStream>(),
               n_tiles, n_cols, entries_.as<Entry>(), nodes_.as<Node>(), coloff_.as<uint32_t>(), colnn_.as<uint32_t>(), colne_.as<uint32_t>(), tilectr_.as<uint32_t>(),
               nullptr, nullptr};
    // Every tile first with the small pool; the tiles that do not fit it are listed and taken again with the large one (their place
    // in the graph arrays comes from the same atomic counter); a tile too big for that sends the window down the scatter path.
    if (!tileredo_.ensure(4ull * n_tiles + 64)) { *err = "out of device memory (link graph)"; return 0; }
    TileArgs A1 = A, A2 = A;
    A1.redo = tileredo_.as<uint32_t>(); A1.todo = nullptr;
    A2.redo = nullptr; A2.todo = tileredo_.as<uint32_t>();
    if (getenv("NP2_TILE_PROF")) {     // phase clocks of the tile kernel (cycles summed over the tiles)
        unsigned long long* d_prof = nullptr;
        unsigned long long h[12];
        (void)npalloc::dev_malloc((void**)&d_prof, 96);
        (void)hipMemsetAsync(d_prof, 0, 96, q);
        k2_tile_graph<true, 768, 64><<<n_tiles, 64, 0, q>>>(A1, d_prof);
        k2_tile_graph<true, 2048, 128><<<n_tiles, 64, 0, q>>>(A2, d_prof + 6);
        (void)npcopy::d2h(h, d_prof, 96, q);
        (void)hipStreamSynchronize(q);
        (void)npalloc::dev_free(d_prof);
        fprintf(stderr, "[np2 tile prof] %u tiles; cycles per tile: setup %.0f, stream state %.0f, lookup %.0f, insert %.0f, finish %.0f; large pool (all its tiles) %.0f\n", n_tiles,
                (double)h[0] / n_tiles, (double)h[1] / n_tiles, (double)h[2] / n_tiles, (double)h[3] / n_tiles, (double)h[4] / n_tiles,
                (double)(h[6] + h[7] + h[8] + h[9] + h[10]));
    } else {
        k2_tile_graph<false, 768, 64><<<n_tiles, 64, 0, q>>>(A1, nullptr);
        k2_tile_graph<false, 2048, 128><<<n_tiles, 64, 0, q>>>(A2, nullptr);
    }
    uint32_t ctr[3] = {0, 0, 0};
    TILEOK(npcopy::d2h(ctr, tilectr_.p, 12, q));
    TILEOK(hipStreamSynchronize(q));
    if (clk && clk->on && ctr[2]) fprintf(stderr, "[np2 graph] %u of %u tiles took the large pool\n", ctr[2], n_tiles);
    if (clk) clk->mark("tiles.graph");
    if (ctr[1]) return 2;
    const uint32_t total = ctr[0];
    if (!live_.ensure((size_t)total + 64) || !ematch_.ensure(sizeof(EMatch) * (size_t)total + 64)) { *err = "out of device memory (link graph)"; return 0; }
    if (total) {
        TILEOK(hipMemsetAsync(live_.p, 1, total, q));
        MsaView bare{coloff_.as<uint32_t>(), colnn_.as<uint32_t>(), nodes_.as<Node>(), entries_.as<Entry>(), nullptr};
        k2_match<<<nblk(total, 256), 256, 0, q>>>(bare, nullptr, live_.as<uint8_t>(), total, ematch_.as<EMatch>());
    }
#undef TILEOK
    *total_out = total;
    graph_compact_ = true;
    return 1;
}

bool HipExec::build_graph(const std::vector<uint32_t>& n_tags, uint32_t n_cols, uint32_t* total_out, std::string* err, StageClock* clk,
                          const uint32_t* m_seen) {
    hipStream_t q = stream_;
    // ---- chunk list (host: O(streams)), non-insertion tag counts per chunk, scan
    std::vector<ChunkDesc> cd;
    win_first_chunk_.assign(n_tags.size(), 0);
    win_n_chunks_.assign(n_tags.size(), 0);
    for (uint32_t s = 0; s < (uint32_t)n_tags.size(); ++s) {
        const uint32_t first_chunk = (uint32_t)cd.size();
        for (uint32_t t = 0; t < n_tags[s]; t += LINK_CHUNK) cd.push_back(ChunkDesc{s, t, std::min(LINK_CHUNK, n_tags[s] - t), first_chunk});
        win_first_chunk_[s] = first_chunk;
        win_n_chunks_[s] = (uint32_t)cd.size() - first_chunk;
    }
    const uint32_t n_chunks = (uint32_t)cd.size();
    if (!chunks_.ensure(sizeof(ChunkDesc) * (size_t)n_chunks + 64) || !chcnt_.ensure(4ull * (n_chunks + 2)) || !chpre_.ensure(4ull * (n_chunks + 2)) ||
        !sums2_.ensure(4ull * (nblk(n_chunks + 1, SCAN_TILE) + 2))) { *err = "out of device memory (chunks)"; return false; }
    if (n_chunks) {
        HIPOK(npcopy::h2d(chunks_.p, cd.data(), sizeof(ChunkDesc) * (size_t)n_chunks, q));
        HIPOK(hipMemsetAsync(chcnt_.as<uint32_t>() + n_chunks, 0, 4, q));
        k2_chunk_count<<<nblk(n_chunks, 4), 256, 0, q>>>(chunks_.as<ChunkDesc>(), n_chunks, tagoff_.as<uint64_t>(), tags_.as<uint8_t>(), chcnt_.as<uint32_t>());
        const uint32_t nsc = nblk(n_chunks + 1, SCAN_TILE);
        k2_scan_sums<<<nsc, SCAN_T, 0, q>>>(chcnt_.as<uint32_t>(), n_chunks + 1, sums2_.as<uint32_t>());
        k2_scan_top<<<1, 1, 0, q>>>(sums2_.as<uint32_t>(), nsc);
        k2_scan_final<<<nsc, SCAN_T, 0, q>>>(chcnt_.as<uint32_t>(), n_chunks + 1, sums2_.as<uint32_t>(), chpre_.as<uint32_t>());
    }
    graph_compact_ = false;
    static const bool scatter_only = getenv("NP2_GRAPH_SCATTER") != nullptr;
    if (n_chunks && !scatter_only) {
        const int r = build_graph_tiles(n_tags, n_chunks, n_cols, total_out, err, clk);
        if (r == 0) return false;
        if (r == 1) return true;
        if (getenv("NP2_TIMING")) fprintf(stderr, "[np2 graph] a tile overflowed: scatter path for this window\n");
    }
    if (n_chunks && !m_seen)
        k2_chunk_links<false><<<nblk(n_chunks, 64), 64, 0, q>>>(chunks_.as<ChunkDesc>(), n_chunks, chpre_.as<uint32_t>(), tagoff_.as<uint64_t>(),
                                                                 alnts_.as<uint32_t>(), tags_.as<uint8_t>(), colcnt_.as<uint32_t>(), nullptr, nullptr, nullptr);
    const uint32_t nsb = nblk(n_cols + 1, SCAN_TILE);
    uint32_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        k2_scan_sums<<<nsb, SCAN_T, 0, q>>>(colcnt_.as<uint32_t>(), n_cols + 1, sums_.as<uint32_t>());
        k2_scan_top<<<1, 1, 0, q>>>(sums_.as<uint32_t>(), nsb);
        k2_scan_final<<<nsb, SCAN_T, 0, q>>>(colcnt_.as<uint32_t>(), n_cols + 1, sums_.as<uint32_t>(), coloff_.as<uint32_t>());
        uint32_t flag = 0;
        HIPOK(npcopy::d2h(&total, coloff_.as<uint32_t>() + n_cols, 4, q));
        if (m_seen && pass == 0) HIPOK(npcopy::d2h(&flag, m_seen, 4, q));
        HIPOK(hipStreamSynchronize(q));
        if (!flag) break;
        // a read carries the base code M: count the observations exactly (rare)
        HIPOK(hipMemsetAsync(colcnt_.p, 0, 4ull * (n_cols + 2), q));
        if (n_chunks)
            k2_chunk_links<false><<<nblk(n_chunks, 64), 64, 0, q>>>(chunks_.as<ChunkDesc>(), n_chunks, chpre_.as<uint32_t>(), tagoff_.as<uint64_t>(),
                                                                     alnts_.as<uint32_t>(), tags_.as<uint8_t>(), colcnt_.as<uint32_t>(), nullptr, nullptr, nullptr);
    }
    if (clk) clk->mark("links.count+scan");
    if (!obs_.ensure(sizeof(DevObs) * (size_t)total + 64) || !obsaux_.ensure(8ull * total + 64) ||
        !entries_.ensure(sizeof(Entry) * (size_t)total + 64) || !live_.ensure((size_t)total + 64) || !ematch_.ensure(sizeof(EMatch) * (size_t)total + 64) ||
        !nodes_.ensure(sizeof(Node) * (size_t)total + 64)) {
        *err = "out of device memory (link graph)";
        return false;
    }
    if (n_chunks)
        k2_chunk_links<true><<<nblk(n_chunks, 64), 64, 0, q>>>(chunks_.as<ChunkDesc>(), n_chunks, chpre_.as<uint32_t>(), tagoff_.as<uint64_t>(),
                                                                alnts_.as<uint32_t>(), tags_.as<uint8_t>(), nullptr, coloff_.as<uint32_t>(), cursor_.as<uint32_t>(),
                                                                obs_.as<DevObs>());
    if (clk) clk->mark("links.scatter");
    HIPOK(hipMemsetAsync(colnn_.p, 0, 4ull * n_cols, q));
    if (total) {
        k2_build_a<<<nblk(total, 256), 256, 0, q>>>(obs_.as<DevObs>(), coloff_.as<uint32_t>(), total, obsaux_.as<uint64_t>());
        HIPOK(hipMemsetAsync(live_.p, 0, total, q));
        k2_build_b<<<nblk(total, 256), 256, 0, q>>>(obs_.as<DevObs>(), coloff_.as<uint32_t>(), total, obsaux_.as<uint64_t>(),
                                                     entries_.as<Entry>(), nodes_.as<Node>(), colnn_.as<uint32_t>(), live_.as<uint8_t>());
        MsaView bare{coloff_.as<uint32_t>(), colnn_.as<uint32_t>(), nodes_.as<Node>(), entries_.as<Entry>(), nullptr};
        k2_match<<<nblk(total, 256), 256, 0, q>>>(bare, nullptr, live_.as<uint8_t>(), total, ematch_.as<EMatch>());
    }
    *total_out = total;
    return true;
}

bool HipExec::extract(const std::vector<SubReq>& req, std::vector<uint32_t>* off, std::string* bases, std::string* err) {
    HIPOK(hipSetDevice(device_));
    hipStream_t q = stream_;
    const uint32_t n = (uint32_t)req.size();
    off->assign((size_t)n + 1, 0);
    bases->clear();
    if (!n) return true;
    if (!win_tags_live_) { *err = "extract without a window in HBM"; return false; }
    std::vector<SubReqDev> rd(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (req[i].stream >= win_n_chunks_.size() || !win_n_chunks_[req[i].stream]) { *err = "extract: bad stream"; return false; }
        rd[i] = SubReqDev{req[i].stream, req[i].start, req[i].end, win_first_chunk_[req[i].stream], win_n_chunks_[req[i].stream]};
    }
    const uint32_t nsb = nblk(n + 1, SCAN_TILE);
    if (!xreq_.ensure(sizeof(SubReqDev) * (size_t)n) || !xfirst_.ensure(4ull * n) || !xlen_.ensure(4ull * (n + 2)) || !xoff_.ensure(4ull * (n + 2)) ||
        !sums2_.ensure(4ull * (nsb + 2))) { *err = "out of device memory (candidates)"; return false; }
    HIPOK(npcopy::h2d(xreq_.p, rd.data(), sizeof(SubReqDev) * (size_t)n, q));
    HIPOK(hipMemsetAsync(xlen_.as<uint32_t>() + n, 0, 4, q));
    k2_extract<false><<<nblk(n, 64), 64, 0, q>>>(xreq_.as<SubReqDev>(), n, chpre_.as<uint32_t>(), tagoff_.as<uint64_t>(), alnts_.as<uint32_t>(), tags_.as<uint8_t>(),
                                                  xfirst_.as<uint32_t>(), xlen_.as<uint32_t>(), nullptr, nullptr);
    k2_scan_sums<<<nsb, SCAN_T, 0, q>>>(xlen_.as<uint32_t>(), n + 1, sums2_.as<uint32_t>());
    k2_scan_top<<<1, 1, 0, q>>>(sums2_.as<uint32_t>(), nsb);
    k2_scan_final<<<nsb, SCAN_T, 0, q>>>(xlen_.as<uint32_t>(), n + 1, sums2_.as<uint32_t>(), xoff_.as<uint32_t>());
    HIPOK(npcopy::d2h(off->data(), xoff_.p, 4ull * (n + 1), q));
    HIPOK(hipStreamSynchronize(q));
    const uint32_t total = (*off)[n];
    if (total) {
        if (!xout_.ensure((size_t)total + 16)) { *err = "out of device memory (candidates)"; return false; }
        k2_extract<true><<<nblk(n, 64), 64, 0, q>>>(xreq_.as<SubReqDev>(), n, chpre_.as<uint32_t>(), tagoff_.as<uint64_t>(), alnts_.as<uint32_t>(), tags_.as<uint8_t>(),
                                                     xfirst_.as<uint32_t>(), nullptr, xoff_.as<uint32_t>(), xout_.as<char>());
        bases->resize(total);
        HIPOK(npcopy::d2h(&(*bases)[0], xout_.p, total, q));
        HIPOK(hipStreamSynchronize(q));
    }
    return true;
}

bool HipExec::read_coords(const std::vector<CoordReq>& req, std::vector<uint32_t>* bases, std::string* err) {
    HIPOK(hipSetDevice(device_));
    hipStream_t q = stream_;
    const uint32_t n = (uint32_t)req.size();
    bases->assign(n, 0);
    if (!n) return true;
    if (!win_tags_live_) { *err = "read_coords without a window in HBM"; return false; }
    std::vector<CoordReqDev> rd(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (req[i].stream >= win_n_chunks_.size() || !win_n_chunks_[req[i].stream]) { *err = "read_coords: bad stream"; return false; }
        rd[i] = CoordReqDev{req[i].stream, req[i].col, req[i].through_col, 0u};
    }
    if (!xreq_.ensure(sizeof(CoordReqDev) * (size_t)n) || !xlen_.ensure(4ull * (n + 2))) { *err = "out of device memory (read coordinates)"; return false; }
    HIPOK(npcopy::h2d(xreq_.p, rd.data(), sizeof(CoordReqDev) * (size_t)n, q));
    k2_read_coord<<<nblk(n, 64), 64, 0, q>>>(xreq_.as<CoordReqDev>(), n, tagoff_.as<uint64_t>(), alnts_.as<uint32_t>(), tags_.as<uint8_t>(), xlen_.as<uint32_t>());
    HIPOK(npcopy::d2h(bases->data(), xlen_.p, 4ull * n, q));
    HIPOK(hipStreamSynchronize(q));
    return true;
}

bool HipExec::run_lq(const LqInput& in, std::string* cons_rev, std::string* err) {
    HIPOK(hipSetDevice(device_));
    hipStream_t q = stream_;
    const uint32_t n_streams = (uint32_t)in.t.size();
    std::vector<char> pool;
    std::vector<uint64_t> str_off;
    std::vector<uint32_t> str_len;
    for (uint32_t i = 0; i < n_streams; ++i) {
        str_off.push_back(pool.size());
        pool.insert(pool.end(), in.t[i].begin(), in.t[i].end());
        pool.push_back('\0');
        str_off.push_back(pool.size());
        pool.insert(pool.end(), in.q[i].begin(), in.q[i].end());
        pool.push_back('\0');
        str_len.push_back((uint32_t)in.t[i].size());
    }
    if (!strpool_.ensure(pool.size() + 16) || !stroff_.ensure(8ull * str_off.size() + 16)) { *err = "out of device memory (low-quality regions)"; return false; }
    HIPOK(npcopy::h2d(strpool_.p, pool.data(), pool.size(), q));
    HIPOK(npcopy::h2d(stroff_.p, str_off.data(), 8ull * str_off.size(), q));
    HIPOK(hipStreamSynchronize(q));   // pool / str_off are locals
    return lq_from_pool(str_len, in.t_len, in.gap_min_len, in.hifi, cons_rev, err);
}

bool HipExec::run_lq_aligned(const LqAlignInput& in, std::string* cons_rev, std::string* err) {
    HIPOK(hipSetDevice(device_));
    hipStream_t q = stream_;
    const uint32_t n_regs = (uint32_t)in.regions.size();
    if (n_regs == 0) { *err = "no low-quality region to align"; return false; }
    // ---- host: the regions, and the (region, round) pairs that are aligned at all (length rule of ctg_cns.c:1343-1344)
    std::vector<OndRegion> regs(n_regs);
    std::vector<int32_t> pair_of((size_t)n_regs * LQ_ROUNDS, -1);
    std::vector<np2ond::Pair> pairs;
    std::vector<uint64_t> coff(in.cand_off.begin(), in.cand_off.end());
    uint64_t out_at = 0;
    uint32_t t_len = 1, max_sum = 0;
    for (uint32_t j = 0; j < n_regs; ++j) {
        const LqAlignRegion& r = in.regions[j];
        regs[j] = OndRegion{r.seed_off, r.seed_len, r.first_cand, r.n_cand, (uint32_t)pairs.size()};
        t_len += r.seed_len + 1;
        for (uint32_t i = 0; i < (uint32_t)LQ_ROUNDS && i < r.n_cand; ++i) {
            const int query_len = (int)in.cand_len[r.first_cand + i], seed_len = (int)r.seed_len;
            if (i && (query_len < seed_len * 0.5 || query_len > seed_len * 1.3)) continue;
            np2ond::Pair P;
            P.q_off = in.cand_off[r.first_cand + i]; P.t_off = r.seed_off; P.q_len = (uint32_t)query_len; P.t_len = r.seed_len;
            P.out_cap = P.q_len + P.t_len + 2; P.out_off = out_at; P.pad = 0;
            out_at += 2ull * P.out_cap;
            pair_of[(size_t)j * LQ_ROUNDS + i] = (int32_t)pairs.size();
            pairs.push_back(P);
            if (P.q_len + P.t_len > max_sum) max_sum = P.q_len + P.t_len;
        }
    }
    const uint32_t n_pairs = (uint32_t)pairs.size();
    const uint32_t max_d_cap = (uint32_t)(0.4 * (double)max_sum) + 1, row_words = (max_d_cap + 1 + 63) / 64;
    const uint32_t grid = std::min<uint32_t>(nblk(std::max(1u, n_pairs), 4), 2048u), n_slots = grid * 4;
    if (!ondpool_.ensure(in.chars.size() + 64) || !ondregs_.ensure(sizeof(OndRegion) * (size_t)n_regs) || !ondcoff_.ensure(8ull * coff.size() + 16) ||
        !ondclen_.ensure(4ull * in.cand_len.size() + 16) || !ondpairof_.ensure(4ull * pair_of.size()) || !ondpairs_.ensure(sizeof(np2ond::Pair) * (size_t)(n_pairs + 1)) ||
        !ondres_.ensure(sizeof(np2ond::PairResult) * (size_t)(n_pairs + 1)) || !ondout_.ensure(out_at + 64) ||
        !ondv_.ensure(4ull * n_slots * (2ull * max_d_cap + 4)) || !ondlo_.ensure(4ull * n_slots * ((size_t)max_d_cap + 1)) ||
        !ondch_.ensure(8ull * n_slots * (size_t)(max_d_cap + 1) * row_words) || !ondplen_.ensure(4ull * LQ_ROUNDS * n_regs) ||
        !ondpkind_.ensure((size_t)LQ_ROUNDS * n_regs + 16) || !ondppos_.ensure(4ull * LQ_ROUNDS * n_regs) || !ondtot_.ensure(4ull * LQ_ROUNDS + 16) ||
        !stroff_.ensure(16ull * LQ_ROUNDS + 16)) {
        *err = "out of device memory (low-quality alignments)";
        return false;
    }
    HIPOK(npcopy::h2d(ondpool_.p, in.chars.data(), in.chars.size(), q));
    HIPOK(npcopy::h2d(ondregs_.p, regs.data(), sizeof(OndRegion) * (size_t)n_regs, q));
    HIPOK(npcopy::h2d(ondcoff_.p, coff.data(), 8ull * coff.size(), q));
    HIPOK(npcopy::h2d(ondclen_.p, in.cand_len.data(), 4ull * in.cand_len.size(), q));
    HIPOK(npcopy::h2d(ondpairof_.p, pair_of.data(), 4ull * pair_of.size(), q));
    if (n_pairs) HIPOK(npcopy::h2d(ondpairs_.p, pairs.data(), sizeof(np2ond::Pair) * (size_t)n_pairs, q));
    if (n_pairs)
        k2_ond_align<<<grid, 256, 0, q>>>(ondpool_.as<uint8_t>(), ondpairs_.as<np2ond::Pair>(), n_pairs, ondout_.as<uint8_t>(), ondres_.as<np2ond::PairResult>(),
                                         ondv_.as<int32_t>(), ondlo_.as<int32_t>(), ondch_.as<uint64_t>(), max_d_cap, row_words);
    k2_ond_pieces<<<nblk(n_regs, 64), 64, 0, q>>>(ondregs_.as<OndRegion>(), n_regs, ondclen_.as<uint32_t>(), ondpairof_.as<int32_t>(),
                                                 ondres_.as<np2ond::PairResult>(), ondplen_.as<uint32_t>(), ondpkind_.as<uint8_t>());
    k2_ond_scan<<<LQ_ROUNDS, 256, 0, q>>>(ondplen_.as<uint32_t>(), n_regs, ondppos_.as<uint32_t>(), ondtot_.as<uint32_t>());
    std::vector<uint32_t> str_len(LQ_ROUNDS);
    HIPOK(npcopy::d2h(str_len.data(), ondtot_.p, 4ull * LQ_ROUNDS, q));
    HIPOK(hipStreamSynchronize(q));
    std::vector<uint64_t> str_off;
    uint64_t at = 0;
    for (int i = 0; i < LQ_ROUNDS; ++i) {
        str_off.push_back(at); at += (uint64_t)str_len[(size_t)i] + 1;
        str_off.push_back(at); at += (uint64_t)str_len[(size_t)i] + 1;
    }
    if (!strpool_.ensure(at + 16)) { *err = "out of device memory (low-quality regions)"; return false; }
    HIPOK(npcopy::h2d(stroff_.p, str_off.data(), 8ull * str_off.size(), q));
    k2_ond_emit<<<nblk((uint64_t)LQ_ROUNDS * n_regs, 4), 256, 0, q>>>(ondregs_.as<OndRegion>(), n_regs, ondpool_.as<uint8_t>(), ondcoff_.as<uint64_t>(),
                                                                     ondclen_.as<uint32_t>(), ondpairof_.as<int32_t>(), ondpairs_.as<np2ond::Pair>(),
                                                                     ondres_.as<np2ond::PairResult>(), ondout_.as<uint8_t>(), ondplen_.as<uint32_t>(),
                                                                     ondpkind_.as<uint8_t>(), ondppos_.as<uint32_t>(), ondtot_.as<uint32_t>(),
                                                                     stroff_.as<uint64_t>(), strpool_.as<char>());
    HIPOK(hipStreamSynchronize(q));   // str_off is a local
    return lq_from_pool(str_len, t_len, in.gap_min_len, in.hifi, cons_rev, err);
}

// The concatenated gapped strings are in strpool_ (string i: t at stroff_[2 i], q at stroff_[2 i + 1], NUL-terminated, str_len[i]
// characters each): tags -> link graph -> DP -> backtrace (get_lqseqs_from_align_tags, ctg_cns.c:986-1163).
bool HipExec::lq_from_pool(const std::vector<uint32_t>& str_len, uint32_t t_len, uint32_t gap_min_len, bool hifi, std::string* cons_rev, std::string* err) {
    HIPOK(hipSetDevice(device_));
    win_tags_live_ = false;   // the chunk tables are rebuilt for the concatenated regions
    hipStream_t q = stream_;
    const uint32_t n_streams = (uint32_t)str_len.size();
    const uint32_t n_cols = t_len + 1 + 32;   // slack: see the fill quirk in np2_lq.cpp
    std::vector<uint64_t> tag_off;
    std::vector<uint32_t> zeros(n_streams, 0);
    uint64_t tag_bytes = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        tag_off.push_back(tag_bytes);
        tag_bytes += ((uint64_t)str_len[i] + 1) / 2 + 1;
        tag_bytes = (tag_bytes + 3) & ~3ull;
    }
    if (!strlen_.ensure(4ull * n_streams + 16) ||
        !tags_.ensure(tag_bytes + 16) || !cnt4_.ensure(16ull * n_cols + 64) || !stat_.ensure(sizeof(ColStat) * (size_t)n_cols) ||
        !tagoff_.ensure(8ull * n_streams + 16) || !alnts_.ensure(4ull * n_streams + 16) || !te_.ensure(4ull * n_streams + 16) ||
        !colcnt_.ensure(4ull * (n_cols + 2)) || !coloff_.ensure(4ull * (n_cols + 2)) || !cursor_.ensure(4ull * (n_cols + 2)) ||
        !sums_.ensure(4ull * (nblk(n_cols + 1, SCAN_TILE) + 2)) || !colnn_.ensure(4ull * n_cols) || !res_.ensure(sizeof(DpResult))) {
        *err = "out of device memory (low-quality regions)";
        return false;
    }
    HIPOK(npcopy::h2d(strlen_.p, str_len.data(), 4ull * n_streams, q));
    HIPOK(npcopy::h2d(tagoff_.p, tag_off.data(), 8ull * n_streams, q));
    HIPOK(npcopy::h2d(alnts_.p, zeros.data(), 4ull * n_streams, q));
    HIPOK(hipMemsetAsync(tags_.p, 0, tag_bytes + 16, q));
    HIPOK(hipMemsetAsync(cnt4_.p, 0, 16ull * n_cols + 64, q));
    HIPOK(hipMemsetAsync(colcnt_.p, 0, 4ull * (n_cols + 2), q));
    HIPOK(hipMemsetAsync(cursor_.p, 0, 4ull * (n_cols + 2), q));
    DevStat st{cnt4_.as<uint32_t>(), cnt4_.as<uint32_t>() + n_cols, cnt4_.as<uint32_t>() + 2ull * n_cols, cnt4_.as<uint32_t>() + 3ull * n_cols};
    {
        std::vector<StrChunk> scs;
        for (uint32_t i = 0; i < n_streams; ++i) {
            const uint32_t first = (uint32_t)scs.size(), nch = (str_len[i] + TAG_CHUNK - 1) / TAG_CHUNK;
            for (uint32_t c = 0; c < nch; ++c)
                scs.push_back(StrChunk{i, c * TAG_CHUNK, std::min(TAG_CHUNK, str_len[i] - c * TAG_CHUNK), first, c + 1 == nch ? 1u : 0u, 0, 0, 0});
            if (nch == 0) { *err = "empty low-quality alignment"; return false; }
        }
        const uint32_t nsc = (uint32_t)scs.size();
        if (!tchunks_.ensure(sizeof(StrChunk) * (size_t)nsc + 64) || !chcnt_.ensure(4ull * (nsc + 2)) || !chpre_.ensure(4ull * (nsc + 2)) ||
            !sums2_.ensure(4ull * (nblk(nsc + 1, SCAN_TILE) + 2))) { *err = "out of device memory (low-quality chunks)"; return false; }
        HIPOK(npcopy::h2d(tchunks_.p, scs.data(), sizeof(StrChunk) * (size_t)nsc, q));
        HIPOK(hipMemsetAsync(chcnt_.as<uint32_t>() + nsc, 0, 4, q));
        k2_str_count<<<nblk(nsc, 64), 64, 0, q>>>(tchunks_.as<StrChunk>(), nsc, strpool_.as<char>(), stroff_.as<uint64_t>(), chcnt_.as<uint32_t>());
        const uint32_t nss = nblk(nsc + 1, SCAN_TILE);
        k2_scan_sums<<<nss, SCAN_T, 0, q>>>(chcnt_.as<uint32_t>(), nsc + 1, sums2_.as<uint32_t>());
        k2_scan_top<<<1, 1, 0, q>>>(sums2_.as<uint32_t>(), nss);
        k2_scan_final<<<nss, SCAN_T, 0, q>>>(chcnt_.as<uint32_t>(), nsc + 1, sums2_.as<uint32_t>(), chpre_.as<uint32_t>());
        k2_tags_str_chunk<<<nblk(nsc, 64), 64, 0, q>>>(tchunks_.as<StrChunk>(), nsc, chpre_.as<uint32_t>(), strpool_.as<char>(), stroff_.as<uint64_t>(),
                                                       gap_min_len, tagoff_.as<uint64_t>(), tags_.as<uint8_t>(), st, te_.as<uint32_t>());
    }
    k2_pack_stat<<<nblk(n_cols, 256), 256, 0, q>>>(st.coverage, st.max_size, st.l_ins, st.l_del, n_cols, stat_.as<ColStat>(), 0u, nullptr, nullptr, nullptr);
    uint32_t total = 0;
    if (!build_graph(str_len, n_cols, &total, err)) return false;
    MsaView mv{coloff_.as<uint32_t>(), colnn_.as<uint32_t>(), nodes_.as<Node>(), entries_.as<Entry>(), stat_.as<ColStat>(),
               getenv("NP2_NO_MATCH") ? nullptr : ematch_.as<EMatch>()};
    uint32_t cons_len = 0;
    if (!solve(mv, (int32_t)t_len, n_cols, total, hifi ? RULE_LQ_HIFI : RULE_LQ, &cons_len, nullptr, err)) return false;
    std::string fwd(cons_len, '\0');
    if (cons_len) HIPOK(npcopy::d2h(&fwd[0], cons_.p, cons_len, q));
    HIPOK(hipStreamSynchronize(q));
    cons_rev->assign(fwd.rbegin(), fwd.rend());   // the reference leaves this string in backtrace order
    return true;
}


// The partial-order pseudo-seeds of a window's low-quality regions in one launch.  Jobs the kernel cannot hold (graph larger than
// its LDS arrays, strings over 255 characters, score table over the slot's scratch) come back flagged and are done by the host
// version (np2_poa.cpp): same result either way.
bool HipExec::run_poa(const PoaBatch& in, std::vector<std::string>* out, std::string* err) {
    HIPOK(hipSetDevice(device_));
    hipStream_t q = stream_;
    const uint32_t n_jobs = (uint32_t)in.job_first.size(), n_str = (uint32_t)in.str_off.size();
    out->assign(n_jobs, std::string());
    if (!n_jobs) return true;
    static const bool host_only = getenv("NP2_POA_HOST") != nullptr;
    std::vector<uint32_t> status(n_jobs, 1u), olen(n_jobs, 0u);
    std::vector<np2poa::Job> jobs(n_jobs);
    std::string obuf;
    if (!host_only) {
        uint64_t out_total = 0;
        static const bool no_small = getenv("NP2_POA_BIG_ONLY") != nullptr;      // test hooks: every job in the first version's class /
        static const bool small_only = getenv("NP2_POA_SMALL_ONLY") != nullptr;  // what the Small class gives back goes straight to the host version
        std::vector<std::pair<uint64_t, uint32_t>> list_small, list_big;          // (work, job): longest first
        for (uint32_t j = 0; j < n_jobs; ++j) {
            uint32_t cap = 0, longest = 0;
            for (uint32_t k = 0; k < in.job_n[j]; ++k) { const uint32_t l = in.str_len[in.job_first[j] + k]; cap += l + 1; longest = std::max(longest, l); }
            // the Small class holds 3072 table cells: (nodes + 1) x (length + 1), and a graph of six ~L-character candidates ends with ~1.4 L
            // nodes (measured on the test windows: 6 x 50 characters -> 70 nodes).  A job expected not to fit starts in the Big class at once
            // instead of after the Small one has given it back.
            const bool small = !no_small && in.job_n[j] <= np2poa::Small::MAXSTR && longest <= np2poa::Small::MAXLEN &&
                               (uint64_t)(longest * 3 / 2 + 2) * (longest + 1) <= np2poa::Small::TAB_LDS;
            jobs[j] = np2poa::Job{in.job_first[j], in.job_n[j], out_total, cap, small ? 1u : 0u};
            (small ? list_small : list_big).emplace_back((uint64_t)cap * (longest + 1), j);
            out_total += cap;
        }
        std::vector<uint32_t> order;
        for (auto* l : {&list_small, &list_big}) {
            std::sort(l->begin(), l->end(), [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
            for (auto& e : *l) order.push_back(e.second);
        }
        const uint32_t n_small = (uint32_t)list_small.size(), n_big = (uint32_t)list_big.size();
        const uint32_t slots = std::min<uint32_t>(std::max<uint32_t>(n_big, std::min<uint32_t>(n_small, 256u)), 1792u);   // resident Big waves (7 per CU by their LDS), each with its slice of table scratch
        constexpr uint32_t TAB_CAP = 1u << 16;          // cells of score table per resident Big wave (rows x columns)
        if (!poapool_.ensure(in.chars.size() + 64) || !poaoff_.ensure(4ull * n_str + 64) || !poalen_.ensure(4ull * n_str + 64) ||
            !poajobs_.ensure(sizeof(np2poa::Job) * (size_t)n_jobs + 64) || !poatabs_.ensure(4ull * TAB_CAP * slots + 64) ||
            !poatabf_.ensure(4ull * TAB_CAP * slots + 64) || !poaout_.ensure(out_total + 64) || !poaolen_.ensure(4ull * n_jobs + 64) ||
            !poastat_.ensure(4ull * n_jobs + 64 + 64) || !poaord_.ensure(4ull * n_jobs + 64)) { *err = "out of device memory (pseudo-seeds)"; return false; }
        uint32_t* queue = poastat_.as<uint32_t>() + n_jobs + 4;      // three job counters behind the status words
        if (!stream2_) { HIPOK(npalloc::stream_create(&stream2_)); HIPOK(hipEventCreateWithFlags(&ev_up_, hipEventDisableTiming)); HIPOK(hipEventCreateWithFlags(&ev_big_, hipEventDisableTiming)); }
        HIPOK(hipMemsetAsync(queue, 0, 12, q));
        HIPOK(hipMemsetAsync(poastat_.p, 0xff, 4ull * n_jobs, q));      // (a job no launch reaches reads as "not done")
        HIPOK(npcopy::h2d(poapool_.p, in.chars.data(), in.chars.size(), q));
        HIPOK(npcopy::h2d(poaoff_.p, in.str_off.data(), 4ull * n_str, q));
        HIPOK(npcopy::h2d(poalen_.p, in.str_len.data(), 4ull * n_str, q));
        HIPOK(npcopy::h2d(poajobs_.p, jobs.data(), sizeof(np2poa::Job) * (size_t)n_jobs, q));
        HIPOK(npcopy::h2d(poaord_.p, order.data(), 4ull * n_jobs, q));
        HIPOK(hipEventRecord(ev_up_, q));
        const uint32_t small_slots = std::min<uint32_t>(std::max<uint32_t>(n_small, 1u), 256u * 9u);      // 16.9 KB of LDS a wave: 9 a CU
        // NP2_POA_DEBUG=1: every wave keeps {stage, counter, job} in a device array the host reads over another stream when the kernels have
        // not finished after five seconds (how round 5 found where a wave was spinning)
        static const bool debug = getenv("NP2_POA_DEBUG") != nullptr;
        uint32_t *dbg_small = nullptr, *dbg_big = nullptr;
        if (debug) {
            if (!poadbg_.ensure(16ull * (small_slots + slots) + 64)) { *err = "out of device memory (pseudo-seed debug)"; return false; }
            HIPOK(hipMemsetAsync(poadbg_.p, 0xff, 16ull * (small_slots + slots), q));
            HIPOK(hipEventRecord(ev_up_, q));
            dbg_small = poadbg_.as<uint32_t>();
            dbg_big = dbg_small + 4 * small_slots;
        }
        const uint32_t* ord = poaord_.as<uint32_t>();
        if (n_big) {      // the jobs that start in the Big class, on the second stream, while the Small class works on the others
            HIPOK(hipStreamWaitEvent(stream2_, ev_up_, 0));
            k2_poa<np2poa::Big><<<std::min<uint32_t>(n_big, slots), 64, 0, stream2_>>>(poapool_.as<char>(), poaoff_.as<uint32_t>(), poalen_.as<uint32_t>(), poajobs_.as<np2poa::Job>(), ord + n_small,
                                                                                       n_big, 0u, poatabs_.as<int32_t>(), poatabf_.as<uint32_t>(), TAB_CAP, poaout_.as<char>(),
                                                                                       poaolen_.as<uint32_t>(), poastat_.as<uint32_t>(), queue + 1, dbg_big);
            HIPOK(hipEventRecord(ev_big_, stream2_));
        }
        if (n_small)
            k2_poa<np2poa::Small><<<small_slots, 64, 0, q>>>(poapool_.as<char>(), poaoff_.as<uint32_t>(), poalen_.as<uint32_t>(), poajobs_.as<np2poa::Job>(), ord, n_small, 0u, nullptr,
                                                             nullptr, 0u, poaout_.as<char>(), poaolen_.as<uint32_t>(), poastat_.as<uint32_t>(), queue, dbg_small);
        if (n_big) HIPOK(hipStreamWaitEvent(q, ev_big_, 0));      // (the third launch reuses the Big slots' table scratch)
        if (n_small && !small_only)
            k2_poa<np2poa::Big><<<std::min<uint32_t>(n_small, slots), 64, 0, q>>>(poapool_.as<char>(), poaoff_.as<uint32_t>(), poalen_.as<uint32_t>(), poajobs_.as<np2poa::Job>(), ord, n_small, 1u,
                                                                                  poatabs_.as<int32_t>(), poatabf_.as<uint32_t>(), TAB_CAP, poaout_.as<char>(), poaolen_.as<uint32_t>(),
                                                                                  poastat_.as<uint32_t>(), queue + 2, dbg_big);
        if (getenv("NP2_TIMING")) fprintf(stderr, "[np2 poa] %u jobs, %u of them offered to the Small class\n", n_jobs, n_small);
        if (debug) {
            timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (;;) {
                if (hipStreamQuery(q) == hipSuccess) break;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > 5.0) {
                    hipStream_t q3;
                    std::vector<uint32_t> h(4 * (size_t)(small_slots + slots));
                    if (npalloc::stream_create(&q3) == hipSuccess && hipMemcpyAsync(h.data(), poadbg_.p, 4 * h.size(), hipMemcpyDeviceToHost, q3) == hipSuccess &&
                        hipStreamSynchronize(q3) == hipSuccess) {
                        for (uint32_t b = 0; b < small_slots + slots; ++b) {
                            const uint32_t* w = &h[4 * (size_t)b];
                            if (w[0] == 0xffffffffu || w[0] == 100u || w[0] == 101u) continue;
                            const uint32_t j = w[2];
                            fprintf(stderr, "[np2 poa debug] %s wave %u: stage %u counter %#x job %u", b < small_slots ? "Small" : "Big", b < small_slots ? b : b - small_slots, w[0], w[1], j);
                            if (j < n_jobs) { fprintf(stderr, " strings:"); for (uint32_t k = 0; k < in.job_n[j]; ++k) fprintf(stderr, " %u", in.str_len[in.job_first[j] + k]); }
                            fprintf(stderr, "\n");
                        }
                    }
                    fprintf(stderr, "[np2 poa debug] the pseudo-seed kernels did not finish within 5 s\n");
                    _exit(3);
                }
                usleep(1000);
            }
        }
        obuf.resize(out_total);
        HIPOK(npcopy::d2h(&obuf[0], poaout_.p, out_total, q));
        HIPOK(npcopy::d2h(olen.data(), poaolen_.p, 4ull * n_jobs, q));
        HIPOK(npcopy::d2h(status.data(), poastat_.p, 4ull * n_jobs, q));
        HIPOK(hipStreamSynchronize(q));
    }
    std::vector<uint32_t> todo;
    static const bool check = getenv("NP2_POA_CHECK") != nullptr;     // test hook: every device result against the host version
    for (uint32_t j = 0; j < n_jobs; ++j) {
        if (status[j] == 0) (*out)[j].assign(obuf.data() + jobs[j].out_off, olen[j]);
        else todo.push_back(j);
        if (check && status[j] == 0) {
            std::vector<std::string> v;
            for (uint32_t k = 0; k < in.job_n[j]; ++k) v.emplace_back(in.chars.data() + in.str_off[in.job_first[j] + k], in.str_len[in.job_first[j] + k]);
            const std::string want = poa_consensus(v);
            if (want != (*out)[j]) {
                *err = "pseudo-seed of the device differs from the host version (job " + std::to_string(j) + ": '" + (*out)[j] + "' vs '" + want + "')";
                return false;
            }
        }
    }
    if (check) fprintf(stderr, "[np2 poa] checked %u device pseudo-seeds against the host version (%zu left to the host)\n", n_jobs - (uint32_t)todo.size(), todo.size());
    if (!todo.empty()) {
        if (getenv("NP2_TIMING")) fprintf(stderr, "[np2 poa] %zu of %u pseudo-seeds on the host\n", todo.size(), n_jobs);
        np::parallel_for(todo.size(), 8, [&](size_t lo, size_t hi) {
            std::vector<std::string> v;
            for (size_t t = lo; t < hi; ++t) {
                const uint32_t j = todo[t];
                v.clear();
                for (uint32_t k = 0; k < in.job_n[j]; ++k) v.emplace_back(in.chars.data() + in.str_off[in.job_first[j] + k], in.str_len[in.job_first[j] + k]);
                (*out)[j] = poa_consensus(v);
            }
        });
    }
    return true;
}

}  // namespace

// NP2_PINNED_RECORDS=1: the big record arrays of the pipeline live in page-locked memory of this library (np2_exec.h: BigMem), installed
// when the library is loaded, before the first window builds them; np_hostcopy.h then copies them with no host-side staging.  Opt-in:
// measured on the long-read leg (24 workers x 12 calls of a 5 Mb window, tests/tools/r4_lgs_quick.py) 157.4 Mbp/s and 0.0859 host CPU-s per
// Mbp with it against 159.2 and 0.0867 through the ring -- the staging copy is not what the leg's host time is made of.
static void* big_make(size_t bytes) {
    std::string e;
    const int d = pick_device(&e);
    if (d < 0 || hipSetDevice(d) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (npalloc::host_malloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}
static void big_drop(void* p) { (void)npalloc::host_free(p); }
static struct BigMemInstall {
    BigMemInstall() { if (getenv("NP2_PINNED_RECORDS")) { BigMem::make = big_make; BigMem::drop = big_drop; } }
} g_big_mem_install;

Exec* make_exec(std::string* err) {
    const int d = pick_device(err);
    if (d < 0) return nullptr;
    HipExec* x = new HipExec(d);
    if (!x->init(err)) { delete x; return nullptr; }
    return x;
}

}  // namespace np2

extern "C" void np2_diag_report(int fd) { npalloc::report(fd, "nextpolish2.so"); }
extern "C" int np2_device_index(void) {
    std::string err;
    return np2::pick_device(&err);
}
