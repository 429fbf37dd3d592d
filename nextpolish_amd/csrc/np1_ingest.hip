// Device-side ingest of the short-read path: the compressed BAM bytes of a batch of contigs go to HBM as they lie in the
// file, and the GPU does what the reference does per record on the host with htslib (bgzf_read_block + bam_read1 behind
// sam_itr_next, source/lib/contig.c:172-174,692-694):
//
//   k_inflate        one wave per BGZF block (np_inflate_dev.h)                        compressed -> inflated BAM bytes
//   k_chase<false>   one wave per anchor segment: hop from record to record (LDS window)  records per segment
//   k_chase<true>    the same walk, writing the byte offset of every record            record offsets
//   k_rec_measure    one lane per record: the iterator's overlap test + pool sizes     keep flag, #CIGAR words, seq / qual bytes
//   k_rec_scatter    one lane per kept record: fixed fields -> SoA, CIGAR / bases / qualities -> aligned pools
//
// The hops need a known record start to begin from; the BAM index supplies one per 16 kb window (linear index) besides the
// first record of every contig (metadata pseudo-bin), so a 100 Mb batch is ~6 000 independent chains of a few thousand
// hops.  The result is exactly the record stream the host loader (np_stream.cpp:load_stream) builds, already in HBM.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np1_batch_priv.h"
#include "np1_ingest.h"
#include "np1_kmer_kernels.h"
#include "np_bgzf.h"
#include <sys/mman.h>

#include "np_inflate_dev.h"
#include "np_inflate_lane.h"
#include "np_inflate_lds.h"
#include "np_crc_dev.h"
#include "np_crc32.h"
#include "np_threads.h"
#include <atomic>

using namespace np1dev;

namespace {

constexpr uint32_t IG_ERR_CHAIN = 1u, IG_ERR_CGTAG = 2u, IG_ERR_RECORD = 4u;

struct Segment { uint64_t beg, end; uint32_t ctg; int32_t tid; int32_t ctg_len; uint32_t pad; };

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) {
    typedef uint32_t __attribute__((aligned(1))) u32u;
    return *reinterpret_cast<const u32u*>(p);
}
__device__ __forceinline__ uint16_t ld16(const uint8_t* p) {
    typedef uint16_t __attribute__((aligned(1))) u16u;
    return *reinterpret_cast<const u16u*>(p);
}

__global__ __launch_bounds__(256, 4) void k_inflate(const uint8_t* __restrict__ comp, const npdev::BlockDesc* __restrict__ blocks, uint32_t n_blocks,
                                                 uint8_t* out, uint32_t* __restrict__ status) {
    __shared__ npdev::InflateLds lds[4];
    const uint32_t wave = npdev::uni(threadIdx.x >> 6);     // wave-uniform by construction: tell the compiler (scalar loads, scalar decode state)
    const uint32_t b = blockIdx.x * 4 + wave;
    if (b >= n_blocks) return;
    const npdev::BlockDesc d = blocks[b];
    int rc = 0;
    if (d.out_len) rc = npdev::inflate_block_wave(comp + d.in_off, d.in_len, out + d.out_off, d.out_len, lds[wave]);
    if ((threadIdx.x & 63u) == 0) status[b] = (uint32_t)rc;
}

// Lane-per-block decoder (np_inflate_lane.h): every lane inflates blocks of its own, start to end (lane, lane + lanes, ...), with a
// table slice of its own in HBM scratch.  Throughput comes from the number of lanes in flight, not from the speed of one stream.
__global__ __launch_bounds__(64, 4) void k_inflate_lanes(const uint8_t* __restrict__ comp, const npdev::BlockDesc* __restrict__ blocks, uint32_t n_blocks,
                                                         uint8_t* out, uint32_t* __restrict__ status, uint32_t* __restrict__ tables) {
    const uint32_t lanes = gridDim.x * 64u, me = blockIdx.x * 64u + threadIdx.x;
    uint32_t* tab = tables + (size_t)me * nplane::LANE_TABLE_WORDS;
    for (uint32_t b = me; b < n_blocks; b += lanes) {
        const npdev::BlockDesc d = blocks[b];
        int rc = 0;
        if (d.out_len) rc = nplane::inflate_block(comp + d.in_off, d.in_len, out + d.out_off, d.out_len, tab);
        status[b] = (uint32_t)rc;
    }
}

// Lane-per-block decoder with its primary tables in LDS (np_inflate_lds.h; round 6): a workgroup is one wave, every lane inflates blocks of
// its own (lane, lane + lanes, ...); 2^LB + 2^DB 16-bit slots per lane, [slot][lane], fill the LDS of a CU with one wave (10 / 8 bits:
// exactly 160 KiB) or two (9 / 6 bits: 72 KiB each).  The grid is as many waves as the chip holds at once.
struct LdsTab {
    uint16_t* col;      // this lane's column of the [slot][lane] array
    __device__ __forceinline__ uint16_t rd(uint32_t i) const { return col[i << 6]; }
    __device__ __forceinline__ void wr(uint32_t i, uint16_t v) { col[i << 6] = v; }
};
template <int LB, int DB, int DBG = 0>
__global__ __launch_bounds__(64) void k_inflate_lds(const uint8_t* __restrict__ comp, const npdev::BlockDesc* __restrict__ blocks, uint32_t n_blocks, uint8_t* out,
                                                    uint32_t* __restrict__ status, nplds::Scratch* __restrict__ scratch) {
    __shared__ uint16_t slots[64u * nplds::Layout<LB, DB>::SLOTS];
    const uint32_t lanes = gridDim.x * 64u, me = blockIdx.x * 64u + threadIdx.x;
    LdsTab tab{slots + threadIdx.x};
    nplds::Scratch* sc = scratch + me;
    for (uint32_t b = me; b < n_blocks; b += lanes) {
        const npdev::BlockDesc d = blocks[b];
        int rc = 0;
        if (d.out_len) rc = nplds::inflate_block<LB, DB, LdsTab, DBG>(comp + d.in_off, d.in_len, out + d.out_off, d.out_len, tab, sc);
        status[b] = (uint32_t)rc;
    }
}
// waves the chip holds of the kernel at once (by its LDS), and the launch
template <int LB, int DB>
int launch_inflate_lds(hipStream_t q, const uint8_t* comp, const npdev::BlockDesc* blocks, uint32_t n_blocks, uint8_t* out, uint32_t* status, DevBuf& scratch) {
    static const uint32_t cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return (uint32_t)n;
    }();
    constexpr uint32_t lds_bytes = 64u * nplds::Layout<LB, DB>::SLOTS * 2u;
    // (NP1_LDS_WAVES_PER_CU: fewer than the LDS allows, for measurements)
    static const uint32_t cap_per_cu = getenv("NP1_LDS_WAVES_PER_CU") ? (uint32_t)atoi(getenv("NP1_LDS_WAVES_PER_CU")) : 64u;
    const uint32_t per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(cap_per_cu, 163840u / lds_bytes));
    const uint32_t waves = std::min<uint32_t>(cus * per_cu, (n_blocks + 63u) / 64u);
    if (scratch.ensure((size_t)waves * 64u * sizeof(nplds::Scratch))) return -1;
    // NP1_LDS_DBG=<bits>: timing experiments with parts of the work left out (np_inflate_lds.h; the output is wrong)
    static const int dbg = getenv("NP1_LDS_DBG") ? atoi(getenv("NP1_LDS_DBG")) : 0;
    if constexpr (LB == 7 && DB == 5) {      // (the experiment builds exist for the default table sizes only)
        if (dbg == 1) { k_inflate_lds<LB, DB, 1><<<waves, 64, 0, q>>>(comp, blocks, n_blocks, out, status, scratch.as<nplds::Scratch>()); return 0; }
        if (dbg == 2) { k_inflate_lds<LB, DB, 2><<<waves, 64, 0, q>>>(comp, blocks, n_blocks, out, status, scratch.as<nplds::Scratch>()); return 0; }
        if (dbg == 3) { k_inflate_lds<LB, DB, 3><<<waves, 64, 0, q>>>(comp, blocks, n_blocks, out, status, scratch.as<nplds::Scratch>()); return 0; }
    }
    k_inflate_lds<LB, DB><<<waves, 64, 0, q>>>(comp, blocks, n_blocks, out, status, scratch.as<nplds::Scratch>());
    return 0;
}

// NP1_INFLATE=lds<LB><DB> (lds96, lds86, lds85, lds76, lds75, lds65; "lds" = the default of the family): which table sizes
int lds_variant(const char* e) {
    if (!e || strncmp(e, "lds", 3) != 0) return 0;
    if (!e[3]) return 75;
    const int v = atoi(e + 3);
    return v == 96 || v == 86 || v == 85 || v == 76 || v == 75 || v == 65 ? v : 0;
}
int launch_inflate_lds_variant(int v, hipStream_t q, const uint8_t* comp, const npdev::BlockDesc* blocks, uint32_t n_blocks, uint8_t* out, uint32_t* status, DevBuf& scratch) {
    switch (v) {
        case 96: return launch_inflate_lds<9, 6>(q, comp, blocks, n_blocks, out, status, scratch);
        case 86: return launch_inflate_lds<8, 6>(q, comp, blocks, n_blocks, out, status, scratch);
        case 76: return launch_inflate_lds<7, 6>(q, comp, blocks, n_blocks, out, status, scratch);
        case 75: return launch_inflate_lds<7, 5>(q, comp, blocks, n_blocks, out, status, scratch);
        case 65: return launch_inflate_lds<6, 5>(q, comp, blocks, n_blocks, out, status, scratch);
        default: return launch_inflate_lds<8, 5>(q, comp, blocks, n_blocks, out, status, scratch);
    }
}

// gzip trailer CRC of every block the decoder accepted (np_crc_dev.h; the reference's htslib rejects a block whose CRC differs)
__global__ __launch_bounds__(256) void k_crc_check(const uint8_t* __restrict__ comp, const npdev::BlockDesc* __restrict__ blocks, uint32_t n_blocks,
                                                   const uint8_t* __restrict__ out, uint32_t* __restrict__ status, const uint32_t* __restrict__ shift) {
    npdev::crc_check_body(comp, blocks, n_blocks, out, status, shift);
}

// the same with phase clocks (diagnostics only)
__global__ __launch_bounds__(256, 4) void k_inflate_prof(const uint8_t* __restrict__ comp, const npdev::BlockDesc* __restrict__ blocks, uint32_t n_blocks,
                                                      uint8_t* out, uint32_t* __restrict__ status, unsigned long long* __restrict__ prof) {
    __shared__ npdev::InflateLds lds[4];
    const uint32_t wave = npdev::uni(threadIdx.x >> 6);
    const uint32_t b = blockIdx.x * 4 + wave;
    if (b >= n_blocks) return;
    const npdev::BlockDesc d = blocks[b];
    npdev::Prof pf;
    int rc = 0;
    if (d.out_len) rc = npdev::inflate_block_wave(comp + d.in_off, d.in_len, out + d.out_off, d.out_len, lds[wave], &pf);
    if ((threadIdx.x & 63u) == 0) {
        status[b] = (uint32_t)rc;
        const unsigned long long v[8] = {pf.t_tables, pf.t_decode, pf.t_flush, pf.tokens, pf.groups, pf.rounds, pf.matches, pf.match_bytes};
        for (int i = 0; i < 8; ++i) atomicAdd(&prof[i], v[i]);
    }
}

// One WAVE per segment [beg, end) of the inflated stream that starts on a record boundary.  The hop from record to record is a
// chain of dependent 4-byte reads a few hundred bytes apart: straight from HBM every hop costs a memory round trip, so the wave
// pulls the stream through an 8 KiB LDS window with coalesced 16-byte loads and hops inside it (the hop is wave-uniform: the
// position lives in scalar registers, the size word is an LDS broadcast); record offsets leave 64 at a time, one per lane.
constexpr uint32_t CHASE_WIN = 8192;
template <bool FILL>
__global__ __launch_bounds__(256) void k_chase(const uint8_t* __restrict__ u, const Segment* __restrict__ segs, uint32_t n_segs, uint32_t* __restrict__ counts,
                                               const uint64_t* __restrict__ rec_base, uint64_t* __restrict__ rec_off, uint32_t* __restrict__ rec_seg,
                                               uint32_t* __restrict__ err) {
    __shared__ uint4 win4[4][CHASE_WIN / 16 + 1];
    const uint32_t wave = npdev::uni(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t i = blockIdx.x * 4 + wave;
    if (i >= n_segs) return;
    const Segment s = segs[i];
    const uint8_t* win = reinterpret_cast<const uint8_t*>(win4[wave]);
    uint64_t p = s.beg, n = 0;
    const uint64_t base = FILL ? rec_base[i] : 0;
    uint64_t wbase = ~0ull;          // 16-byte aligned stream offset of the window's first byte
    uint64_t mine = 0;               // FILL: offset of record (n & ~63) + lane
    bool bad = false;
    while (p < s.end) {
        if (wbase == ~0ull || p < wbase || p + 4 > wbase + CHASE_WIN) {     // (re)load the window at p
            wbase = p & ~15ull;
            const uint4* src = reinterpret_cast<const uint4*>(u + wbase);   // the inflated buffer has slack behind its end
#pragma unroll
            for (uint32_t t = 0; t < CHASE_WIN / 16 / 64; ++t) win4[wave][t * 64 + lane] = src[t * 64 + lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t o = (uint32_t)(p - wbase);
        const uint32_t bs = npdev::uni((uint32_t)win[o] | (uint32_t)win[o + 1] << 8 | (uint32_t)win[o + 2] << 16 | (uint32_t)win[o + 3] << 24);
        if (bs < 32 || bs > (1u << 28) || p + 4 + bs > s.end) { bad = true; break; }
        if (FILL) {
            if (lane == (uint32_t)(n & 63u)) mine = p;
            if ((n & 63u) == 63u) { rec_off[base + (n & ~63ull) + lane] = mine; rec_seg[base + (n & ~63ull) + lane] = i; }
        }
        ++n;
        p += 4ull + bs;
    }
    if (bad && lane == 0) atomicOr(err, IG_ERR_CHAIN);
    if (FILL) {
        if ((n & 63u) && lane < (uint32_t)(n & 63u)) { rec_off[base + (n & ~63ull) + lane] = mine; rec_seg[base + (n & ~63ull) + lane] = i; }
    } else if (lane == 0) counts[i] = (uint32_t)n;
}

// record layout (SAMv1 4.2): block_size refID pos l_read_name mapq bin n_cigar_op flag l_seq next_refID next_pos tlen | name cigar seq qual aux
__global__ __launch_bounds__(256) void k_rec_measure(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_seg,
                                                     const Segment* __restrict__ segs, uint64_t n_rec, int with_qual, uint32_t* __restrict__ keep,
                                                     uint32_t* __restrict__ ncw, uint32_t* __restrict__ seqb, uint32_t* __restrict__ qualb,
                                                     uint32_t* __restrict__ max_lq, uint32_t* __restrict__ err) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lq_keep = 0;
    if (r < n_rec) {
        const uint8_t* p = u + rec_off[r];
        const Segment s = segs[rec_seg[r]];
        const uint32_t bs = ld32(p);
        const int32_t tid = (int32_t)ld32(p + 4), pos = (int32_t)ld32(p + 8);
        const uint32_t l_name = p[12];
        const uint32_t n_cigar = ld16(p + 16), flag = ld16(p + 18);
        const int32_t l_seq = (int32_t)ld32(p + 20);
        bool k = tid == s.tid && pos >= 0 && pos < s.ctg_len && l_seq >= 0;
        if (tid != s.tid || l_seq < 0 || 32u + l_name + 4u * n_cigar + (uint32_t)((l_seq + 1) / 2) + (uint32_t)l_seq > bs) { atomicOr(err, IG_ERR_RECORD); k = false; }
        if (k && pos == 0 && !(flag & 4u) && n_cigar > 0) {   // the iterator keeps a record iff its end is > 0: pos + rlen (htslib 1.9 bam_endpos)
            const uint8_t* c = p + 36 + l_name;
            uint32_t rl = 0;
            for (uint32_t j = 0; j < n_cigar; ++j) {
                const uint32_t op = ld32(c + 4 * j);
                const uint32_t o = op & 15u;
                if (o == 0 || o == 2 || o == 3 || o == 7 || o == 8) rl += op >> 4;
            }
            if (rl == 0) k = false;
        }
        if (k && n_cigar == 2) {   // placeholder CIGAR "<l_seq>S<rlen>N" of a record whose real CIGAR sits in the CG tag: the host loader swaps it in
            const uint8_t* c = p + 36 + l_name;
            const uint32_t o0 = ld32(c), o1 = ld32(c + 4);
            if ((o0 & 15u) == 4 && (o0 >> 4) == (uint32_t)l_seq && (o1 & 15u) == 3) atomicOr(err, IG_ERR_CGTAG);
        }
        keep[r] = k ? 1u : 0u;
        ncw[r] = k ? n_cigar : 0u;
        seqb[r] = k ? (uint32_t)((l_seq + 1) / 2) : 0u;
        if (with_qual) qualb[r] = k ? (uint32_t)l_seq : 0u;
        lq_keep = k ? (uint32_t)l_seq : 0u;
    }
    // one atomic per wave
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) lq_keep = max(lq_keep, (uint32_t)__shfl_xor(lq_keep, d, 64));
    if ((threadIdx.x & 63u) == 0 && lq_keep) atomicMax(max_lq, lq_keep);
}

struct ScatterOut {
    int32_t* pos; uint32_t* ctg; uint16_t* flag; uint32_t* ncig; int32_t* lq; uint64_t* cigoff; uint64_t* seqoff;
    uint32_t* cigar; uint8_t* seq; uint8_t* mapq; int32_t* isize; uint64_t* qualoff; uint8_t* qual;
};

// n bytes between two byte-aligned places, 16 at a time (round 6: word loads and BYTE stores before -- 75 store instructions per record, every one
// a transaction per lane; global loads and stores of any width take any alignment on this device, as the block decoder's do)
struct __attribute__((packed, aligned(1))) Bytes16 { uint64_t a, b; };
__device__ __forceinline__ void copy_unaligned(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n) {
    typedef uint64_t __attribute__((aligned(1))) u64u;
    typedef uint32_t __attribute__((aligned(1))) u32u;
    typedef uint16_t __attribute__((aligned(1))) u16u;
    uint32_t i = 0;
    for (; i + 16 <= n; i += 16) *reinterpret_cast<Bytes16*>(dst + i) = *reinterpret_cast<const Bytes16*>(src + i);
    if (n & 8u) { *reinterpret_cast<u64u*>(dst + i) = *reinterpret_cast<const u64u*>(src + i); i += 8; }
    if (n & 4u) { *reinterpret_cast<u32u*>(dst + i) = *reinterpret_cast<const u32u*>(src + i); i += 4; }
    if (n & 2u) { *reinterpret_cast<u16u*>(dst + i) = *reinterpret_cast<const u16u*>(src + i); i += 2; }
    if (n & 1u) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_rec_scatter(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_seg,
                                                     const Segment* __restrict__ segs, uint64_t n_rec, const uint32_t* __restrict__ keep,
                                                     const uint32_t* __restrict__ kidx, const uint64_t* __restrict__ cig_at, const uint64_t* __restrict__ seq_at,
                                                     const uint64_t* __restrict__ qual_at, int with_qual, ScatterOut o) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec || !keep[r]) return;
    const uint8_t* p = u + rec_off[r];
    const uint32_t j = kidx[r];
    const uint32_t l_name = p[12];
    const uint32_t n_cigar = ld16(p + 16);
    const int32_t l_seq = (int32_t)ld32(p + 20);
    o.pos[j] = (int32_t)ld32(p + 8);
    o.ctg[j] = segs[rec_seg[r]].ctg;
    o.flag[j] = ld16(p + 18);
    o.ncig[j] = n_cigar;
    o.lq[j] = l_seq;
    const uint64_t ca = cig_at[r], sa = seq_at[r];
    o.cigoff[j] = ca;
    o.seqoff[j] = sa;
    const uint8_t* c = p + 36 + l_name;
    for (uint32_t i = 0; i < n_cigar; ++i) o.cigar[ca + i] = ld32(c + 4 * i);
    const uint8_t* sq = c + 4 * n_cigar;
    const uint32_t sb = (uint32_t)((l_seq + 1) / 2);
    copy_unaligned(o.seq + sa, sq, sb);      // (the pool is only byte aligned per record)
    if (with_qual) {
        o.mapq[j] = p[13];
        o.isize[j] = (int32_t)ld32(p + 32);
        const uint64_t qa = qual_at[r];
        o.qualoff[j] = qa;
        const uint8_t* ql = sq + sb;
        copy_unaligned(o.qual + qa, ql, (uint32_t)l_seq);
    }
}

// BGZF virtual offsets of the kept records, as the reference's reader would report them with bgzf_tell before and behind each record
// (htslib 1.9 bgzf.c: a position at the end of a block is the start of the next block of the FILE): what the replay of the region
// iterator addresses records by (np1_replay.h).  geo[3 b .. 3 b + 3) = {file offset << 16, tell at the block's first byte, (file offset
// + size) << 16} of block b; blocks are in stream order, empty blocks share the stream offset of their successor.
__device__ __forceinline__ uint32_t block_of(const npdev::BlockDesc* __restrict__ blocks, uint32_t n_blocks, uint64_t p) {   // last block with out_off <= p
    uint32_t lo = 0, hi = n_blocks;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (blocks[mid].out_off <= p) lo = mid; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void k_rec_voff(const uint8_t* __restrict__ u, const uint64_t* __restrict__ rec_off, uint64_t n_rec, const uint32_t* __restrict__ keep,
                                                  const uint32_t* __restrict__ kidx, const npdev::BlockDesc* __restrict__ blocks, const uint64_t* __restrict__ geo,
                                                  uint32_t n_blocks, uint64_t* __restrict__ voff, uint64_t* __restrict__ voff_end) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec || !keep[r]) return;
    const uint64_t p = rec_off[r], e = p + 4ull + ld32(u + p);
    const uint32_t j = kidx[r];
    const uint32_t b0 = block_of(blocks, n_blocks, p);
    voff[j] = p == blocks[b0].out_off ? geo[3 * (size_t)b0 + 1] : geo[3 * (size_t)b0] | (p - blocks[b0].out_off);
    const uint32_t b1 = block_of(blocks, n_blocks, e - 1);
    voff_end[j] = e == blocks[b1].out_off + blocks[b1].out_len ? geo[3 * (size_t)b1 + 2] : geo[3 * (size_t)b1] | (e - blocks[b1].out_off);
}

__global__ void k_read_begin(const uint32_t* __restrict__ first_seg, uint32_t nc, const uint64_t* __restrict__ rec_base, const uint32_t* __restrict__ kidx,
                             uint64_t n_rec, uint64_t* __restrict__ read_begin) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > nc) return;
    // first_seg[c] = index of the first segment of contig c (n_segs for c == nc or a contig without records: then the next one's)
    const uint64_t r = rec_base[first_seg[c]];
    read_begin[c] = r < n_rec ? kidx[r] : kidx[n_rec];
}

inline unsigned nblk(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

namespace np1ingest {

struct Scratch {
    DevBuf comp, inflated, blocks, status, segs, counts, rec_base, first_seg, small, scan_tmp, rec_off, rec_seg, keep, kidx, ncw, seqb, qualb, cig_at, seq_at, qual_at, geo, voff,
        voff_end, crc_shift, lane_tables;
    std::vector<uint32_t> h_status;
    uint64_t n_host_blocks = 0;
    // HIP-event times of the block decoder and the CRC pass and the bytes they moved, summed over the batches of this lane (bench.py's
    // roofline of the from-files leg: np1_pipe_ingest_stats)
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    double inflate_ms = 0, crc_ms = 0;
    uint64_t comp_bytes = 0, inflated_bytes = 0, launches = 0;
    ~Scratch() {
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        DevBuf* all[] = {&comp, &inflated, &blocks, &status, &segs, &counts, &rec_base, &first_seg, &small, &scan_tmp, &rec_off, &rec_seg, &keep, &kidx, &ncw,
                         &seqb, &qualb, &cig_at, &seq_at, &qual_at, &geo, &voff, &voff_end, &crc_shift, &lane_tables};
        for (DevBuf* b : all) b->release();
    }
};
Scratch* scratch_create() { return new Scratch(); }
void scratch_destroy(Scratch* s) { delete s; }
uint64_t scratch_host_blocks(const Scratch* s) { return s ? s->n_host_blocks : 0; }
void scratch_stats(const Scratch* s, double out[5]) {
    if (!s) return;
    out[0] += s->inflate_ms; out[1] += s->crc_ms; out[2] += (double)s->comp_bytes; out[3] += (double)s->inflated_bytes; out[4] += (double)s->launches;
}
void scratch_stats_reset(Scratch* s) { if (s) { s->inflate_ms = s->crc_ms = 0; s->comp_bytes = s->inflated_bytes = s->launches = 0; } }

// pageable host memory that only grows (the draft strings of a batch: they leave through npcopy::h2d, which takes the bytes at call time)
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        free(p);
        cap = 0;
        p = malloc(bytes + bytes / 4 + 4096);
        if (!p) return false;
        cap = bytes + bytes / 4 + 4096;
        return true;
    }
    ~HostBuf() { free(p); }
};

// a run of file bytes the batch needs: [coff, coff + bytes) of the BAM lands at [comp_at, comp_at + bytes) of the device's compressed buffer
struct FileRun { uint64_t coff, bytes, comp_at, last_block; };   // last_block: highest file offset at which a needed block may start

struct Staging::Impl {
    // Round 6: the compressed bytes of a batch are no longer copied into pinned memory of the staging object.  The cold start of the
    // from-files CLI was page-locking: a staging object grew to the compressed size of the largest batch it met (2.2 GB for a 250 Mb contig
    // at 30x) at ~1.6 GB/s of hipHostMalloc, every loader thread its own, every growth from scratch -- 5 of the 8 s of a 3 Gb run.  Now
    // prepare() walks the BGZF headers in the mapped file (no bulk read), and ingest() sends the runs to the device through the small ring
    // of pinned slots every large pageable copy already uses (np_hostcopy.h): pread() straight into a slot on the helper threads, DMA out.
    const BamSource* src = nullptr;
    std::vector<FileRun> runs;
    HostBuf draft;
    std::vector<npdev::BlockDesc> blocks;
    std::vector<uint64_t> block_coff;          // file offset of every block, ascending
    std::vector<uint32_t> block_size;          // its size in the file
    std::vector<uint64_t> block_geo;           // 3 per block: file offset << 16, bgzf_tell at its first byte, (file offset + size) << 16
    std::vector<int32_t> tid;                  // BAM reference id of every contig (-1: not in the header)
    std::vector<Segment> segs;
    std::vector<uint32_t> first_seg;           // nc + 1
    std::vector<uint32_t> ctg_off;             // nc + 1
    std::vector<int32_t> ctg_len;
    std::vector<std::string> names;
    uint64_t comp_bytes = 0, inflated_bytes = 0;
};

Staging::Staging() : impl(new Impl()) {}
Staging::~Staging() { delete impl; }
const std::vector<std::string>& Staging::names() const { return impl->names; }
uint64_t Staging::compressed_bytes() const { return impl->comp_bytes; }

// Host half: FASTA strings, the compressed extents of the batch's contigs into pinned memory, the block table, the anchors.
// Returns 1 when this batch cannot take the device path (index without the per-contig offsets), 0 on success, -1 on error.
int prepare(BamSource& src, const std::vector<std::string>& names, Staging* st, std::string* err) {
    static const bool timing = getenv("NP1_TIMING") != nullptr;
    auto now_ms = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_begin = now_ms();
    double t_draft = 0, t_alloc = 0, t_read = 0;
    Staging::Impl& S = *st->impl;
    S.names = names;
    S.blocks.clear(); S.block_coff.clear(); S.block_size.clear(); S.segs.clear(); S.first_seg.clear(); S.ctg_off.assign(1, 0); S.ctg_len.clear();
    const size_t nc = names.size();
    // ---- drafts
    std::vector<int> fid(nc), tid(nc);
    size_t draft_total = 0;
    for (size_t c = 0; c < nc; ++c) {
        fid[c] = src.fai.find(names[c]);
        if (fid[c] < 0) { *err = "contig not in FASTA index: " + names[c]; return -1; }
        draft_total += (size_t)src.fai.entry(fid[c]).len;
        tid[c] = src.hdr.name2id(names[c]);
    }
    if (draft_total >= 0xfff00000ull) { *err = "batch too large: draft must stay below 2^32 slots"; return -1; }
    { const double t0 = now_ms(); if (!S.draft.ensure(draft_total + 64)) { *err = "out of host memory"; return -1; } t_alloc += now_ms() - t0; }
    const double t_d0 = now_ms();
    std::string seq;
    size_t at = 0;
    for (size_t c = 0; c < nc; ++c) {
        if (!src.fai.fetch(fid[c], &seq)) { *err = "cannot fetch contig: " + names[c]; return -1; }
        memcpy((char*)S.draft.p + at, seq.data(), seq.size());
        at += seq.size();
        S.ctg_len.push_back((int32_t)seq.size());
        S.ctg_off.push_back((uint32_t)at);
    }
    t_draft = now_ms() - t_d0;
    // ---- compressed extents: [block of the contig's first record, block of the end of its last record]
    std::vector<std::pair<np::voff_t, np::voff_t>> vr(nc, {0, 0});
    for (size_t c = 0; c < nc; ++c) {
        if (tid[c] < 0 || tid[c] >= (int)src.bai.refs.size()) continue;
        const np::BaiRef& r = src.bai.refs[(size_t)tid[c]];
        auto it = r.bins.find(37450u);
        if (it == r.bins.end() || it->second.size() < 2) {
            if (r.bins.empty()) continue;      // no records at all
            return 1;                          // an index without the metadata pseudo-bin: host loader
        }
        if (it->second[0].end > it->second[0].beg) vr[c] = {it->second[0].beg, it->second[0].end};
    }
    // ---- the file bytes of the contigs' record ranges, a few large reads into pinned memory, then the BGZF headers in memory
    // byte range of contig c: from the block of its first record up to (and with) the block that holds the end of its last one;
    // that block's size is not known before its header is read, so up to 64 KiB more are taken (never beyond the file)
    std::vector<size_t> order;
    for (size_t c = 0; c < nc; ++c) if (vr[c].second > vr[c].first) order.push_back(c);
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return vr[a].first < vr[b].first; });
    typedef FileRun Run;
    std::vector<Run>& runs = S.runs;
    runs.clear();
    S.src = &src;
    if (!src.map) return 1;      // (the file could not be mapped: host loader)
    for (size_t c : order) {
        const uint64_t c0 = vr[c].first >> 16;
        const bool partial = (vr[c].second & 0xffffu) != 0;
        const uint64_t last = partial ? (vr[c].second >> 16) : (vr[c].second >> 16) - 1;    // no block starts inside another: "< ve's block" == "<= ve's block - 1"
        uint64_t end = partial ? (vr[c].second >> 16) + 65536 : (vr[c].second >> 16);
        if (end > src.file_size) end = src.file_size;
        if (!runs.empty() && c0 <= runs.back().coff + runs.back().bytes) {
            Run& r = runs.back();
            if (end > r.coff + r.bytes) r.bytes = end - r.coff;
            if (last > r.last_block) r.last_block = last;
        } else {
            runs.push_back(Run{c0, end - c0, 0, last});
        }
    }
    uint64_t comp_at = 0, u_at = 0;
    for (Run& r : runs) { r.comp_at = comp_at; comp_at += r.bytes; }
    const double t_r0 = now_ms();
    for (const Run& r : runs) {
        uint64_t p = 0;
        while (r.coff + p <= r.last_block) {
            if (p + 18 > r.bytes) { *err = "BAM truncated inside the indexed range"; return -1; }
            const uint8_t* h = src.map + r.coff + p;      // (the header walk touches two pages of the mapped file per block)
            if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { *err = "not a BGZF block where the index points"; return -1; }
            const uint32_t xlen = h[10] | (h[11] << 8);
            if (p + 12 + xlen > r.bytes) { *err = "BAM truncated inside the indexed range"; return -1; }
            uint32_t bsize = 0;
            for (uint32_t i = 0; i + 4 <= xlen;) {
                const uint8_t* x = h + 12 + i;
                const uint32_t slen = x[2] | (x[3] << 8);
                if (x[0] == 'B' && x[1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = (uint32_t)(x[4] | (x[5] << 8)) + 1;
                i += 4 + slen;
            }
            if (bsize < 12 + xlen + 8 || p + bsize > r.bytes) { *err = "bad BGZF block size"; return -1; }
            uint32_t isize;
            memcpy(&isize, h + bsize - 4, 4);
            if (isize > 65536) { *err = "BGZF block larger than 64 KiB"; return -1; }
            npdev::BlockDesc d;
            d.in_off = r.comp_at + p + 12 + xlen;
            d.in_len = bsize - (12 + xlen) - 8;
            d.out_off = u_at;
            d.out_len = isize;
            S.blocks.push_back(d);
            S.block_coff.push_back(r.coff + p);
            S.block_size.push_back(bsize);
            u_at += isize;
            p += bsize;
        }
    }
    t_read = now_ms() - t_r0;
    S.comp_bytes = comp_at;
    S.inflated_bytes = u_at;
    S.tid.assign(tid.begin(), tid.end());
    S.block_geo.resize(3 * S.blocks.size());
    for (size_t i = 0; i < S.blocks.size(); ++i) {
        const uint64_t here = S.block_coff[i] << 16;
        // the reader's position at the first byte of block i: behind an empty block of the same run it still reports that block's start
        const bool run_start = i == 0 || S.block_coff[i - 1] + S.block_size[i - 1] != S.block_coff[i];
        S.block_geo[3 * i] = here;
        S.block_geo[3 * i + 1] = (!run_start && S.blocks[i - 1].out_len == 0) ? S.block_geo[3 * (i - 1) + 1] : here;
        S.block_geo[3 * i + 2] = (S.block_coff[i] + S.block_size[i]) << 16;
    }
    // ---- anchors -> segments (inflated-stream offsets)
    auto to_u = [&](np::voff_t v, uint64_t* out) {
        const uint64_t coff = v >> 16;
        auto it = std::lower_bound(S.block_coff.begin(), S.block_coff.end(), coff);
        if (it == S.block_coff.end() || *it != coff) {
            // the position right behind a block that was read (the end of a contig's last record at a block boundary)
            if ((v & 0xffffu) == 0 && it != S.block_coff.begin()) {   // (binary search works: block_coff is ascending, runs are in file order)
                const size_t pb = (size_t)(it - S.block_coff.begin()) - 1;
                const npdev::BlockDesc& d = S.blocks[pb];
                if (S.block_coff[pb] + S.block_size[pb] == coff) { *out = d.out_off + d.out_len; return true; }
            }
            return false;
        }
        const size_t b = (size_t)(it - S.block_coff.begin());
        if ((v & 0xffffu) > S.blocks[b].out_len) return false;
        *out = S.blocks[b].out_off + (v & 0xffffu);
        return true;
    };
    S.first_seg.assign(nc + 1, 0);
    std::vector<uint64_t> anchors;
    for (size_t c = 0; c < nc; ++c) {
        S.first_seg[c] = (uint32_t)S.segs.size();
        if (vr[c].second <= vr[c].first) continue;
        uint64_t ub, ue;
        if (!to_u(vr[c].first, &ub) || !to_u(vr[c].second, &ue)) { *err = "BAM index points outside the blocks read"; return -1; }
        anchors.clear();
        anchors.push_back(ub);
        const np::BaiRef& r = src.bai.refs[(size_t)tid[c]];
        for (np::voff_t v : r.linear) {
            if (v <= vr[c].first || v >= vr[c].second) continue;
            uint64_t x;
            if (to_u(v, &x) && x > ub && x < ue) anchors.push_back(x);
        }
        std::sort(anchors.begin(), anchors.end());
        anchors.erase(std::unique(anchors.begin(), anchors.end()), anchors.end());
        for (size_t a = 0; a < anchors.size(); ++a)
            S.segs.push_back(Segment{anchors[a], a + 1 < anchors.size() ? anchors[a + 1] : ue, (uint32_t)c, (int32_t)tid[c], S.ctg_len[c], 0});
    }
    S.first_seg[nc] = (uint32_t)S.segs.size();
    if (timing)
        fprintf(stderr, "[np1 staging] %zu contigs, %.1f MB of draft, %.1f MB compressed | ms: allocations %.1f  draft fetch %.1f  header walk %.1f  segments %.1f\n",
                nc, draft_total / 1e6, S.comp_bytes / 1e6, t_alloc, t_draft, t_read, now_ms() - t_begin - t_alloc - t_draft - t_read);
    return 0;
}

// The runs' bytes of the file -> the device buffer, through the ring of pinned slots (np_hostcopy.h): rounds of up to kWave pieces of one slot
// each, every piece read by pread() on a helper thread STRAIGHT INTO ITS SLOT (one host copy, the kernel's), then this thread -- the lane's,
// the only one that talks to the runtime -- enqueues the DMA of each slot on q and gives it back with its event.  The ring holds twice a
// round's slots, so the DMA of round k overlaps the reads of round k + 1.
static int stream_runs_to_device(int fd, const std::vector<FileRun>& runs, uint8_t* dst, hipStream_t q) {
    struct Piece { uint64_t at, coff; size_t bytes; };
    std::vector<Piece> pieces;
    for (const FileRun& r : runs)
        for (uint64_t done = 0; done < r.bytes; done += npcopy::kSlotBytes)
            pieces.push_back(Piece{r.comp_at + done, r.coff + done, (size_t)std::min<uint64_t>(npcopy::kSlotBytes, r.bytes - done)});
    npcopy::Ring& R = npcopy::ring();
    const size_t kWave = std::max<size_t>(1, std::min<size_t>(npcopy::kSlots / 2, np::host_threads()));
    std::vector<npcopy::Slot> slot(kWave);
    for (size_t p0 = 0; p0 < pieces.size();) {
        const size_t n = std::min(kWave, pieces.size() - p0);
        size_t have = 0;
        for (; have < n; ++have)      // (the first of a round may be waited for; further ones only if they can be had without waiting for other threads)
            if (!(have == 0 ? R.acquire(&slot[have]) : R.try_acquire(&slot[have]))) break;
        if (have == 0) return -1;
        std::atomic<bool> ok{true};
        np::parallel_for(have, 1, [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) {
                const Piece& pc = pieces[p0 + k];
                size_t done = 0;
                while (done < pc.bytes) {
                    const ssize_t g = pread(fd, (char*)slot[k].p + done, pc.bytes - done, (off_t)(pc.coff + done));
                    if (g <= 0) { ok = false; break; }
                    done += (size_t)g;
                }
            }
        });
        hipError_t e = ok ? hipSuccess : hipErrorUnknown;
        for (size_t k = 0; k < have; ++k) {
            if (e == hipSuccess) e = hipMemcpyAsync(dst + pieces[p0 + k].at, slot[k].p, pieces[p0 + k].bytes, hipMemcpyHostToDevice, q);
            if (e == hipSuccess) e = hipEventRecord(slot[k].ev, q);
            slot[k].pending = e == hipSuccess;
        }
        if (e != hipSuccess) (void)hipStreamSynchronize(q);
        for (size_t k = 0; k < have; ++k) R.release(slot[k]);
        if (e != hipSuccess) return -1;
        p0 += have;      // (fewer slots than pieces this round: the rest next time)
    }
    return 0;
}

// Device half: fills the batch object.  Returns 0, 1 (take the host loader for this batch: CG-tag CIGARs) or -1.
static double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

int ingest(np1_batch* b, Staging* st, bool with_qual, Scratch* scr, const np::BaiIndex* replay_bai) {
    Staging::Impl& S = *st->impl;
    static const bool timing = getenv("NP1_TIMING") != nullptr;
    const double t_0 = timing ? now_ms() : 0;
    double t_1 = 0, t_2 = 0;
    np1_ctx* ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    hipStream_t q = ctx->stream;
    const uint32_t nc = (uint32_t)S.names.size();
    const uint32_t n_blocks = (uint32_t)S.blocks.size(), n_segs = (uint32_t)S.segs.size();
    b->nc = nc;
    b->G = S.ctg_off.back();
    b->h_ctg_off = S.ctg_off;
    b->max_lq = 0;
    b->replay.on = false;
    b->force_staged = false;
    b->ran = false;
    b->out_cached = false;
    b->out_pinned = false;
    Scratch& W = *scr;
    if (b->draft.ensure(b->G + 64) || b->ctg_off.ensure(4 * (size_t)(nc + 1)) || b->read_begin.ensure(8 * (size_t)(nc + 2)) ||
        W.comp.ensure(S.comp_bytes + 4096) || W.inflated.ensure(S.inflated_bytes + 2 * CHASE_WIN + 4096) || W.blocks.ensure(sizeof(npdev::BlockDesc) * (size_t)(n_blocks + 1)) ||
        W.status.ensure(4 * (size_t)(n_blocks + 1)) || W.segs.ensure(sizeof(Segment) * (size_t)(n_segs + 1)) || W.counts.ensure(4 * (size_t)(n_segs + 2)) ||
        W.rec_base.ensure(8 * (size_t)(n_segs + 2)) || W.first_seg.ensure(4 * (size_t)(nc + 2)) || W.small.ensure(256) ||
        W.scan_tmp.ensure(8 * (np1k::scan_tmp_words((uint64_t)n_segs + 1) + 8)))
        return -1;
    if (b->G) HIPCHK(npcopy::h2d(b->draft.p, S.draft.p, b->G, q));
    HIPCHK(npcopy::h2d(b->ctg_off.p, S.ctg_off.data(), 4 * (size_t)(nc + 1), q));
    if (S.comp_bytes) {
        if (stream_runs_to_device(S.src->fd, S.runs, W.comp.as<uint8_t>(), q) != 0) { np1_set_error("BAM read failed"); return -1; }
        HIPCHK(hipMemsetAsync(W.comp.as<uint8_t>() + S.comp_bytes, 0, 4096, q));
    }
    if (n_blocks) HIPCHK(npcopy::h2d(W.blocks.p, S.blocks.data(), sizeof(npdev::BlockDesc) * (size_t)n_blocks, q));
    if (n_segs) HIPCHK(npcopy::h2d(W.segs.p, S.segs.data(), sizeof(Segment) * (size_t)n_segs, q));
    HIPCHK(npcopy::h2d(W.first_seg.p, S.first_seg.data(), 4 * (size_t)(nc + 1), q));
    HIPCHK(hipMemsetAsync(W.small.p, 0, 256, q));
    uint32_t* d_err = W.small.as<uint32_t>();           // [0] error bits, [1] max l_qseq
    uint64_t* d_tot = reinterpret_cast<uint64_t*>(W.small.as<uint8_t>() + 64);   // scan totals
    uint64_t n_rec = 0;
    if (n_blocks) {
        // which decoder: a lane per block with its tables in LDS (np_inflate_lds.h; round 6: 7 / 5-bit primaries, three waves per CU -- measured
        // 14.4 ms for 34 k blocks, 45.5 ms for 138 k, against 42.6 / 94.8 ms of the round-4 lane decoder: profiles/r6_inflate_lds.txt) when the
        // batch has blocks enough to fill the chip with lanes, else a wave per block (np_inflate_dev.h).
        // NP1_INFLATE=lds<LB><DB> | lanes (np_inflate_lane.h, tables in HBM) | wave forces one.
        static const int mode = [] {
            const char* e = getenv("NP1_INFLATE");
            return !e ? 0 : strcmp(e, "lanes") == 0 ? 1 : strcmp(e, "wave") == 0 ? 2 : lds_variant(e) ? 3 : 0;
        }();
        static const int variant = lds_variant(getenv("NP1_INFLATE")) ? lds_variant(getenv("NP1_INFLATE")) : 75;
        // lanes in flight: a multiple of the wave, at least one wave, at most 2^18 (their tables are ~15 KB each in HBM)
        static const uint32_t max_lanes = [] {
            long v = getenv("NP1_INFLATE_LANES") ? atol(getenv("NP1_INFLATE_LANES")) : 131072;
            if (v < 64) v = 64;
            if (v > 262144) v = 262144;
            return (uint32_t)((v + 63) / 64 * 64);
        }();
        for (hipEvent_t& e : W.ev) if (!e) (void)hipEventCreate(&e);
        (void)hipEventRecord(W.ev[0], q);
        if (mode == 3 || (mode == 0 && n_blocks >= 4096u)) {
            if (launch_inflate_lds_variant(variant, q, W.comp.as<uint8_t>(), W.blocks.as<npdev::BlockDesc>(), n_blocks, W.inflated.as<uint8_t>(),
                                           W.status.as<uint32_t>(), W.lane_tables)) return -1;
        } else if (mode == 1) {
            const uint32_t lanes = std::min<uint32_t>((n_blocks + 63u) & ~63u, max_lanes);
            if (W.lane_tables.ensure((size_t)lanes * nplane::LANE_TABLE_WORDS * 4)) return -1;
            k_inflate_lanes<<<lanes / 64, 64, 0, q>>>(W.comp.as<uint8_t>(), W.blocks.as<npdev::BlockDesc>(), n_blocks, W.inflated.as<uint8_t>(), W.status.as<uint32_t>(),
                                                       W.lane_tables.as<uint32_t>());
        } else
        k_inflate<<<nblk(n_blocks, 4), 256, 0, q>>>(W.comp.as<uint8_t>(), W.blocks.as<npdev::BlockDesc>(), n_blocks, W.inflated.as<uint8_t>(), W.status.as<uint32_t>());
        (void)hipEventRecord(W.ev[1], q);
        static const bool check_crc = getenv("NP_BGZF_NO_CRC") == nullptr;      // the same switch as the host reader's (np_bgzf.cpp)
        if (check_crc) {
            if (!W.crc_shift.p) {
                uint32_t t[64];
                npdev::crc_shift_table(t);
                if (W.crc_shift.ensure(sizeof(t))) return -1;
                HIPCHK(npcopy::h2d_sync(W.crc_shift.p, t, sizeof(t)));
            }
            k_crc_check<<<nblk(n_blocks, 4), 256, 0, q>>>(W.comp.as<uint8_t>(), W.blocks.as<npdev::BlockDesc>(), n_blocks, W.inflated.as<uint8_t>(), W.status.as<uint32_t>(),
                                                         W.crc_shift.as<uint32_t>());
        }
        (void)hipEventRecord(W.ev[2], q);
        W.h_status.resize(n_blocks);
        HIPCHK(npcopy::d2h(W.h_status.data(), W.status.p, 4 * (size_t)n_blocks, q));
    }
    if (n_segs) {
        k_chase<false><<<nblk(n_segs, 4), 256, 0, q>>>(W.inflated.as<uint8_t>(), W.segs.as<Segment>(), n_segs, W.counts.as<uint32_t>(), nullptr, nullptr, nullptr, d_err);
        np1k::launch_scan_rows(q, W.counts.as<uint32_t>(), n_segs, W.rec_base.as<uint64_t>(), W.scan_tmp.as<uint64_t>(), &d_tot[0]);
        HIPCHK(npcopy::d2h(&n_rec, &d_tot[0], 8, q));
    } else {
        HIPCHK(hipMemsetAsync(W.rec_base.p, 0, 16, q));
    }
    HIPCHK(hipStreamSynchronize(q));
    if (timing) t_1 = now_ms();
    if (n_blocks) {
        float a = 0, c = 0;
        if (hipEventElapsedTime(&a, W.ev[0], W.ev[1]) == hipSuccess && hipEventElapsedTime(&c, W.ev[1], W.ev[2]) == hipSuccess) {
            W.inflate_ms += a; W.crc_ms += c; W.comp_bytes += S.comp_bytes; W.inflated_bytes += S.inflated_bytes; ++W.launches;
        }
    }
    // blocks the device decoder did not accept: inflate them on the host and patch them in, then redo the count
    bool patched = false;
    for (uint32_t i = 0; i < n_blocks; ++i) {
        if (!W.h_status[i]) continue;
        const npdev::BlockDesc& d = S.blocks[i];
        std::vector<uint8_t> tmp(d.out_len ? d.out_len : 1);
        const uint8_t* payload = S.src->map + S.block_coff[i] + (S.block_size[i] - d.in_len - 8u);      // (in the mapped file: header, payload, CRC + ISIZE)
        if (!np::bgzf_inflate_block(payload, d.in_len, tmp.data(), d.out_len)) { np1_set_error("corrupt BGZF block in the BAM"); return -1; }
        if (getenv("NP_BGZF_NO_CRC") == nullptr && d.out_len) {     // blocks that come back to the host are checked here (the device checked the others)
            uint32_t want;
            memcpy(&want, payload + d.in_len, 4);
            if (np::crc32_block(tmp.data(), d.out_len) != want) { np1_set_error("corrupt BGZF block in the BAM (CRC mismatch)"); return -1; }
        }
        HIPCHK(npcopy::h2d_sync(W.inflated.as<uint8_t>() + d.out_off, tmp.data(), d.out_len));
        patched = true;
        ++W.n_host_blocks;
    }
    if (patched && n_segs) {
        HIPCHK(hipMemsetAsync(d_err, 0, 4, q));
        k_chase<false><<<nblk(n_segs, 4), 256, 0, q>>>(W.inflated.as<uint8_t>(), W.segs.as<Segment>(), n_segs, W.counts.as<uint32_t>(), nullptr, nullptr, nullptr, d_err);
        np1k::launch_scan_rows(q, W.counts.as<uint32_t>(), n_segs, W.rec_base.as<uint64_t>(), W.scan_tmp.as<uint64_t>(), &d_tot[0]);
        HIPCHK(npcopy::d2h(&n_rec, &d_tot[0], 8, q));
        HIPCHK(hipStreamSynchronize(q));
    }
    if (n_rec >= 0xfffffff0ull) { np1_set_error("batch too large: more than 2^32 records"); return -1; }
    const size_t nr = (size_t)(n_rec ? n_rec : 1);
    if (W.rec_off.ensure(8 * nr) || W.rec_seg.ensure(4 * nr) || W.keep.ensure(4 * (nr + 1)) || W.kidx.ensure(4 * (nr + 2)) || W.ncw.ensure(4 * (nr + 1)) ||
        W.seqb.ensure(4 * (nr + 1)) || W.qualb.ensure(4 * (nr + 1)) || W.cig_at.ensure(8 * (nr + 2)) || W.seq_at.ensure(8 * (nr + 2)) ||
        W.qual_at.ensure(8 * (nr + 2)) || W.scan_tmp.ensure(8 * (np1k::scan_tmp_words((uint64_t)nr + 1) + np1k::scan_tmp_words((uint64_t)n_segs + 1) + 8)))
        return -1;
    uint64_t totals[4] = {0, 0, 0, 0};   // kept records, cigar words, seq bytes, qual bytes
    uint32_t h_small[2] = {0, 0};
    if (n_rec) {
        k_chase<true><<<nblk(n_segs, 4), 256, 0, q>>>(W.inflated.as<uint8_t>(), W.segs.as<Segment>(), n_segs, nullptr, W.rec_base.as<uint64_t>(), W.rec_off.as<uint64_t>(),
                                                     W.rec_seg.as<uint32_t>(), d_err);
        k_rec_measure<<<nblk(n_rec, 256), 256, 0, q>>>(W.inflated.as<uint8_t>(), W.rec_off.as<uint64_t>(), W.rec_seg.as<uint32_t>(), W.segs.as<Segment>(), n_rec,
                                                      with_qual ? 1 : 0, W.keep.as<uint32_t>(), W.ncw.as<uint32_t>(), W.seqb.as<uint32_t>(), W.qualb.as<uint32_t>(),
                                                      d_err + 1, d_err);
        uint64_t* tmp = W.scan_tmp.as<uint64_t>();
        np1k::launch_scan_u32(q, W.keep.as<uint32_t>(), n_rec, W.kidx.as<uint32_t>(), tmp, &d_tot[1]);
        np1k::launch_scan_rows(q, W.ncw.as<uint32_t>(), n_rec, W.cig_at.as<uint64_t>(), tmp, &d_tot[2]);
        np1k::launch_scan_rows(q, W.seqb.as<uint32_t>(), n_rec, W.seq_at.as<uint64_t>(), tmp, &d_tot[3]);
        if (with_qual) np1k::launch_scan_rows(q, W.qualb.as<uint32_t>(), n_rec, W.qual_at.as<uint64_t>(), tmp, &d_tot[4]);
        HIPCHK(npcopy::d2h(totals, &d_tot[1], with_qual ? 32 : 24, q));
    } else {
        HIPCHK(hipMemsetAsync(W.kidx.p, 0, 8, q));
    }
    HIPCHK(npcopy::d2h(h_small, d_err, 8, q));
    HIPCHK(hipStreamSynchronize(q));
    if (timing) t_2 = now_ms();
    if (h_small[0] & IG_ERR_CGTAG) return 1;
    if (h_small[0] & (IG_ERR_CHAIN | IG_ERR_RECORD)) { np1_set_error("BAM records do not line up with the index (corrupt BAM or stale .bai)"); return -1; }
    const size_t n = (size_t)totals[0];
    b->n_reads = (int64_t)n;
    b->max_lq = h_small[1];
    const size_t nn = n ? n : 1;
    if (b->pos.ensure(4 * nn) || b->ctg.ensure(4 * nn) || b->flag.ensure(2 * nn) || b->ncig.ensure(4 * nn) || b->lq.ensure(4 * nn) || b->cigoff.ensure(8 * nn) ||
        b->seqoff.ensure(8 * nn) || b->cigar.ensure(4 * (size_t)totals[1] + 64) || b->seq.ensure((size_t)totals[2] + 64))
        return -1;
    b->has_qual = with_qual;
    if (with_qual && (b->mapq.ensure(nn) || b->isize.ensure(4 * nn) || b->qualoff.ensure(8 * nn) || b->qual.ensure((size_t)totals[3] + 64))) return -1;
    if (n_rec) {
        ScatterOut o{b->pos.as<int32_t>(), b->ctg.as<uint32_t>(), b->flag.as<uint16_t>(), b->ncig.as<uint32_t>(), b->lq.as<int32_t>(), b->cigoff.as<uint64_t>(),
                     b->seqoff.as<uint64_t>(), b->cigar.as<uint32_t>(), b->seq.as<uint8_t>(), b->mapq.as<uint8_t>(), b->isize.as<int32_t>(),
                     b->qualoff.as<uint64_t>(), b->qual.as<uint8_t>()};
        k_rec_scatter<<<nblk(n_rec, 256), 256, 0, q>>>(W.inflated.as<uint8_t>(), W.rec_off.as<uint64_t>(), W.rec_seg.as<uint32_t>(), W.segs.as<Segment>(), n_rec,
                                                      W.keep.as<uint32_t>(), W.kidx.as<uint32_t>(), W.cig_at.as<uint64_t>(), W.seq_at.as<uint64_t>(),
                                                      W.qual_at.as<uint64_t>(), with_qual ? 1 : 0, o);
    }
    k_read_begin<<<nblk(nc + 1, 64), 64, 0, q>>>(W.first_seg.as<uint32_t>(), nc, W.rec_base.as<uint64_t>(), W.kidx.as<uint32_t>(), n_rec, b->read_begin.as<uint64_t>());
    b->h_read_begin.resize((size_t)nc + 1);
    HIPCHK(npcopy::d2h(b->h_read_begin.data(), b->read_begin.p, 8 * (size_t)(nc + 1), q));
    if (replay_bai) {   // kmer_count / snp_valid replay the reference's region iterator: the records' virtual offsets come down with the batch
        np1_batch::Replay& R = b->replay;
        if (W.geo.ensure(8 * (S.block_geo.size() + 3)) || W.voff.ensure(8 * nn) || W.voff_end.ensure(8 * nn)) return -1;
        if (!R.own_voff.ensure(8 * nn) || !R.own_voff_end.ensure(8 * nn)) { np1_set_error("hipHostMalloc failed"); return -1; }
        if (n_rec) {
            HIPCHK(npcopy::h2d(W.geo.p, S.block_geo.data(), 8 * S.block_geo.size(), q));
            k_rec_voff<<<nblk(n_rec, 256), 256, 0, q>>>(W.inflated.as<uint8_t>(), W.rec_off.as<uint64_t>(), n_rec, W.keep.as<uint32_t>(), W.kidx.as<uint32_t>(),
                                                       W.blocks.as<npdev::BlockDesc>(), W.geo.as<uint64_t>(), n_blocks, W.voff.as<uint64_t>(), W.voff_end.as<uint64_t>());
            HIPCHK(npcopy::d2h(R.own_voff.p, W.voff.p, 8 * n, q));
            HIPCHK(npcopy::d2h(R.own_voff_end.p, W.voff_end.p, 8 * n, q));
        }
    }
    HIPCHK(hipStreamSynchronize(q));
    if (replay_bai && np1_batch_enable_replay_ingested(b, replay_bai, S.tid) != 0) return -1;
    b->input_bytes = b->G + 32 * n + 4 * (size_t)totals[1] + (size_t)totals[2];
    if (timing)
        fprintf(stderr, "[np1 ingest] %.1f MB compressed, %.1f MB inflated, %u blocks, %u segments, %llu records (%zu kept) | ms: h2d+inflate+count %.2f  offsets+measure+scans %.2f  scatter %.2f\n",
                S.comp_bytes / 1e6, S.inflated_bytes / 1e6, n_blocks, n_segs, (unsigned long long)n_rec, n, t_1 - t_0, t_2 - t_1, now_ms() - t_2);
    return 0;
}

bool BamSource::open(const std::string& fasta, const std::string& bam, std::string* err) {
    if (!fai.load(fasta)) { *err = "cannot load FASTA/index: " + fasta; return false; }
    np::BamReader rd;
    if (!rd.open(bam)) { *err = "cannot open BAM: " + bam; return false; }
    hdr = rd.header();
    have_bai = bai.load(bam + ".bai");
    fd = ::open(bam.c_str(), O_RDONLY);
    if (fd < 0) { *err = "cannot open BAM: " + bam; return false; }
    const off_t sz = lseek(fd, 0, SEEK_END);
    file_size = sz > 0 ? (uint64_t)sz : 0;
    if (file_size) {      // read-only view of the file for the BGZF header walk and the few blocks the host inflates itself (a failure only costs the device path)
        void* m = mmap(nullptr, (size_t)file_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) map = static_cast<const uint8_t*>(m);      // (default read-around: a fault brings the next header's page along)
    }
    return true;
}
BamSource::~BamSource() {
    if (map) (void)munmap(const_cast<uint8_t*>(map), (size_t)file_size);
    if (fd >= 0) ::close(fd);
}

}  // namespace np1ingest

// test / diagnostics hook: inflates a buffer of concatenated BGZF blocks on the device; out receives the inflated bytes,
// status one word per block (0 = accepted).  Returns the number of blocks or -1.
extern "C" int64_t np1_debug_inflate_device_prof(int device, const uint8_t* bgzf, uint64_t n, uint8_t* out, uint64_t out_cap, uint32_t* status, int64_t status_cap,
                                                 unsigned long long* prof /* 8 words or NULL */, float* kernel_ms);
extern "C" int64_t np1_debug_inflate_device(int device, const uint8_t* bgzf, uint64_t n, uint8_t* out, uint64_t out_cap, uint32_t* status, int64_t status_cap) {
    return np1_debug_inflate_device_prof(device, bgzf, n, out, out_cap, status, status_cap, nullptr, nullptr);
}
extern "C" int64_t np1_debug_inflate_device_prof(int device, const uint8_t* bgzf, uint64_t n, uint8_t* out, uint64_t out_cap, uint32_t* status, int64_t status_cap,
                                                 unsigned long long* prof, float* kernel_ms) {
    if (hipSetDevice(device) != hipSuccess) { np1_set_error("hipSetDevice failed"); return -1; }
    std::vector<npdev::BlockDesc> blocks;
    uint64_t p = 0, u = 0;
    while (p + 18 <= n) {
        const uint8_t* h = bgzf + p;
        if (h[0] != 31 || h[1] != 139) { np1_set_error("not BGZF"); return -1; }
        const uint32_t xlen = h[10] | (h[11] << 8);
        uint32_t bsize = 0;
        for (uint32_t i = 0; i + 4 <= xlen;) {
            const uint8_t* x = h + 12 + i;
            const uint32_t slen = x[2] | (x[3] << 8);
            if (x[0] == 'B' && x[1] == 'C' && slen == 2) bsize = (uint32_t)(x[4] | (x[5] << 8)) + 1;
            i += 4 + slen;
        }
        if (!bsize || p + bsize > n) { np1_set_error("truncated BGZF"); return -1; }
        uint32_t isize;
        memcpy(&isize, h + bsize - 4, 4);
        blocks.push_back(npdev::BlockDesc{p + 12 + xlen, u, bsize - (12 + xlen) - 8, isize});
        u += isize;
        p += bsize;
    }
    if (u > out_cap || (int64_t)blocks.size() > status_cap) { np1_set_error("output buffers too small"); return -1; }
    DevBuf dc, du, db, ds;
    if (dc.ensure(n + 4096) || du.ensure(u + 64) || db.ensure(sizeof(npdev::BlockDesc) * (blocks.size() + 1)) || ds.ensure(4 * (blocks.size() + 1))) return -1;
    HIPCHK(hipMemsetAsync(dc.p, 0, n + 4096, nullptr));
    HIPCHK(npcopy::h2d_sync(dc.p, bgzf, n));
    HIPCHK(npcopy::h2d_sync(db.p, blocks.data(), sizeof(npdev::BlockDesc) * blocks.size()));
    HIPCHK(hipMemsetAsync(du.p, 0xEE, u + 64, nullptr));
    DevBuf dp;
    if (dp.ensure(64)) return -1;
    HIPCHK(hipMemsetAsync(dp.p, 0, 64, nullptr));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    static const bool use_lanes = getenv("NP1_INFLATE") && strcmp(getenv("NP1_INFLATE"), "lanes") == 0;
    static const int use_lds = lds_variant(getenv("NP1_INFLATE"));
    DevBuf dt;
    if (use_lanes && !prof && dt.ensure((size_t)((blocks.size() + 63) & ~63ull) * nplane::LANE_TABLE_WORDS * 4)) return -1;
    if (!blocks.empty() && use_lds && !prof) {
        const int rc = launch_inflate_lds_variant(use_lds, nullptr, dc.as<uint8_t>(), db.as<npdev::BlockDesc>(), (uint32_t)blocks.size(), du.as<uint8_t>(), ds.as<uint32_t>(), dt);
        if (rc) return -1;
    } else if (!blocks.empty() && use_lanes && !prof) {
        const uint32_t lanes = (uint32_t)((blocks.size() + 63) & ~63ull);
        k_inflate_lanes<<<lanes / 64, 64>>>(dc.as<uint8_t>(), db.as<npdev::BlockDesc>(), (uint32_t)blocks.size(), du.as<uint8_t>(), ds.as<uint32_t>(), dt.as<uint32_t>());
    } else if (!blocks.empty()) {
        if (prof) k_inflate_prof<<<nblk(blocks.size(), 4), 256>>>(dc.as<uint8_t>(), db.as<npdev::BlockDesc>(), (uint32_t)blocks.size(), du.as<uint8_t>(), ds.as<uint32_t>(), dp.as<unsigned long long>());
        else k_inflate<<<nblk(blocks.size(), 4), 256>>>(dc.as<uint8_t>(), db.as<npdev::BlockDesc>(), (uint32_t)blocks.size(), du.as<uint8_t>(), ds.as<uint32_t>());
    }
    (void)hipEventRecord(e1, nullptr);
    HIPCHK(hipDeviceSynchronize());
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (prof) HIPCHK(npcopy::d2h_sync(prof, dp.p, 64));
    dp.release();
    if (u) HIPCHK(npcopy::d2h_sync(out, du.p, u));
    if (!blocks.empty()) HIPCHK(npcopy::d2h_sync(status, ds.p, 4 * blocks.size()));
    dc.release(); du.release(); db.release(); ds.release(); dt.release();
    return (int64_t)blocks.size();
}
