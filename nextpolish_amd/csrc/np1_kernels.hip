// HIP kernels of the short-read score_chain pass for gfx950 (CDNA4, wave64).
//
// The reference walks every read twice through htslib and keeps, per draft base, malloc'd lists of
// (3-base context, count) pairs, then runs a sequential fp64 chain DP over the whole contig
// (reference: source/lib/contig.c:170-496).  Here the same result is produced by
// data-parallel stages over a decoded record stream resident in HBM:
//
//   k_prep      1 lane / record : filter, trimmed query window, insertion-column max-reduce   (contig.c:202-245,333-358,667-677)
//   scan        draft -> slot offsets (a "slot" = a draft base or one insertion column after it)
//   k_slotinfo  1 lane / draft base : per-slot draft symbol + contig-boundary / lowercase bits    (contig.c:81-102,373-383)
//   k_rowcap    1 lane / record : row placement in slot space; scan -> row offsets
//   k_rows      1 lane / record : CIGAR walk -> the record's gapped row of 4-bit symbols          (contig.c:247-331)
//   k_vote      1 wave / 62 slots: per-slot context histogram in first-seen order, single-state
//               slots resolved at once, multi-state runs spilled as compact records               (base.c:60-71, contig.c:424-454)
//   k_dp        1 lane / multi-state run : exact fixed-point chain DP + backtrace                  (contig.c:424-496)
//   k_fixfirst  reference quirk: base 0 of a contig keeps its input state when it owns insertion columns
//   scan+k_emit polished characters, lowercase mask with the carried "sign"                       (contig.c:736-786)
//
// Integer / byte work throughout: no MFMA.  Bound: HBM bytes of the record stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "np1_core.h"
#include "np1_desc.h"
#include "np1_tile9.h"
#include "np1_kernels.h"

namespace np1k {

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep(ReadsDev R, int64_t n_reads, const uint32_t* __restrict__ ctg_off,
                                              int trim, int32_t* __restrict__ qs_out, int32_t* __restrict__ qe_out,
                                              int32_t* __restrict__ span_out, uint32_t* __restrict__ ins,
                                              uint32_t* __restrict__ counters) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    prep_record(R, r, ctg_off, trim, qs_out, qe_out, span_out, ins, counters);
}

// ------------------------------------------------------------------------------------------------
// exclusive scans (3 launches: tile reduce, tile-sum scan, tile scan + offset)
struct LoadInsPlus1 {
    const uint32_t* p;
    __device__ uint64_t operator()(uint64_t i) const { return 1ull + p[i]; }
};
struct LoadU32 {
    const uint32_t* p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
struct LoadKeep {
    const uint16_t* p;
    __device__ uint64_t operator()(uint64_t i) const { return (p[i] & 0xff) != 3 ? 1ull : 0ull; }
};

constexpr int SCAN_T = 256, SCAN_ITEMS = 16, SCAN_TILE = SCAN_T * SCAN_ITEMS;

__device__ __forceinline__ uint64_t block_reduce_sum(uint64_t v, uint64_t* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    uint64_t t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < SCAN_T / 64; ++i) t += sh[i];
    return t;   // valid in thread 0
}

template <class F>
__global__ __launch_bounds__(SCAN_T) void k_scan_reduce(F f, uint64_t n, uint64_t* __restrict__ tile_sums) {
    __shared__ uint64_t sh[SCAN_T / 64];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    uint64_t acc = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + (uint64_t)k * SCAN_T + threadIdx.x;
        if (i < n) acc += f(i);
    }
    uint64_t t = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = t;
}

// single block: in-place exclusive scan of the tile sums; total -> *total_out
__global__ __launch_bounds__(1024) void k_scan_tiles(uint64_t* __restrict__ tile_sums, uint64_t n_tiles,
                                                     uint64_t* __restrict__ total_out) {
    __shared__ uint64_t sh[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t b = 0; b < n_tiles; b += 1024) {
        uint64_t i = b + threadIdx.x;
        uint64_t v = i < n_tiles ? tile_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint64_t add = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        uint64_t incl = sh[threadIdx.x];
        uint64_t c = carry;
        if (i < n_tiles) tile_sums[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

template <class F, class OutT>
__global__ __launch_bounds__(SCAN_T) void k_scan_final(F f, uint64_t n, const uint64_t* __restrict__ tile_offs,
                                                       OutT* __restrict__ out, const uint64_t* __restrict__ total) {
    __shared__ uint64_t sh[SCAN_T];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    // thread t owns SCAN_ITEMS consecutive elements; they come in COALESCED through a padded tile in LDS (round 6: read from its own 64 bytes,
    // every load instruction of a wave touched 64 different lines; every element of every scan of this file is below 2^32)
    __shared__ uint32_t tile[SCAN_TILE + SCAN_TILE / 32];
    uint64_t first = base + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t j = (uint32_t)k * SCAN_T + threadIdx.x;
        const uint64_t i = base + j;
        tile[j + (j >> 5)] = i < n ? (uint32_t)f(i) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t j = threadIdx.x * SCAN_ITEMS + k;
        v[k] = tile[j + (j >> 5)];
        acc += v[k];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 1; o < SCAN_T; o <<= 1) {
        uint64_t add = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    uint64_t run = tile_offs[blockIdx.x] + sh[threadIdx.x] - acc;
    if constexpr (sizeof(OutT) == 4) {
        // Round 6: the results leave COALESCED.  A thread owns 16 consecutive elements, so stored from its registers every store instruction of a
        // wave wrote 4 bytes into each of 64 different lines; through a (padded) tile in LDS a wave writes 256 contiguous bytes per instruction.
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            const uint32_t j = threadIdx.x * SCAN_ITEMS + k;
            tile[j + (j >> 5)] = (uint32_t)run;      // (everybody has read its inputs out of the tile: the barriers of the block scan lie between)
            run += v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            const uint32_t j = (uint32_t)k * SCAN_T + threadIdx.x;
            const uint64_t i = base + j;
            if (i < n) out[i] = (OutT)tile[j + (j >> 5)];
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            uint64_t i = first + k;
            if (i < n) out[i] = (OutT)run;
            run += v[k];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (OutT)*total;
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t contig_of(const uint32_t* __restrict__ ctg_off, uint32_t nc, uint32_t g) {
    uint32_t lo = 0, hi = nc;   // largest c with ctg_off[c] <= g
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (ctg_off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_slotinfo(const uint8_t* __restrict__ draft, uint32_t G,
                                                  const uint32_t* __restrict__ ctg_off, uint32_t nc,
                                                  const uint32_t* __restrict__ soff, uint8_t* __restrict__ slot_info,
                                                  uint32_t* __restrict__ slot_g) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    uint32_t c = contig_of(ctg_off, nc, g);
    slotinfo_base(draft, g, ctg_off[c], ctg_off[c + 1], soff, slot_info, slot_g);
}

__global__ __launch_bounds__(256) void k_rowcap(ReadsDev R, int64_t n_reads, const uint32_t* __restrict__ ctg_off,
                                                const uint32_t* __restrict__ soff, const int32_t* __restrict__ qs,
                                                const int32_t* __restrict__ qe, const int32_t* __restrict__ span,
                                                uint32_t* __restrict__ rbase, uint32_t* __restrict__ cap_bytes) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    rowcap_record(R, r, ctg_off, soff, qs, qe, span, rbase, cap_bytes);
}

__global__ __launch_bounds__(256) void k_rows(ReadsDev R, int64_t n_reads, const uint32_t* __restrict__ ctg_off,
                                              const uint32_t* __restrict__ soff, const int32_t* __restrict__ qs_in,
                                              const int32_t* __restrict__ qe_in, const uint32_t* __restrict__ rbase,
                                              const uint64_t* __restrict__ rowoff, uint8_t* __restrict__ rows,
                                              uint4* __restrict__ meta, uint32_t* __restrict__ chunk_first,
                                              uint32_t* __restrict__ chunk_last,
                                              unsigned long long* __restrict__ votes) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long myvotes = 0;
    if (r < n_reads) myvotes = rows_record(R, r, ctg_off, soff, qs_in, qe_in, rbase, rowoff, rows, meta, chunk_first, chunk_last);
    for (int o = 32; o > 0; o >>= 1) myvotes += __shfl_down(myvotes, o);
    if ((threadIdx.x & 63) == 0 && myvotes) atomicAdd(votes, myvotes);
}

// Shared tail of k_vote / k_tile: resolve single-state slots, spill DP records, list run heads.
template <int E>
__device__ __forceinline__ void vote_epilogue(const VoteLane<E>& vl, const uint32_t* L, int lane, uint32_t c, bool valid,
                                              uint32_t s, uint32_t info, uint32_t dsym, bool first, uint32_t prev_dsym,
                                              uint32_t basemask, uint32_t S, uint16_t* __restrict__ slot_res,
                                              uint32_t* __restrict__ slot_rec, uint32_t* __restrict__ pool,
                                              uint32_t pool_cap, uint32_t* __restrict__ counters,
                                              uint32_t* __restrict__ heads, uint32_t* __restrict__ redo_out,
                                              uint32_t redo_ci, uint32_t flag_single) {
    (void)S;
    if (__ballot(vl.ovf) != 0ull) {   // more distinct contexts in a slot than this instantiation keeps: redo with a larger E
        if (lane == 0) {
            if (redo_out) redo_out[atomicAdd(&counters[redo_ci], 1u)] = c;
            else atomicOr(&counters[CNT_ERR], ERR_CTX_OVERFLOW);
        }
        return;
    }
    const bool own = lane >= 2 && valid;
    const uint32_t total = vl.total(L, lane);
    const bool single = __popc(basemask) == 1;
    uint32_t psingle = __shfl_up((uint32_t)single, 1);
    const bool prev_is_single = first || psingle != 0;
    const bool is_head = own && !single && prev_is_single;
    const bool need_rec = own && (!single || !prev_is_single);
    if (own) {
        uint32_t res = 0xffu;
        if (single) res = dsym | (((total == 1 ? 1u : 0u) | flag_single) << 8);
        slot_res[s] = (uint16_t)res;
    }
    const uint32_t words = need_rec ? vl.n + REC_FIXED_WORDS : 0u;
    uint32_t incl = words;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const uint32_t wave_total = __shfl(incl, 63);
    uint32_t base = 0;
    if (wave_total) {
        if (lane == 63) base = atomicAdd(&counters[CNT_POOL], wave_total);
        base = __shfl(base, 63);
    }
    const bool fits = (uint64_t)base + wave_total <= (uint64_t)pool_cap;
    if (!fits && lane == 0) atomicOr(&counters[CNT_ERR], ERR_POOL_OVERFLOW);
    uint32_t my_off = 0xffffffffu;
    if (need_rec && fits) {
        my_off = base + incl - words;
        uint32_t hdr = (single ? REC_SINGLE : 0u) | ((info & SI_LAST) ? REC_CTG_LAST : 0u) |
                       (first ? REC_CTG_FIRST : 0u) | (prev_dsym << 4);
        vl.write_record(pool + my_off, s, total, hdr, L, lane);
    }
    if (own) slot_rec[s] = my_off;
    const unsigned long long hb = __ballot(is_head && fits);
    if (hb) {
        uint32_t hbase = 0;
        if (lane == 0) hbase = atomicAdd(&counters[CNT_HEADS], (uint32_t)__popcll(hb));
        hbase = __shfl(hbase, 0);
        if (is_head && fits) heads[hbase + __popcll(hb & ((1ull << lane) - 1ull))] = my_off;
    }
}

// ------------------------------------------------------------------------------------------------
struct SeqLds {   // the record's packed bases, staged in LDS with the rest of its batch
    const uint8_t* b;
    __device__ __forceinline__ uint32_t operator()(int32_t q) const { return (b[q >> 1] >> ((~q & 1) << 2)) & 0xf; }
};

// whole-wave shift right by one lane (lane i receives lane i-1, lane 0 receives 0): one DPP move, no LDS traffic
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}

// ------------------------------------------------------------------------------------------------
// k_desc (fused pipeline, default): one lane per record -> its descriptor (np1_core.h build_desc) and the
// candidate record range of every vote chunk (wave-aggregated min/max instead of per-record atomics)
__global__ __launch_bounds__(256) void k_desc(ReadsDev R, int64_t n_reads, const uint32_t* __restrict__ ctg_off,
                                              const uint32_t* __restrict__ soff, const int32_t* __restrict__ qs,
                                              const int32_t* __restrict__ qe, uint32_t* __restrict__ desc,
                                              uint32_t* __restrict__ ovf_pool, uint32_t ovf_cap,
                                              uint32_t* __restrict__ chunk_first, uint32_t* __restrict__ chunk_last,
                                              uint32_t* __restrict__ counters) {
    // The 96-byte descriptors of a workgroup's 256 records are one contiguous 24 KB stretch of the array: they are built in LDS (the
    // builder writes single words, in no particular order, some of them twice) and leave as full 16-byte lanes, coalesced -- written
    // word by word from the lanes the same bytes cost 2.7 x their size in HBM writes (partial lines evicted between the words).
    __shared__ __attribute__((aligned(16))) uint32_t stage[256 * DESC_WORDS];
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t c0 = 1, c1 = 0;
    uint32_t* mine = stage + threadIdx.x * DESC_WORDS;
    if (r < n_reads) {
#pragma unroll
        for (int k = 0; k < DESC_WORDS; ++k) mine[k] = 0u;      // (words a short descriptor never touches: no stale LDS goes to HBM)
        desc_record_at(mine, R, r, ctg_off, soff, qs, qe, ovf_pool, ovf_cap, counters, &c0, &c1);
    }
    __syncthreads();
    {
        const int64_t r_block = (int64_t)blockIdx.x * blockDim.x;
        const int64_t n_here = n_reads - r_block < 256 ? n_reads - r_block : 256;
        uint4* dst = reinterpret_cast<uint4*>(desc + (uint64_t)r_block * DESC_WORDS);
        const uint4* src = reinterpret_cast<const uint4*>(stage);
        for (int64_t i = threadIdx.x; i < n_here * (DESC_WORDS / 4); i += 256) dst[i] = src[i];
    }
    const bool has = c0 <= c1;
    if (__ballot(has) == 0ull) return;
    uint32_t lo = has ? c0 : 0xffffffffu, hi = has ? c1 : 0u;
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
        if (a < lo) lo = a;
        if (b > hi) hi = b;
    }
    const uint32_t r_wave = (uint32_t)(r - lane);
    if (hi - lo <= 48) {
        for (uint32_t cc = lo; cc <= hi; ++cc) {
            const unsigned long long mk = __ballot(has && c0 <= cc && cc <= c1);
            if (mk && lane == 0) {
                atomicMin(&chunk_first[cc], r_wave + (uint32_t)__builtin_ctzll(mk));
                atomicMax(&chunk_last[cc], r_wave + 63u - (uint32_t)__builtin_clzll(mk));
            }
        }
    } else if (has) {   // records of one wave far apart (sparse coverage, contig boundaries): plain per-record updates
        for (uint32_t cc = c0; cc <= c1; ++cc) {
            atomicMin(&chunk_first[cc], (uint32_t)r);
            atomicMax(&chunk_last[cc], (uint32_t)r);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_tile3 (fused pipeline, default): pileup columns through LDS without symbol rows.
// A workgroup owns NW consecutive vote chunks.  Per batch of candidate records it stages, fully coalesced,
// the records' descriptors (88 B) and packed bases (~75 B) into LDS; then every wave walks the batch in
// record order (wave-uniform loop, descriptor reads are LDS broadcasts) and each lane (= slot) evaluates
// the record's symbol at its own (draft index, insertion column) straight from the staged bases, takes
// its two left neighbours' symbols with DPP shifts and tallies the 3-base context.  HBM traffic: the
// record stream once per overlapped tile (x1.3), slot_info/slot_g, per-slot results, DP records.
// Shared tail of the fused kernels: single-state slots are final; multi-state runs spill DP records.  Pool space
// and run-head slots are claimed ONCE PER WORKGROUP on a counter sharded 8 ways (a single device-scope counter
// bumped by every wave saturates at ~90 atomics/us and used to cost as much as the voting itself).
// Must be reached by every wave of the workgroup (two barriers inside).
template <int E, int NW>
__device__ __forceinline__ void tile_epilogue(const VoteLane<E>& vl, const uint32_t* L, int tid, uint32_t c, bool live, bool redo,
                                              bool valid, uint32_t s, uint32_t info, uint32_t dsym, bool first,
                                              uint32_t prev_dsym, bool single, bool prev_is_single, uint32_t total,
                                              uint16_t* __restrict__ slot_res, uint32_t* __restrict__ slot_rec,
                                              uint32_t* __restrict__ pool, uint32_t pool_cap, uint32_t* __restrict__ counters,
                                              uint32_t* __restrict__ heads, uint32_t heads_cap, uint32_t* __restrict__ redo_out,
                                              uint32_t redo_ci, uint32_t flag_single, uint32_t nvotes_wave,
                                              unsigned long long* __restrict__ votes, bool all_rec = false) {
    __shared__ uint32_t sh_e[3 * NW + 4];
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) sh_e[2 * NW + 2 + wave] = live ? nvotes_wave : 0u;   // vote statistic: one atomic per workgroup, sharded
    if (redo && lane == 0) {   // redo this chunk with a roomier instantiation
        if (redo_out) redo_out[atomicAdd(&counters[redo_ci], 1u)] = c;
        else atomicOr(&counters[CNT_ERR], ERR_CTX_OVERFLOW);
    }
    const bool own = live && lane >= 2 && valid;
    // all_rec (general-rate fp64 path): every slot carries a record and a contig is one run headed by its first slot
    const bool is_head = own && (all_rec ? first : (!single && prev_is_single));
    const bool need_rec = own && (all_rec || !single || !prev_is_single);
    if (own) {
        uint32_t res = 0xffu;
        if (single) res = dsym | (((total == 1 ? 1u : 0u) | flag_single) << 8);
        slot_res[s] = (uint16_t)res;
    }
    const uint32_t words = need_rec ? vl.n + REC_FIXED_WORDS : 0u;
    uint32_t incl = words;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const unsigned long long hb = __ballot(is_head);
    if (lane == 63) sh_e[wave] = incl;
    if (lane == 0) sh_e[NW + wave] = (uint32_t)__popcll(hb);
    __syncthreads();
    if (tid == 0) {
        uint32_t wsum = 0, hsum = 0, vsum = 0;
        for (int w = 0; w < NW; ++w) { wsum += sh_e[w]; hsum += sh_e[NW + w]; vsum += sh_e[2 * NW + 2 + w]; }
        const uint32_t shard = blockIdx.x & (POOL_SHARDS - 1);
        if (vsum) atomicAdd(&votes[shard], (unsigned long long)vsum);
        const uint32_t pregion = pool_cap / POOL_SHARDS, hregion = heads_cap / POOL_SHARDS;
        uint32_t pbase = 0xffffffffu, hbase = 0;
        if (wsum) {
            const uint32_t o = atomicAdd(&counters[CNT_POOL_S0 + shard], wsum);
            if ((uint64_t)o + wsum <= pregion) pbase = shard * pregion + o;
            else atomicOr(&counters[CNT_ERR], ERR_POOL_OVERFLOW);
        }
        if (hsum) {
            const uint32_t o = atomicAdd(&counters[CNT_HEADS_S0 + shard], hsum);
            if ((uint64_t)o + hsum <= hregion) hbase = shard * hregion + o;
            else { atomicOr(&counters[CNT_ERR], ERR_POOL_OVERFLOW); pbase = 0xffffffffu; }
        }
        sh_e[2 * NW] = pbase;
        sh_e[2 * NW + 1] = hbase;
    }
    __syncthreads();
    uint32_t pbase = sh_e[2 * NW], hbase = sh_e[2 * NW + 1];
    const bool fits = pbase != 0xffffffffu;
    for (int w = 0; w < wave; ++w) { pbase += sh_e[w]; hbase += sh_e[NW + w]; }
    uint32_t my_off = 0xffffffffu;
    if (need_rec && fits) {
        my_off = pbase + incl - words;
        const uint32_t hdr = (single ? REC_SINGLE : 0u) | ((info & SI_LAST) ? REC_CTG_LAST : 0u) |
                             (first ? REC_CTG_FIRST : 0u) | (prev_dsym << 4);
        vl.write_record(pool + my_off, s, total, hdr, L, lane);
    }
    if (own) slot_rec[s] = my_off;
    if (is_head && fits) heads[hbase + __popcll(hb & ((1ull << lane) - 1ull))] = my_off;
}

// one part of one record against this wave's 64 slots (d may live in LDS or, for overflow parts, in HBM: the two
// call sites keep the address spaces apart so the common path compiles to ds_read)
template <int E>
__device__ __forceinline__ void vote_part(const uint32_t* d, const SeqLds& sq, bool valid, uint32_t s, uint32_t g,
                                          int32_t jj, int lane, uint32_t& rsym, uint32_t& basemask, VoteLane<E>& vl,
                                          uint32_t* L, uint32_t ablate = 0) {
    const bool cov = valid && s >= d[0] && s <= d[1];
    if (cov) rsym = (ablate & 8u) ? (g & 0xfu) : desc_symbol(d, g, jj, sq);
    const uint32_t p1 = wave_shr1(rsym), p2 = wave_shr1(p1);
    if (cov) {
        basemask |= 1u << rsym;
        if (lane >= 2 && !(ablate & 4u)) vl.tally(p2 << 8 | p1 << 4 | rsym, L, lane);
    }
}

template <int E, int NW>
__global__ __launch_bounds__(NW * 64) void k_tile3(ReadsDev R, const uint32_t* __restrict__ soff,
                                                   const uint32_t* __restrict__ desc,
                                                   const uint32_t* __restrict__ ovf_pool,
                                                   const uint32_t* __restrict__ chunk_first,
                                                   const uint32_t* __restrict__ chunk_last, uint32_t n_chunks,
                                                   const uint32_t* __restrict__ redo_in, uint32_t n_items,
                                                   const uint8_t* __restrict__ slot_info,
                                                   const uint32_t* __restrict__ slot_g, uint32_t S, uint32_t seq_w,
                                                   uint32_t nb_max, uint16_t* __restrict__ slot_res,
                                                   uint32_t* __restrict__ slot_rec, uint32_t* __restrict__ pool,
                                                   uint32_t pool_cap, uint32_t* __restrict__ counters,
                                                   uint32_t* __restrict__ heads, uint32_t heads_cap,
                                                   uint32_t* __restrict__ redo_out,
                                                   uint32_t redo_ci, uint32_t flag_single,
                                                   unsigned long long* __restrict__ votes, uint32_t ablate) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ __attribute__((aligned(16))) uint32_t sh_r[4];
    uint32_t* lists = lds;                             // NW * (E-2) * 64
    uint32_t* dsc = lists + NW * (E - 2) * 64;         // nb_max * DESC_WORDS
    uint32_t* seqst = dsc + (nb_max + 1) * DESC_WORDS; // nb_max * seq_w + 8 (one spare descriptor slot before it)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t item = blockIdx.x;
    if (item >= n_items) return;
    const uint32_t cbase = redo_in ? redo_in[item] : item * NW;
    const uint32_t c = cbase + wave;
    const bool chunk_ok = c < n_chunks;
    if (tid == 0) {
        uint32_t r0 = 0xffffffffu, r1 = 0;
        for (int w = 0; w < NW; ++w) {
            uint32_t cc = cbase + w;
            if (cc < n_chunks) {
                uint32_t f = chunk_first[cc];
                if (f != 0xffffffffu) {
                    if (f < r0) r0 = f;
                    uint32_t l = chunk_last[cc];
                    if (l > r1) r1 = l;
                }
            }
        }
        sh_r[0] = r0;
        sh_r[1] = r1;
    }
    uint32_t* L = lists + wave * (E - 2) * 64;
    const int64_t s64 = (int64_t)c * VOTE_CH - 2 + lane;
    const bool valid = chunk_ok && s64 >= 0 && s64 < (int64_t)S;
    const uint32_t s = (uint32_t)s64;
    const uint32_t info = valid ? slot_info[s] : 0u;
    const uint32_t g = valid ? slot_g[s] : 0u;
    const int32_t jj = (valid && (info & SI_INSERT)) ? (int32_t)(s - soff[g]) - 1 : -1;   // insertion column, -1 = base slot
    const uint32_t dsym = info & 0xf;
    const bool first = (info & SI_FIRST) != 0;
    // the draft votes once per slot with its own rolling context (contig.c:373-383)
    uint32_t d1 = wave_shr1(dsym), d2 = wave_shr1(d1);
    const uint32_t f1 = wave_shr1((uint32_t)first);
    const uint32_t prev_dsym = d1;
    if (first) { d1 = 0; d2 = 0; }
    else if (f1) d2 = 0;
    VoteLane<E> vl;
    vl.init(d2 << 8 | d1 << 4 | dsym);
    uint32_t basemask = 1u << dsym;
    const uint32_t sv = valid ? s : 0xffffffffu;   // slot for coverage tests (never covered when invalid)
    const int64_t cs = (int64_t)c * VOTE_CH - 2, ce = (int64_t)c * VOTE_CH + VOTE_CH - 1;
    __syncthreads();
    const uint32_t r0 = sh_r[0], r1 = sh_r[1];
    if (r0 != 0xffffffffu) {
        for (uint64_t rb = r0; rb <= r1; rb += nb_max) {
            const uint32_t nb = (uint32_t)((r1 - rb + 1 < nb_max) ? (r1 - rb + 1) : nb_max);
            // ---- stage descriptors and packed bases of the batch (both contiguous in HBM): 16-byte lanes, coalesced
            {
                const uint4* dsrc = reinterpret_cast<const uint4*>(desc + rb * DESC_WORDS);
                uint4* ddst = reinterpret_cast<uint4*>(dsc);
                for (uint32_t i = tid; i < nb * (DESC_WORDS / 4); i += NW * 64) ddst[i] = dsrc[i];
            }
            const uint64_t sq0 = R.seq_off[rb] & ~15ull;
            const uint64_t sq1 = R.seq_off[rb + nb - 1] + (((uint64_t)R.l_qseq[rb + nb - 1] + 1) >> 1);
            const uint32_t sq_quads = (uint32_t)((sq1 - sq0 + 15) >> 4);
            // the pool holds the records back to back, so nb records never need more than nb * seq_w words (+ alignment)
            const uint32_t sq_fit = sq_quads * 4 <= nb_max * seq_w + 8 ? sq_quads : (nb_max * seq_w + 8) / 4;
            if (sq_fit != sq_quads && tid == 0) atomicOr(&counters[CNT_ERR], ERR_BAD_RECORD);
            {
                const uint4* src = reinterpret_cast<const uint4*>(R.seq + sq0);
                uint4* dst = reinterpret_cast<uint4*>(seqst);
                for (uint32_t i = tid; i < sq_fit; i += NW * 64) dst[i] = src[i];
            }
            const uint32_t sq0_lo = (uint32_t)sq0;   // descriptors carry the low word of their record's pool offset
            __syncthreads();
            // ---- this wave's chunk votes over the batch, in record order
            if (chunk_ok && !(ablate & 2u)) {
                uint32_t a = nb, b = 0;
                for (uint32_t base = 0; base < nb; base += 64) {
                    const uint32_t i = base + lane;
                    bool hit = false;
                    if (i < nb) {
                        const uint32_t sf = dsc[i * DESC_WORDS], sl = dsc[i * DESC_WORDS + DESC_NEXT + 1];   // whole record
                        hit = sf <= sl && dsc[i * DESC_WORDS + 1] >= sf && (int64_t)sl + 2 >= cs && (int64_t)sf <= ce;
                    }
                    const unsigned long long mk = __ballot(hit);
                    if (mk) {
                        const uint32_t lo = base + (uint32_t)__builtin_ctzll(mk), hi = base + 63u - (uint32_t)__builtin_clzll(mk);
                        if (lo < a) a = lo;
                        if (hi > b) b = hi;
                    }
                }
                if (a < nb && !(ablate & 16u)) {
                    const uint8_t* seqb = reinterpret_cast<const uint8_t*>(seqst);
                    // descriptor head (sfirst, slast, counts, base offset) and first segment of record a; the loop
                    // fetches record i+1's while it votes record i (one spare descriptor slot keeps the loads in range)
                    uint4 h = *reinterpret_cast<const uint4*>(dsc + a * DESC_WORDS);
                    uint2 sg = *reinterpret_cast<const uint2*>(dsc + a * DESC_WORDS + DESC_SEG0);
                    for (uint32_t i = a; i <= b; ++i) {
                        const uint4 hn = *reinterpret_cast<const uint4*>(dsc + (i + 1) * DESC_WORDS);
                        const uint2 sgn = *reinterpret_cast<const uint2*>(dsc + (i + 1) * DESC_WORDS + DESC_SEG0);
                        if (!(h.z & DESC_CHAIN) && !(ablate & 32u)) {
                            // wave-uniform shape test; the body is branch-free apart from uniform trip counts
                            const uint32_t nseg = h.z & 0xffu, nins = (h.z >> 8) & 0xffu;
                            const bool cov = sv >= h.x && sv <= h.y;
                            uint32_t q = 0;
                            bool isdel = true;   // covered insertion column the record merely passes (or pads): DEL
                            {
                                const uint32_t off = g - sg.x;
                                const bool in = jj < 0 && off < (sg.y & 0xffffu);
                                isdel = in ? (sg.y >> 16) == 0xffffu : isdel;
                                q = in ? (sg.y >> 16) + off : q;
                            }
                            for (uint32_t k2 = 1; k2 < nseg; ++k2) {
                                const uint2 sk = *reinterpret_cast<const uint2*>(dsc + i * DESC_WORDS + DESC_SEG0 + 2 * k2);
                                const uint32_t off = g - sk.x;
                                const bool in = jj < 0 && off < (sk.y & 0xffffu);
                                isdel = in ? (sk.y >> 16) == 0xffffu : isdel;
                                q = in ? (sk.y >> 16) + off : q;
                            }
                            for (uint32_t k2 = 0; k2 < nins; ++k2) {
                                const uint2 ik = *reinterpret_cast<const uint2*>(dsc + i * DESC_WORDS + DESC_INS0 + 2 * k2);
                                const bool in = jj >= 0 && ik.x == g && (uint32_t)jj < (ik.y & 0xffffu);
                                isdel = in ? false : isdel;
                                q = in ? (ik.y >> 16) + (uint32_t)jj : q;
                            }
                            q = (cov && !isdel) ? q : 0u;
                            const uint32_t byte = seqb[(h.w - sq0_lo) + (q >> 1)];
                            uint32_t sym = (byte >> ((~q & 1u) << 2)) & 0xfu;
                            sym = cov ? (isdel ? 3u : sym) : 0u;
                            const uint32_t p1 = wave_shr1(sym), p2 = wave_shr1(p1);
                            const uint32_t k = p2 << 8 | p1 << 4 | sym;
                            basemask |= cov ? 1u << sym : 0u;
                            const bool vote = cov && lane >= 2;
                            const bool m0 = vote && k == vl.k0, m1 = vote && k == vl.k1;
                            vl.c0 += m0 ? 1u : 0u;
                            vl.c1 += m1 ? 1u : 0u;
                            const bool rest = vote && !m0 && !m1;
                            if (__ballot(rest) != 0ull) {
                                if (rest) vl.tally(k, L, lane);   // a context seen for the first time, or one kept in the LDS list
                            }
                        } else {
                            const uint32_t* d = dsc + i * DESC_WORDS;   // LDS: every access below is a ds_read broadcast
                            const SeqLds sq{seqb + (h.w - sq0_lo)};
                            uint32_t rsym = 0;   // this record's symbol at my slot (kept across the parts of a chained record)
                            vote_part<E>(d, sq, valid, s, g, jj, lane, rsym, basemask, vl, L);
                            uint32_t nx = d[DESC_NEXT];
                            while (nx) {   // rare: record with more indel operations than one descriptor holds; parts live in HBM
                                const uint32_t* dg = ovf_pool + (uint64_t)(nx - 1) * DESC_WORDS;
                                vote_part<E>(dg, sq, valid, s, g, jj, lane, rsym, basemask, vl, L);
                                nx = dg[DESC_NEXT];
                            }
                        }
                        h = hn;
                        sg = sgn;
                    }
                }
            }
            __syncthreads();
        }
    }
    // the vote statistic: every vote went into exactly one counter of its lane, and the draft's own context opened c0 with 1 (round 6: it
    // used to be counted vote by vote inside the record loop)
    uint32_t tsum = vl.c0 + vl.c1;
    for (uint32_t e = 2; e < vl.n; ++e) tsum += L[(e - 2) * 64 + lane] & 0xffffu;
    uint32_t nvotes = tsum - 1u;
    for (int o = 32; o > 0; o >>= 1) nvotes += __shfl_down(nvotes, o);
    const bool ovf_any = chunk_ok && __ballot(vl.ovf) != 0ull;
    const bool single = __popc(basemask) == 1;
    const uint32_t psingle = wave_shr1((uint32_t)single);   // every lane must execute the DPP move: keep it out of the || below
    const bool prev_is_single = first || psingle != 0;
    // flag_single bit 8 (FLAG_ALL_RECORDS): general-rate path, every slot spills a record (np1_core.h:dp_run<true>)
    tile_epilogue<E, NW>(vl, L, tid, c, chunk_ok && !ovf_any, ovf_any, valid, s, info, dsym, first, prev_dsym, single, prev_is_single,
                         tsum & 0xffffu, slot_res, slot_rec, pool, pool_cap, counters, heads, heads_cap, redo_out, redo_ci,
                         flag_single & 0xffu, nvotes, votes, (flag_single & FLAG_ALL_RECORDS) != 0);
}


// ------------------------------------------------------------------------------------------------
// k_tile9 (default since round 4; np1_tile9.h has the idea and the per-lane logic, shared with the host model).
// A workgroup of NW waves stages the descriptors and packed bases of its candidate records like k_tile3 does; every wave owns
// T9_CH vote chunks (248 slots), four slots per lane.
//   record loop   wave-uniform over the staged records, the record's fields fetched one record ahead and moved to scalar registers;
//                 per lane two range checks (does the record cover my window / touch my slots), one range check per segment,
//                 one 32-bit LDS read of six bases, one compare: agreement -> a register counter; any other touching pair -> an
//                 entry (record, lane) on the wave's list in LDS, appended densely per step and chained per lane
//   evaluation    after the loop of a staging round: 64 list entries at a time, whoever owns them -- the owner's window from the
//                 slot arrays, the record's segment table into registers, the symbol at each covered position (t9_code)
//   tally         after the last round: lanes with a few entries walk their chains; lanes with many (a draft error under the
//                 window: every covering record is an entry) are taken by the whole wave, distinct contexts and counts by ballot
//   epilogue      as tile_epilogue, four slots per lane
constexpr uint32_t T9_NIL = 0xffffu;
template <int E>
__host__ __device__ constexpr uint32_t t9_wave_words() { return 4u * (E - 2) * 64u + 2u * T9_DL; }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

template <int E, int NW>
__global__ __launch_bounds__(NW * 64) void k_tile9(ReadsDev R, const uint32_t* __restrict__ soff, const uint32_t* __restrict__ desc,
                                                   const uint32_t* __restrict__ ovf_pool, const uint32_t* __restrict__ chunk_first,
                                                   const uint32_t* __restrict__ chunk_last, uint32_t n_chunks, const uint8_t* __restrict__ slot_info,
                                                   const uint32_t* __restrict__ slot_g, uint32_t S, uint32_t seq_w, uint32_t nb_max,
                                                   uint16_t* __restrict__ slot_res, uint32_t* __restrict__ slot_rec, uint32_t* __restrict__ pool,
                                                   uint32_t pool_cap, uint32_t* __restrict__ counters, uint32_t* __restrict__ heads, uint32_t heads_cap,
                                                   uint32_t* __restrict__ redo_out, uint32_t redo_ci, uint32_t flag_single,
                                                   unsigned long long* __restrict__ votes, unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t sh_r[4];
    __shared__ uint32_t sh_e[3 * NW + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // NP1_T9_PHASES=1: shader-clock cycles per wave and phase (0 setup, 1 staging + barriers, 2 record loop, 3 evaluation, 4 tally,
    // 5 epilogue; 6 waves, 7 record-loop steps, 8 deferred entries, 9 staging rounds, 10 lanes tallied by the whole wave), kept per wave
    // in registers and added up once at the end on 64 shards (one counter bumped by every wave costs more than the phases)
    long long tph = dbg ? clock64() : 0;
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};
    auto phase = [&](int k) {
        if (dbg) {
            const long long now = clock64();
            pc[k] += (unsigned long long)(now - tph);
            tph = now;
        }
    };
    constexpr uint32_t PER_WAVE = t9_wave_words<E>();
    uint32_t* L = lds + (uint32_t)wave * PER_WAVE;                            // context lists: slot j's at L + j * (E - 2) * 64
    uint2* dl = reinterpret_cast<uint2*>(L + 4 * (E - 2) * 64);               // deferred entries: x = record index, later the evaluated code; y = next of the same lane | owner lane << 16
    uint32_t* dsc = lds + (uint32_t)NW * PER_WAVE;                            // (nb_max + 1) descriptors
    uint32_t* seqst = dsc + (nb_max + 1) * DESC_WORDS;                        // nb_max * seq_w + 8 words of packed bases
    const uint32_t t = blockIdx.x * NW + wave;                                // this wave's tile
    const uint32_t c0 = t * T9_CH, tile_s0 = t * T9_SLOTS;
    const bool wave_ok = c0 < n_chunks;
    if (tid == 0) {
        uint32_t r0 = 0xffffffffu, r1 = 0;
        for (uint32_t k = 0; k < NW * T9_CH; ++k) {
            const uint32_t cc = blockIdx.x * NW * T9_CH + k;
            if (cc < n_chunks) {
                const uint32_t f = chunk_first[cc];
                if (f != 0xffffffffu) {
                    if (f < r0) r0 = f;
                    const uint32_t l = chunk_last[cc];
                    if (l > r1) r1 = l;
                }
            }
        }
        sh_r[0] = r0;
        sh_r[1] = r1;
    }
    // ---- the lane's window: its own four slots with two vector loads, the two slots in front of them from the lane below (lane 0: two
    // more loads); only windows at the ends of the batch take the slot-by-slot path
    T9Win w;
    uint32_t info[6];
    {
        const int64_t s0l = (int64_t)tile_s0 + 4 * ((int64_t)lane - 1);
        const bool own_all = wave_ok && lane <= 62 && s0l >= 0 && s0l + 4 <= (int64_t)S;
        uint32_t iw = 0;
        uint4 gw = make_uint4(0, 0, 0, 0);
        if (own_all) {
            iw = *reinterpret_cast<const uint32_t*>(slot_info + s0l);
            gw = *reinterpret_cast<const uint4*>(slot_g + s0l);
        }
        uint32_t ih = wave_shr1(iw) >> 16, gh0 = wave_shr1(gw.z), gh1 = wave_shr1(gw.w);
        const bool below_all = wave_shr1((uint32_t)own_all) != 0;
        const bool halo_direct = lane == 0 && own_all && s0l >= 2;
        if (halo_direct) {
            ih = *reinterpret_cast<const uint16_t*>(slot_info + s0l - 2);
            const uint2 gg = *reinterpret_cast<const uint2*>(slot_g + s0l - 2);
            gh0 = gg.x; gh1 = gg.y;
        }
        const bool fast = own_all && (lane == 0 ? halo_direct : below_all);
        if (fast) {
            uint32_t g6[6] = {gh0, gh1, gw.x, gw.y, gw.z, gw.w};
            info[0] = ih & 0xffu; info[1] = (ih >> 8) & 0xffu;
            info[2] = iw & 0xffu; info[3] = (iw >> 8) & 0xffu; info[4] = (iw >> 16) & 0xffu; info[5] = iw >> 24;
            t9_window_from((uint32_t)s0l, true, 63u, info, g6, &w);
        } else if (wave_ok && lane <= 62) {
            t9_window(tile_s0, lane, S, slot_info, slot_g, &w, info);
        } else {
            uint32_t g6[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < 6; ++p) info[p] = 0;
            t9_window_from((uint32_t)s0l, false, 0u, info, g6, &w);
        }
    }
    VoteLane<E> vl[4];
    uint32_t basemask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = j + 2;
        uint32_t d0 = info[p] & 0xfu, d1 = info[p - 1] & 0xfu, d2 = info[p - 2] & 0xfu;
        if (info[p] & SI_FIRST) { d1 = 0; d2 = 0; }
        else if (info[p - 1] & SI_FIRST) d2 = 0;
        vl[j].init(d2 << 8 | d1 << 4 | d0);
        basemask[j] = 1u << d0;
    }
    const bool act = w.active, act_plain = w.active && w.plain;
    const uint32_t s0 = w.s0, s0p3 = w.s0 + 3u, ga = w.g[0], D = w.D;
    uint32_t c_all = 0, nvotes = 0, my_n = 0;
    uint32_t head = T9_NIL, prev = T9_NIL;      // this lane's chain through the deferred list
    uint32_t dl_n = 0;                          // entries appended so far (wave-uniform; beyond T9_DL: overflow, nothing is stored)
    const int64_t cs = (int64_t)tile_s0 - 4, ce = (int64_t)tile_s0 + T9_SLOTS - 1;
    __syncthreads();
    phase(0);
    uint32_t n_steps = 0, n_rounds = 0, n_hot = 0;
    const uint32_t r0 = sh_r[0], r1 = sh_r[1];
    if (r0 != 0xffffffffu) {
        for (uint64_t rb = r0; rb <= r1; rb += nb_max) {
            ++n_rounds;
            const uint32_t nb = (uint32_t)((r1 - rb + 1 < nb_max) ? (r1 - rb + 1) : nb_max);
            {
                const uint4* dsrc = reinterpret_cast<const uint4*>(desc + rb * DESC_WORDS);
                uint4* ddst = reinterpret_cast<uint4*>(dsc);
                for (uint32_t i = tid; i < nb * (DESC_WORDS / 4); i += NW * 64) ddst[i] = dsrc[i];
            }
            const uint64_t sq0 = R.seq_off[rb] & ~15ull;
            const uint64_t sq1 = R.seq_off[rb + nb - 1] + (((uint64_t)R.l_qseq[rb + nb - 1] + 1) >> 1);
            const uint32_t sq_quads = (uint32_t)((sq1 - sq0 + 15) >> 4);
            const uint32_t sq_fit = sq_quads * 4 <= nb_max * seq_w + 4 ? sq_quads : (nb_max * seq_w + 4) / 4;    // (four words stay free behind: t9_fetch8 reads a word from any base)
            if (sq_fit != sq_quads && tid == 0) atomicOr(&counters[CNT_ERR], ERR_BAD_RECORD);
            {
                const uint4* src = reinterpret_cast<const uint4*>(R.seq + sq0);
                uint4* dst = reinterpret_cast<uint4*>(seqst);
                for (uint32_t i = tid; i < sq_fit; i += NW * 64) dst[i] = src[i];
            }
            const uint32_t sq0_lo = (uint32_t)sq0;
            __syncthreads();
            phase(1);
            const uint8_t* seqb = reinterpret_cast<const uint8_t*>(seqst);
            if (wave_ok) {
                uint32_t a = nb, b = 0;
                for (uint32_t base = 0; base < nb; base += 64) {
                    const uint32_t i = base + lane;
                    bool hit = false;
                    if (i < nb) {
                        const uint32_t sf = dsc[i * DESC_WORDS], sl = dsc[i * DESC_WORDS + DESC_NEXT + 1];
                        hit = sf <= sl && (int64_t)sl >= cs && (int64_t)sf <= ce;
                    }
                    const unsigned long long mk = __ballot(hit);
                    if (mk) {
                        const uint32_t lo = base + (uint32_t)__builtin_ctzll(mk), hi = base + 63u - (uint32_t)__builtin_clzll(mk);
                        if (lo < a) a = lo;
                        if (hi > b) b = hi;
                    }
                }
                const uint32_t round_start = dl_n < T9_DL ? dl_n : T9_DL;
                if (a < nb) {
                    n_steps += b - a + 1;
                    // d[0..3] (run start, -, flag word, base offset), the first three segments d[4..9] and the end of the whole run d[23] are
                    // fetched one record ahead (one spare descriptor slot keeps the loads in range)
                    const uint32_t* dp = dsc + a * DESC_WORDS;
                    uint4 h = *reinterpret_cast<const uint4*>(dp);
                    uint4 s01 = *reinterpret_cast<const uint4*>(dp + DESC_SEG0);
                    uint2 s2 = *reinterpret_cast<const uint2*>(dp + DESC_SEG0 + 4);
                    uint32_t slw = dp[DESC_NEXT + 1];
                    for (uint32_t i = a; i <= b; ++i) {
                        const uint32_t* dn = dsc + (i + 1) * DESC_WORDS;
                        const uint4 hn = *reinterpret_cast<const uint4*>(dn);
                        const uint4 s01n = *reinterpret_cast<const uint4*>(dn + DESC_SEG0);
                        const uint2 s2n = *reinterpret_cast<const uint2*>(dn + DESC_SEG0 + 4);
                        const uint32_t slwn = dn[DESC_NEXT + 1];
                        const uint32_t sf = rfl(h.x), sl = rfl(slw), cnt = rfl(h.z);
                        if (sf <= sl) {      // (a record that votes on nothing: filtered, or trimmed away)
                            // coverage as two range checks: the window is covered iff s0 in [sf + 2, sl - 3]; the own slots are touched iff
                            // s0 + 3 in [sf, sl + 3]
                            const uint32_t a2 = sf + 2u;
                            const bool can_full = sl >= sf + 5u;
                            const bool cfull = can_full && (s0 - a2) <= (sl - 3u - a2);
                            const bool crel = (s0p3 - sf) <= (sl + 3u - sf);
                            // a matched segment holds the window iff (ga - g_lo) < len - 5; the first such segment gives the query index
                            const uint32_t g0 = rfl(s01.x), w0 = rfl(s01.y), g1 = rfl(s01.z), w1 = rfl(s01.w), g2 = rfl(s2.x), w2 = rfl(s2.y);
                            const uint32_t nseg = cnt & 0xffu;
                            const uint32_t l0 = nseg >= 1u ? t9_seg_lim(w0) : 0u, l1 = nseg >= 2u ? t9_seg_lim(w1) : 0u, l2 = nseg >= 3u ? t9_seg_lim(w2) : 0u;
                            const uint32_t o0 = ga - g0, o1 = ga - g1, o2 = ga - g2;
                            const bool in0 = o0 < l0, in1 = o1 < l1, in2 = o2 < l2;
                            uint32_t q = in2 ? (w2 >> 16) + o2 : 0u;
                            q = in1 ? (w1 >> 16) + o1 : q;
                            q = in0 ? (w0 >> 16) + o0 : q;
                            const uint32_t F = t9_fetch8(seqb + (rfl(h.w) - sq0_lo), q);
                            const bool agree = act_plain && cfull && (in0 || in1 || in2) && !(cnt & DESC_CHAIN) && (F >> 8) == D;
                            c_all += agree ? 1u : 0u;
                            const bool isent = act && crel && !agree;
                            const unsigned long long m = __ballot(isent);
                            if (m) {
                                const uint32_t idx = dl_n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                                if (isent) {
                                    ++my_n;
                                    if (idx < T9_DL) {
                                        dl[idx] = make_uint2(i, T9_NIL | (uint32_t)lane << 16);
                                        if (prev != T9_NIL) *reinterpret_cast<uint16_t*>(&dl[prev].y) = (uint16_t)idx; else head = idx;
                                        prev = idx;
                                    }
                                }
                                dl_n += (uint32_t)__popcll(m);
                            }
                        }
                        h = hn; s01 = s01n; s2 = s2n; slw = slwn;
                    }
                }
                // ---- this round's entries: (record, lane) -> the record's symbols at the covered positions of the lane's window.
                // Dense over the list; the owner's window comes from the slot arrays (L2), the record from LDS (gone with the round).
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                phase(2);
                const uint32_t round_end = dl_n < T9_DL ? dl_n : T9_DL;
                for (uint32_t x = round_start + lane; x < round_end; x += 64) {
                    const uint2 e = dl[x];
                    const uint32_t owner = e.y >> 16, i = e.x;
                    const int64_t so = (int64_t)tile_s0 + 4 * ((int64_t)owner - 1);
                    uint32_t g6[6], inf6[6];
                    if (so >= 2 && so + 4 <= (int64_t)S) {
                        const uint2 ga2 = *reinterpret_cast<const uint2*>(slot_g + so - 2);
                        const uint4 gb4 = *reinterpret_cast<const uint4*>(slot_g + so);
                        const uint32_t ia = *reinterpret_cast<const uint16_t*>(slot_info + so - 2), ib = *reinterpret_cast<const uint32_t*>(slot_info + so);
                        g6[0] = ga2.x; g6[1] = ga2.y; g6[2] = gb4.x; g6[3] = gb4.y; g6[4] = gb4.z; g6[5] = gb4.w;
                        inf6[0] = ia & 0xffu; inf6[1] = ia >> 8; inf6[2] = ib & 0xffu; inf6[3] = (ib >> 8) & 0xffu; inf6[4] = (ib >> 16) & 0xffu; inf6[5] = ib >> 24;
                    } else {
#pragma unroll
                        for (int p = 0; p < 6; ++p) {
                            const int64_t sp = so - 2 + p;
                            const bool v = sp >= 0 && sp < (int64_t)S;
                            g6[p] = v ? slot_g[sp] : 0u;
                            inf6[p] = v ? slot_info[sp] : 0u;
                        }
                    }
                    int32_t jj6[6];
#pragma unroll
                    for (int p = 0; p < 6; ++p) {
                        jj6[p] = -1;
                        if (inf6[p] & SI_INSERT) jj6[p] = (int32_t)((uint32_t)(so - 2 + p) - soff[g6[p]]) - 1;     // (rare: an insertion column in the window)
                    }
                    const uint32_t* d = dsc + i * DESC_WORDS;
                    dl[x].x = t9_code(d, ovf_pool, seqb + (d[3] - sq0_lo), (uint32_t)so, g6, jj6);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                phase(3);
            }
            __syncthreads();
        }
    }
    phase(1);
    // ---- tally, in record order per slot
    const bool dl_ovf = dl_n > T9_DL;
    const bool hot = my_n > T9_HOT;
    if (!dl_ovf) {
        uint32_t x = hot ? T9_NIL : head;          // a lane with a few entries walks its own chain
        while (x != T9_NIL) {
            const uint2 e = dl[x];
            nvotes += t9_tally<E>(e.x, vl, basemask, L, lane);
            x = e.y & 0xffffu;
        }
        unsigned long long hm = __ballot(hot && act);
        while (hm) {                               // a lane with many: the whole wave, 64 list entries at a time
            const int hl = __builtin_ctzll(hm);
            hm &= hm - 1ull;
            ++n_hot;
            for (uint32_t base = 0; base < dl_n; base += 64) {
                const uint32_t x2 = base + lane;
                uint2 e = make_uint2(0u, 0xffffffffu);
                if (x2 < dl_n) e = dl[x2];
                const bool sel = (e.y >> 16) == (uint32_t)hl;
                if (__ballot(sel) == 0ull) continue;
                const uint32_t lo = (e.x >> 24) & 7u, hi = (e.x >> 27) & 7u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t p = (uint32_t)j + 2u;
                    const bool on = sel && p >= lo && p <= hi;
                    const uint32_t k = (e.x >> (20 - 4 * p)) & 0xfffu;
                    unsigned long long rem = __ballot(on);
                    while (rem) {                  // distinct contexts of this slot in list order = first-seen order, with their counts
                        const int ld = __builtin_ctzll(rem);
                        const uint32_t kL = (uint32_t)__builtin_amdgcn_readlane((int)k, ld);
                        const unsigned long long mm = __ballot(on && k == kL);
                        rem &= ~mm;
                        const uint32_t cnt = (uint32_t)__popcll(mm);
                        t9_tally_ctx<E>(kL, cnt, lane == hl, vl[j], basemask[j], L + j * (E - 2) * 64, lane);
                        nvotes += lane == hl ? cnt : 0u;
                    }
                }
            }
        }
    }
    phase(4);
    bool lane_ovf = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) { vl[j].c0 += c_all; lane_ovf = lane_ovf || vl[j].ovf; }
    nvotes += 4u * c_all;
    if (lane == 0 || lane == 63) nvotes = 0;      // (lane 0 carries the four slots in front of the tile)
    for (int o = 32; o > 0; o >>= 1) nvotes += __shfl_down(nvotes, o);
    const bool redo = wave_ok && (dl_ovf || __ballot(lane_ovf) != 0ull);
    const bool live = wave_ok && !redo;
    // ---- epilogue: single-state slots are final, multi-state runs spill DP records (tile_epilogue, four slots per lane)
    if (redo && lane == 0) {
        const uint32_t nch = n_chunks - c0 < T9_CH ? n_chunks - c0 : T9_CH;
        if (redo_out) {
            const uint32_t at = atomicAdd(&counters[redo_ci], nch);
            for (uint32_t k = 0; k < nch; ++k) redo_out[at + k] = c0 + k;
        } else atomicOr(&counters[CNT_ERR], ERR_CTX_OVERFLOW);
    }
    bool single[4], is_head[4], need_rec[4];
    uint32_t total[4], words[4], res4[4], off4[4];
    const bool own4 = live && lane >= 1 && lane <= 62 && w.active && (uint64_t)w.s0 + 4 <= (uint64_t)S;     // all four slots: one store each for slot_res / slot_rec
    uint32_t lane_words = 0, lane_heads = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) single[j] = __popc(basemask[j]) == 1;
    const uint32_t psingle3 = wave_shr1((uint32_t)single[3]);      // every lane executes the DPP move
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = w.s0 + (uint32_t)j;
        const bool own = live && lane >= 1 && lane <= 62 && w.active && s < S;
        const bool first = (info[j + 2] & SI_FIRST) != 0;
        const bool prev_is_single = first || (j > 0 ? single[j > 0 ? j - 1 : 0] : psingle3 != 0);
        is_head[j] = own && !single[j] && prev_is_single;
        need_rec[j] = own && (!single[j] || !prev_is_single);
        total[j] = vl[j].total(L + j * (E - 2) * 64, lane);
        res4[j] = 0xffu;
        if (single[j]) res4[j] = (info[j + 2] & 0xfu) | (((total[j] == 1 ? 1u : 0u) | flag_single) << 8);
        if (own && !own4) slot_res[s] = (uint16_t)res4[j];
        words[j] = need_rec[j] ? vl[j].n + REC_FIXED_WORDS : 0u;
        lane_words += words[j];
        lane_heads += is_head[j] ? 1u : 0u;
    }
    if (own4) *reinterpret_cast<uint2*>(slot_res + w.s0) = make_uint2(res4[0] | res4[1] << 16, res4[2] | res4[3] << 16);
    uint32_t incl = lane_heads << 16 | lane_words;       // (a wave's records are a few thousand words at most)
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) { sh_e[wave] = incl & 0xffffu; sh_e[NW + wave] = incl >> 16; }
    if (lane == 0) sh_e[2 * NW + 2 + wave] = live ? nvotes : 0u;
    __syncthreads();
    if (tid == 0) {
        uint32_t wsum = 0, hsum = 0, vsum = 0;
        for (int k = 0; k < NW; ++k) { wsum += sh_e[k]; hsum += sh_e[NW + k]; vsum += sh_e[2 * NW + 2 + k]; }
        const uint32_t shard = blockIdx.x & (POOL_SHARDS - 1);
        if (vsum) atomicAdd(&votes[shard], (unsigned long long)vsum);
        const uint32_t pregion = pool_cap / POOL_SHARDS, hregion = heads_cap / POOL_SHARDS;
        uint32_t pbase = 0xffffffffu, hbase = 0;
        if (wsum) {
            const uint32_t o = atomicAdd(&counters[CNT_POOL_S0 + shard], wsum);
            if ((uint64_t)o + wsum <= pregion) pbase = shard * pregion + o;
            else atomicOr(&counters[CNT_ERR], ERR_POOL_OVERFLOW);
        }
        if (hsum) {
            const uint32_t o = atomicAdd(&counters[CNT_HEADS_S0 + shard], hsum);
            if ((uint64_t)o + hsum <= hregion) hbase = shard * hregion + o;
            else { atomicOr(&counters[CNT_ERR], ERR_POOL_OVERFLOW); pbase = 0xffffffffu; }
        }
        sh_e[2 * NW] = pbase;
        sh_e[2 * NW + 1] = hbase;
    }
    __syncthreads();
    uint32_t pbase = sh_e[2 * NW], hbase = sh_e[2 * NW + 1];
    const bool fits = pbase != 0xffffffffu;
    for (int k = 0; k < wave; ++k) { pbase += sh_e[k]; hbase += sh_e[NW + k]; }
    uint32_t woff = pbase + (incl & 0xffffu) - lane_words, hoff = hbase + (incl >> 16) - lane_heads;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = w.s0 + (uint32_t)j;
        const bool own = live && lane >= 1 && lane <= 62 && w.active && s < S;
        uint32_t my_off = 0xffffffffu;
        if (need_rec[j] && fits) {
            my_off = woff;
            const bool first = (info[j + 2] & SI_FIRST) != 0;
            const uint32_t hdr = (single[j] ? REC_SINGLE : 0u) | ((info[j + 2] & SI_LAST) ? REC_CTG_LAST : 0u) | (first ? REC_CTG_FIRST : 0u) |
                                 ((info[j + 1] & 0xfu) << 4);
            vl[j].write_record(pool + my_off, s, total[j], hdr, L + j * (E - 2) * 64, lane);
        }
        woff += words[j];
        off4[j] = my_off;
        if (own && !own4) slot_rec[s] = my_off;
        if (is_head[j] && fits) heads[hoff] = my_off;
        hoff += is_head[j] ? 1u : 0u;
    }
    if (own4) *reinterpret_cast<uint4*>(slot_rec + w.s0) = make_uint4(off4[0], off4[1], off4[2], off4[3]);
    phase(5);
    if (dbg && lane == 0) {
        unsigned long long* sh = dbg + 16 * ((blockIdx.x * NW + wave) & 63u);
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&sh[k], pc[k]);
        atomicAdd(&sh[6], 1ull);
        atomicAdd(&sh[7], (unsigned long long)n_steps);
        atomicAdd(&sh[8], (unsigned long long)dl_n);
        atomicAdd(&sh[9], (unsigned long long)n_rounds);
        atomicAdd(&sh[10], (unsigned long long)n_hot);
    }
}

// ------------------------------------------------------------------------------------------------
// k_vote: one wave per 62 consecutive slots (+2 halo lanes that only provide left context).
// Lanes are slots, the loop runs over the records overlapping the chunk in file order (wave-uniform),
// so every lane sees its votes in first-seen order without atomics; left context comes from the two
// neighbouring lanes.  Single-state slots (one distinct base voted) are final here: the chain DP can
// neither change them nor couple across them (scores are exact integers), so only multi-state runs
// (plus their terminating single-state slot) are spilled as compact records for k_dp.
template <int E>
__global__ __launch_bounds__(256) void k_vote(const uint4* __restrict__ meta, const uint8_t* __restrict__ rows,
                                              const uint8_t* __restrict__ slot_info, uint32_t S,
                                              const uint32_t* __restrict__ chunk_first,
                                              const uint32_t* __restrict__ chunk_last, uint32_t n_chunks,
                                              const uint32_t* __restrict__ redo_in, uint32_t n_redo_in,
                                              uint16_t* __restrict__ slot_res, uint32_t* __restrict__ slot_rec,
                                              uint32_t* __restrict__ pool, uint32_t pool_cap,
                                              uint32_t* __restrict__ counters, uint32_t* __restrict__ heads,
                                              uint32_t* __restrict__ redo_out, uint32_t redo_ci, uint32_t flag_single,
                                              uint32_t* __restrict__ hbm_lists) {
    extern __shared__ uint32_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t ci = blockIdx.x * 4 + wave;
    uint32_t c;
    if (redo_in) {
        if (ci >= n_redo_in) return;
        c = redo_in[ci];
    } else {
        c = ci;
        if (c >= n_chunks) return;
    }
    // the wave's context lists: LDS, or (E = every possible context) its stretch of the HBM scratch
    uint32_t* L = hbm_lists ? hbm_lists + (size_t)ci * (E - 2) * 64 : lds + wave * (E - 2) * 64;
    const int64_t s64 = (int64_t)c * VOTE_CH - 2 + lane;
    const bool valid = s64 >= 0 && s64 < (int64_t)S;
    const uint32_t s = (uint32_t)s64;
    const uint32_t info = valid ? slot_info[s] : 0u;
    const uint32_t dsym = info & 0xf;
    const bool first = (info & SI_FIRST) != 0;
    // the draft votes once per slot with its own rolling context (contig.c:373-383)
    uint32_t d1 = __shfl_up(dsym, 1), d2 = __shfl_up(dsym, 2);
    const uint32_t f1 = __shfl_up((uint32_t)first, 1);
    const uint32_t prev_dsym = d1;
    if (first) { d1 = 0; d2 = 0; }
    else if (f1) d2 = 0;
    VoteLane<E> vl;
    vl.init(d2 << 8 | d1 << 4 | dsym);
    uint32_t basemask = 1u << dsym;
    const uint32_t r0 = chunk_first[c], r1 = chunk_last[c];
    if (r0 != 0xffffffffu) {
        for (uint32_t r = r0; r <= r1; ++r) {
            const uint4 m = meta[r];
            const bool cov = valid && s >= m.x && s <= m.y;
            uint32_t sym = 0;
            if (cov) {
                uint32_t nn = s - m.z;
                uint32_t byte = rows[(uint64_t)m.w * 4 + (nn >> 1)];
                sym = (byte >> ((nn & 1) * 4)) & 0xf;
            }
            uint32_t p1 = __shfl_up(sym, 1), p2 = __shfl_up(sym, 2);
            if (lane < 1) p1 = 0;
            if (lane < 2) p2 = 0;
            if (cov) {
                basemask |= 1u << sym;
                if (lane >= 2) vl.tally(p2 << 8 | p1 << 4 | sym, L, lane);
            }
        }
    }
    vote_epilogue<E>(vl, L, lane, c, valid, s, info, dsym, first, prev_dsym, basemask, S, slot_res, slot_rec, pool, pool_cap,
                     counters, heads, redo_out, redo_ci, flag_single);
}

// ------------------------------------------------------------------------------------------------
// k_dp: one lane per multi-state run
constexpr int DP_T = 64;
struct DpLds {
    long long (*sc_)[16][DP_T];
    uint16_t (*km_)[16][DP_T];
    uint8_t (*rk_)[16][DP_T];
    int t;
    __device__ __forceinline__ long long& sc(int buf, uint32_t b) { return sc_[buf][b][t]; }
    __device__ __forceinline__ uint16_t& km(int buf, uint32_t b) { return km_[buf][b][t]; }
    __device__ __forceinline__ uint8_t& rk(int buf, uint32_t b) { return rk_[buf][b][t]; }
};

template <bool FP>
__global__ __launch_bounds__(DP_T) void k_dp(const uint32_t* __restrict__ heads, const uint32_t* __restrict__ counters,
                                             uint32_t cnt0, uint32_t n_shards, uint32_t heads_region,
                                             uint32_t* __restrict__ pool, const uint32_t* __restrict__ slot_rec,
                                             uint16_t* __restrict__ slot_res, int K, long long Rfix, double rate, double min_ratio,
                                             uint32_t* __restrict__ err) {
    __shared__ long long sc[2][16][DP_T];
    __shared__ uint16_t km[2][16][DP_T];
    __shared__ uint8_t rk[2][16][DP_T];
    DpLds st{sc, km, rk, (int)threadIdx.x};
    uint32_t n_heads = 0;
    for (uint32_t sh = 0; sh < n_shards; ++sh) n_heads += counters[cnt0 + sh];
    for (uint32_t hi = blockIdx.x * DP_T + threadIdx.x; hi < n_heads; hi += gridDim.x * DP_T) {
        uint32_t k = hi, sh = 0;   // run-head lists are sharded like the record pool
        while (sh + 1 < n_shards && k >= counters[cnt0 + sh]) { k -= counters[cnt0 + sh]; ++sh; }
        if (!dp_run<FP>(heads[sh * heads_region + k], pool, slot_rec, slot_res, K, Rfix, rate, min_ratio, st))
            atomicOr(err, ERR_DP_INCONSISTENT);
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void k_fixfirst(const uint32_t* __restrict__ ctg_off, uint32_t nc, const uint32_t* __restrict__ soff,
                           const uint8_t* __restrict__ slot_info, uint16_t* __restrict__ slot_res) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    fixfirst_contig(ctg_off[c], ctg_off[c + 1], soff, slot_info, slot_res);
}

__global__ __launch_bounds__(256) void k_emit(const uint16_t* __restrict__ slot_res, const uint8_t* __restrict__ slot_info,
                                              const uint32_t* __restrict__ opos, uint32_t S, uint32_t mask,
                                              uint8_t* __restrict__ out) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    emit_slot(s, slot_res, slot_info, opos, mask, out);
}

__global__ void k_contig_bounds(const uint32_t* __restrict__ ctg_off, uint32_t nc, const uint32_t* __restrict__ soff,
                                const uint32_t* __restrict__ opos, uint32_t* __restrict__ out_bounds) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > nc) return;
    out_bounds[c] = opos[soff[ctg_off[c]]];
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (host)
static inline unsigned nblk(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

void launch_prep(hipStream_t st, const ReadsDev& R, int64_t n_reads, const uint32_t* ctg_off, int trim, int32_t* qs,
                 int32_t* qe, int32_t* span, uint32_t* ins, uint32_t* counters) {
    if (n_reads == 0) return;
    k_prep<<<nblk(n_reads, 256), 256, 0, st>>>(R, n_reads, ctg_off, trim, qs, qe, span, ins, counters);
}

template <class F, class OutT>
static void scan_impl(hipStream_t st, F f, uint64_t n, OutT* out, uint64_t* tile_tmp, uint64_t* total_dev) {
    unsigned tiles = nblk(n, SCAN_TILE);
    if (tiles == 0) tiles = 1;
    k_scan_reduce<F><<<tiles, SCAN_T, 0, st>>>(f, n, tile_tmp);
    k_scan_tiles<<<1, 1024, 0, st>>>(tile_tmp, tiles, total_dev);
    k_scan_final<F, OutT><<<tiles, SCAN_T, 0, st>>>(f, n, tile_tmp, out, total_dev);
}
uint64_t scan_tmp_words(uint64_t n) { return nblk(n, SCAN_TILE) + 2; }

void launch_scan_slots(hipStream_t st, const uint32_t* ins, uint64_t G, uint32_t* soff, uint64_t* tmp, uint64_t* total) {
    scan_impl<LoadInsPlus1, uint32_t>(st, LoadInsPlus1{ins}, G, soff, tmp, total);
}
void launch_scan_rows(hipStream_t st, const uint32_t* cap_bytes, uint64_t n, uint64_t* rowoff, uint64_t* tmp, uint64_t* total) {
    scan_impl<LoadU32, uint64_t>(st, LoadU32{cap_bytes}, n, rowoff, tmp, total);
}
void launch_scan_keep(hipStream_t st, const uint16_t* slot_res, uint64_t S, uint32_t* opos, uint64_t* tmp, uint64_t* total) {
    scan_impl<LoadKeep, uint32_t>(st, LoadKeep{slot_res}, S, opos, tmp, total);
}
struct LoadU8 {
    const uint8_t* p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
void launch_scan_u8(hipStream_t st, const uint8_t* v, uint64_t n, uint32_t* out, uint64_t* tmp, uint64_t* total) {
    scan_impl<LoadU8, uint32_t>(st, LoadU8{v}, n, out, tmp, total);
}
// the per-record arrays of a dense record stream rebuilt on the device instead of uploaded (np1_device.hip:fill_batch): pool offsets
// as running sums of the CIGAR lengths / packed base bytes, and the contig of every record from the contigs' record ranges
struct LoadNcig {
    const uint32_t* p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
struct LoadSeqBytes {
    const int32_t* p;
    __device__ uint64_t operator()(uint64_t i) const { return ((uint64_t)(uint32_t)p[i] + 1) >> 1; }
};
__global__ __launch_bounds__(256) void k_record_contig(const uint64_t* __restrict__ read_begin, uint32_t nc, uint64_t n, uint32_t* __restrict__ ctg) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint32_t lo = 0, hi = nc;            // largest c with read_begin[c] <= r
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (read_begin[mid] <= r) lo = mid; else hi = mid;
    }
    ctg[r] = lo;
}
__global__ __launch_bounds__(256) void k_widen_u16(const uint16_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
// 4 bytes of seq2 (16 bases) -> 8 bytes of seq per lane: coalesced 4-byte loads, 8-byte stores
__global__ __launch_bounds__(256) void k_unpack_seq2(const uint8_t* __restrict__ seq2, uint64_t n2, uint8_t* __restrict__ seq) {
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;      // word of seq2
    if (4 * w >= n2) return;
    uint32_t v;
    if (4 * w + 4 <= n2) v = reinterpret_cast<const uint32_t*>(seq2)[w];
    else { v = 0; for (uint64_t k = 4 * w; k < n2; ++k) v |= (uint32_t)seq2[k] << (8 * (k - 4 * w)); }
    uint32_t out[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t b = (v >> (16 * h + 8 * k)) & 0xffu;       // seq2 byte 4w + 2h + k: two bytes of seq
            const uint32_t hi = ((1u << ((b >> 6) & 3u)) << 4) | (1u << ((b >> 4) & 3u));
            const uint32_t lo = ((1u << ((b >> 2) & 3u)) << 4) | (1u << (b & 3u));
            o |= (hi | lo << 8) << (16 * k);
        }
        out[h] = o;
    }
    uint2* dst = reinterpret_cast<uint2*>(seq) + w;                    // (the buffer holds 2 * n2 rounded up to 8 bytes)
    *dst = make_uint2(out[0], out[1]);
}
__global__ __launch_bounds__(256) void k_patch_seq(const uint64_t* __restrict__ at, const uint8_t* __restrict__ val, uint64_t n, uint8_t* __restrict__ seq) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) seq[at[i]] = val[i];
}
// one lane per byte of draft4 = two characters ("ACGT"[code] | 0x20 when lower)
__global__ __launch_bounds__(256) void k_unpack_draft4(const uint8_t* __restrict__ d4, uint64_t G, uint8_t* __restrict__ draft) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (2 * j >= G) return;
    const uint32_t v = d4[j];
    const uint32_t lut = 0x54474341u;      // 'A' 'C' 'G' 'T'
    const uint32_t a = ((lut >> (8 * ((v >> 4) & 3u))) & 0xffu) | ((v >> 6) & 1u) << 5;
    const uint32_t b = ((lut >> (8 * (v & 3u))) & 0xffu) | ((v >> 2) & 1u) << 5;
    draft[2 * j] = (uint8_t)a;
    if (2 * j + 1 < G) draft[2 * j + 1] = (uint8_t)b;
}
void launch_unpack_draft4(hipStream_t st, const uint8_t* d4, uint64_t G, uint8_t* draft, const uint64_t* esc_at, const uint8_t* esc_val, uint64_t n_esc) {
    if (G) k_unpack_draft4<<<nblk((G + 1) / 2, 256), 256, 0, st>>>(d4, G, draft);
    if (n_esc) k_patch_seq<<<nblk(n_esc, 256), 256, 0, st>>>(esc_at, esc_val, n_esc, draft);
}
void launch_unpack_seq2(hipStream_t st, const uint8_t* seq2, uint64_t n2, uint8_t* seq, const uint64_t* esc_at, const uint8_t* esc_val, uint64_t n_esc) {
    if (n2) k_unpack_seq2<<<nblk((n2 + 3) / 4, 256), 256, 0, st>>>(seq2, n2, seq);
    if (n_esc) k_patch_seq<<<nblk(n_esc, 256), 256, 0, st>>>(esc_at, esc_val, n_esc, seq);
}
// ---- the compact upload form of the per-record fields, undone (np1_priv.h: np1_stream::Compact)
struct LoadNotPlain {
    const uint32_t* bits;
    __device__ uint64_t operator()(uint64_t i) const { return ((bits[i >> 5] >> (i & 31u)) & 1u) ^ 1u; }
};
struct LoadPosEsc {
    const uint8_t* d;
    __device__ uint64_t operator()(uint64_t i) const { return d[i] == 255 ? 1ull : 0ull; }
};
struct LoadPosStep {
    const uint8_t* d;
    __device__ uint64_t operator()(uint64_t i) const { return d[i] == 255 ? 0ull : (uint64_t)d[i]; }
};
// work: xidx[n + 1] | pidx[n + 1] | steps[n + 1] | esc_rec[n_xpos] (8-byte words each; x_cigoff[nx + 1] reuses pidx.. later)
__global__ __launch_bounds__(256) void k_expand_a(CompactDev c, const uint64_t* __restrict__ xidx, const uint64_t* __restrict__ pidx, uint32_t* __restrict__ ncig,
                                                  int32_t* __restrict__ lq, uint64_t* __restrict__ esc_rec) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= c.n) return;
    const bool plain = (c.plain[r >> 5] >> (r & 31u)) & 1u;
    if (plain) { ncig[r] = 1u; lq[r] = (int32_t)c.common_lq; }
    else { const uint64_t k = xidx[r]; ncig[r] = c.x_ncig[k]; lq[r] = c.x_lq[k]; }
    if (c.dpos[r] == 255) esc_rec[pidx[r]] = r;
}
__global__ __launch_bounds__(256) void k_expand_pos(CompactDev c, const uint64_t* __restrict__ pidx, const uint64_t* __restrict__ steps, const uint64_t* __restrict__ esc_rec,
                                                    int32_t* __restrict__ pos) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= c.n) return;
    const bool esc = c.dpos[r] == 255;
    const uint64_t k = pidx[r] + (esc ? 1u : 0u) - 1u;      // the last full position at or before r (record 0 carries one)
    const uint64_t e = esc_rec[k];
    // steps[] are exclusive sums, and a record with a full position contributes no step: the steps from e + 1 to r inclusive
    const uint64_t upto_r = steps[r] + (esc ? 0ull : (uint64_t)c.dpos[r]);
    pos[r] = c.x_pos[k] + (int32_t)(upto_r - steps[e]);
}
__global__ __launch_bounds__(256) void k_expand_cigar(CompactDev c, const uint64_t* __restrict__ xidx, const uint64_t* __restrict__ x_cigoff, const uint32_t* __restrict__ ncig,
                                                      const uint64_t* __restrict__ cigoff, uint32_t* __restrict__ cigar) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= c.n) return;
    const bool plain = (c.plain[r >> 5] >> (r & 31u)) & 1u;
    uint32_t* dst = cigar + cigoff[r];
    if (plain) { dst[0] = c.common_lq << 4; return; }
    const uint32_t* src = c.x_cigar + x_cigoff[xidx[r]];
    const uint32_t m = ncig[r];
    for (uint32_t j = 0; j < m; ++j) dst[j] = src[j];
}
void launch_expand_records(hipStream_t st, const CompactDev& c, int32_t* pos, uint32_t* ncig, int32_t* lq, uint64_t* work, uint64_t* tmp, uint64_t* total) {
    if (c.n == 0) return;
    uint64_t* xidx = work, *pidx = work + (c.n + 1), *steps = work + 2 * (c.n + 1), *esc_rec = work + 3 * (c.n + 1);
    scan_impl<LoadNotPlain, uint64_t>(st, LoadNotPlain{c.plain}, c.n, xidx, tmp, total);
    scan_impl<LoadPosEsc, uint64_t>(st, LoadPosEsc{c.dpos}, c.n, pidx, tmp, total);
    scan_impl<LoadPosStep, uint64_t>(st, LoadPosStep{c.dpos}, c.n, steps, tmp, total);
    k_expand_a<<<nblk(c.n, 256), 256, 0, st>>>(c, xidx, pidx, ncig, lq, esc_rec);
    k_expand_pos<<<nblk(c.n, 256), 256, 0, st>>>(c, pidx, steps, esc_rec, pos);
}
void launch_expand_cigars(hipStream_t st, const CompactDev& c, const uint32_t* ncig, const uint64_t* cigoff, uint32_t* cigar, uint64_t* work, uint64_t* tmp, uint64_t* total) {
    if (c.n == 0) return;
    uint64_t* xidx = work, *x_cigoff = work + (c.n + 1);      // (pidx is done with)
    if (c.nx) scan_impl<LoadU32, uint64_t>(st, LoadU32{c.x_ncig}, c.nx, x_cigoff, tmp, total);
    k_expand_cigar<<<nblk(c.n, 256), 256, 0, st>>>(c, xidx, x_cigoff, ncig, cigoff, cigar);
}
void launch_widen_u16(hipStream_t st, const uint16_t* src, uint32_t* dst, uint64_t n) {
    if (n) k_widen_u16<<<nblk(n, 256), 256, 0, st>>>(src, dst, n);
}
void launch_record_offsets(hipStream_t st, const uint32_t* ncig, const int32_t* lq, const uint64_t* read_begin, uint32_t nc, uint64_t n, uint64_t* cigoff,
                           uint64_t* seqoff, uint32_t* ctg, uint64_t* tmp, uint64_t* total) {
    if (n == 0) return;
    scan_impl<LoadNcig, uint64_t>(st, LoadNcig{ncig}, n, cigoff, tmp, total);
    scan_impl<LoadSeqBytes, uint64_t>(st, LoadSeqBytes{lq}, n, seqoff, tmp, total);
    k_record_contig<<<nblk(n, 256), 256, 0, st>>>(read_begin, nc, n, ctg);
}

void launch_scan_u32(hipStream_t st, const uint32_t* v, uint64_t n, uint32_t* out, uint64_t* tmp, uint64_t* total) {
    scan_impl<LoadU32, uint32_t>(st, LoadU32{v}, n, out, tmp, total);
}

void launch_slotinfo(hipStream_t st, const uint8_t* draft, uint32_t G, const uint32_t* ctg_off, uint32_t nc,
                     const uint32_t* soff, uint8_t* slot_info, uint32_t* slot_g) {
    if (G == 0) return;
    k_slotinfo<<<nblk(G, 256), 256, 0, st>>>(draft, G, ctg_off, nc, soff, slot_info, slot_g);
}

void launch_rowcap(hipStream_t st, const ReadsDev& R, int64_t n_reads, const uint32_t* ctg_off, const uint32_t* soff,
                   const int32_t* qs, const int32_t* qe, const int32_t* span, uint32_t* rbase, uint32_t* cap_bytes) {
    if (n_reads == 0) return;
    k_rowcap<<<nblk(n_reads, 256), 256, 0, st>>>(R, n_reads, ctg_off, soff, qs, qe, span, rbase, cap_bytes);
}

void launch_rows(hipStream_t st, const ReadsDev& R, int64_t n_reads, const uint32_t* ctg_off, const uint32_t* soff,
                 const int32_t* qs, const int32_t* qe, const uint32_t* rbase, const uint64_t* rowoff, uint8_t* rows,
                 uint4* meta, uint32_t* chunk_first, uint32_t* chunk_last, unsigned long long* votes) {
    if (n_reads == 0) return;
    k_rows<<<nblk(n_reads, 256), 256, 0, st>>>(R, n_reads, ctg_off, soff, qs, qe, rbase, rowoff, rows, meta,
                                               chunk_first, chunk_last, votes);
}

void launch_vote(hipStream_t st, int E, const uint4* meta, const uint8_t* rows, const uint8_t* slot_info, uint32_t S,
                 const uint32_t* chunk_first, const uint32_t* chunk_last, uint32_t n_chunks, const uint32_t* redo_in,
                 uint32_t n_redo_in, uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool, uint32_t pool_cap,
                 uint32_t* counters, uint32_t* heads, uint32_t* redo_out, uint32_t redo_ci, uint32_t flag_single, uint32_t* hbm_lists) {
    uint32_t work = redo_in ? n_redo_in : n_chunks;
    if (work == 0) return;
    unsigned blocks = nblk(work, 4);
    if (E > 160) {   // hbm_lists: vote_hbm_list_words() words per chunk of the redo list
        k_vote<VOTE_E_ALL><<<blocks, 256, 0, st>>>(meta, rows, slot_info, S, chunk_first, chunk_last, n_chunks, redo_in, n_redo_in, slot_res, slot_rec, pool,
                                                  pool_cap, counters, heads, redo_out, redo_ci, flag_single, hbm_lists);
        return;
    }
#define NP1_VOTE(EE)                                                                                              \
    k_vote<EE><<<blocks, 256, 4 * ((EE)-2) * 64 * sizeof(uint32_t), st>>>(                                        \
        meta, rows, slot_info, S, chunk_first, chunk_last, n_chunks, redo_in, n_redo_in, slot_res, slot_rec, pool, \
        pool_cap, counters, heads, redo_out, redo_ci, flag_single, nullptr)
    if (E <= 16) NP1_VOTE(16);
    else if (E <= 64) NP1_VOTE(64);
    else NP1_VOTE(160);
#undef NP1_VOTE
}

void launch_desc(hipStream_t st, const ReadsDev& R, int64_t n_reads, const uint32_t* ctg_off, const uint32_t* soff,
                 const int32_t* qs, const int32_t* qe, uint32_t* desc, uint32_t* ovf_pool, uint32_t ovf_cap,
                 uint32_t* chunk_first, uint32_t* chunk_last, uint32_t* counters) {
    if (n_reads == 0) return;
    k_desc<<<nblk(n_reads, 256), 256, 0, st>>>(R, n_reads, ctg_off, soff, qs, qe, desc, ovf_pool, ovf_cap, chunk_first,
                                               chunk_last, counters);
}
static uint32_t ablate_env() {   // timing experiments only (results are wrong when set)
    static const uint32_t v = getenv("NP1_ABLATE") ? (uint32_t)atoi(getenv("NP1_ABLATE")) : 0u;
    return v;
}

int launch_tile3(hipStream_t st, int level, const ReadsDev& R, const uint32_t* soff, const uint32_t* desc,
                 const uint32_t* ovf_pool, const uint32_t* chunk_first, const uint32_t* chunk_last, uint32_t n_chunks, const uint32_t* redo_in,
                 uint32_t n_redo_in, const uint8_t* slot_info, const uint32_t* slot_g, uint32_t S, uint32_t max_lq,
                 uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool, uint32_t pool_cap, uint32_t* counters,
                 uint32_t* heads, uint32_t heads_cap, uint32_t* redo_out, uint32_t redo_ci, uint32_t flag_single,
                 unsigned long long* votes) {
    const uint32_t seq_w = (((max_lq + 1) >> 1) + 3) / 4 + 1;   // packed bases per record, in words (upper bound)
    const uint32_t per = (uint32_t)DESC_WORDS + seq_w;
    const uint32_t ablate = ablate_env();
#define NP1_TILE3(EE, NWW, BUDGET)                                                                                   \
    do {                                                                                                             \
        const uint32_t fixed = (uint32_t)(NWW) * (uint32_t)((EE)-2) * 64u + 8u + (uint32_t)DESC_WORDS;               \
        uint32_t budget = (BUDGET);                                                                                  \
        if (budget < fixed + per) budget = fixed + per;                                                              \
        if (budget > 40960u - 64u) return -1;                                                                        \
        uint32_t nb_max = (budget - fixed) / per;                                                                    \
        if (nb_max > 512u) nb_max = 512u;                                                                            \
        const uint32_t bytes = (fixed + nb_max * per) * 4u;                                                          \
        uint32_t items = redo_in ? n_redo_in : (n_chunks + (NWW)-1) / (NWW);                                          \
        if (items == 0) return 0;                                                                                    \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile3<EE, NWW>),                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);                 \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        k_tile3<EE, NWW><<<items, (NWW)*64, bytes, st>>>(R, soff, desc, ovf_pool, chunk_first, chunk_last, n_chunks,    \
                                                          redo_in,                                                   \
                                                          items, slot_info, slot_g, S, seq_w, nb_max, slot_res,      \
                                                          slot_rec, pool, pool_cap, counters, heads, heads_cap,      \
                                                          redo_out, redo_ci, flag_single, votes, ablate);            \
    } while (0)
    if (level == 0) NP1_TILE3(8, 8, 13312u);        // 52 KiB: three workgroups per CU
    else if (level == 1) NP1_TILE3(64, 1, 13312u);
    else NP1_TILE3(160, 1, 24576u);
#undef NP1_TILE3
    return 0;
}

int launch_tile9(hipStream_t st, const ReadsDev& R, const uint32_t* soff, const uint32_t* desc, const uint32_t* ovf_pool, const uint32_t* chunk_first,
                 const uint32_t* chunk_last, uint32_t n_chunks, const uint8_t* slot_info, const uint32_t* slot_g, uint32_t S, uint32_t max_lq,
                 uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool, uint32_t pool_cap, uint32_t* counters, uint32_t* heads, uint32_t heads_cap,
                 uint32_t* redo_out, uint32_t redo_ci, uint32_t flag_single, unsigned long long* votes, unsigned long long* dbg) {
    constexpr int E9 = 6, NW9 = 4;      // (a slot with more than six contexts sends its wave's chunks to k_tile3<64>)
    const uint32_t seq_w = (((max_lq + 1) >> 1) + 3) / 4 + 1;   // packed bases per record, in words (upper bound)
    const uint32_t fixed = (uint32_t)NW9 * t9_wave_words<E9>() + 8u + (uint32_t)DESC_WORDS;
    const uint32_t per = (uint32_t)DESC_WORDS + seq_w;
    static const uint32_t budget_env = getenv("NP1_TILE9_LDS_WORDS") ? (uint32_t)atoi(getenv("NP1_TILE9_LDS_WORDS")) : 20224u;   // 79 KiB: two workgroups per CU
    uint32_t budget = budget_env;
    if (budget < fixed + 8u * per) budget = fixed + 8u * per;
    if (budget > 40960u - 64u) return -1;
    uint32_t nb_max = (budget - fixed) / per;
    if (nb_max > 512u) nb_max = 512u;
    const uint32_t bytes = (fixed + nb_max * per) * 4u;
    const uint32_t items = (n_chunks + NW9 * T9_CH - 1) / (NW9 * T9_CH);
    if (items == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile9<E9, NW9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        attr_set = true;
    }
    k_tile9<E9, NW9><<<items, NW9 * 64, bytes, st>>>(R, soff, desc, ovf_pool, chunk_first, chunk_last, n_chunks, slot_info, slot_g, S, seq_w, nb_max, slot_res, slot_rec,
                                                      pool, pool_cap, counters, heads, heads_cap, redo_out, redo_ci, flag_single, votes, dbg);
    return 0;
}

// ---- intra-contig tiling: which slots left the vote with one state (the chain restarts behind each of them), and
// what a tile needs to know to be joined: is there such a slot in each halo, and where do its own bases start / end in the output
__global__ __launch_bounds__(256) void k_single_map(const uint16_t* __restrict__ slot_res, uint32_t S, uint8_t* __restrict__ single) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) single[s] = (slot_res[s] & 0xffu) != 0xffu ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_join_info(const uint32_t* __restrict__ soff, const uint8_t* __restrict__ single, const uint32_t* __restrict__ opos,
                                                   uint32_t i_elo, uint32_t i_a, uint32_t i_b, uint32_t i_ehi, uint32_t skip, uint32_t* __restrict__ out) {
    __shared__ uint32_t any[2];
    if (threadIdx.x < 2) any[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t l0 = soff[i_elo] + skip, l1 = soff[i_a], r0 = soff[i_b], r1 = soff[i_ehi];
    bool a = false, b = false;
    for (uint32_t s = l0 + threadIdx.x; s < l1; s += 256) a = a || single[s] != 0;
    for (uint32_t s = r0 + threadIdx.x; s < r1; s += 256) b = b || single[s] != 0;
    if (a) any[0] = 1;
    if (b) any[1] = 1;
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = any[0]; out[1] = any[1]; out[2] = opos[soff[i_a]]; out[3] = opos[soff[i_b]]; }
}
void launch_single_map(hipStream_t st, const uint16_t* slot_res, uint32_t S, uint8_t* single) {
    if (S) k_single_map<<<nblk(S, 256), 256, 0, st>>>(slot_res, S, single);
}
void launch_join_info(hipStream_t st, const uint32_t* soff, const uint8_t* single, const uint32_t* opos, uint32_t i_elo, uint32_t i_a, uint32_t i_b, uint32_t i_ehi,
                      uint32_t skip, uint32_t* out) {
    k_join_info<<<1, 256, 0, st>>>(soff, single, opos, i_elo, i_a, i_b, i_ehi, skip, out);
}

void launch_dp(hipStream_t st, const uint32_t* heads, uint32_t* counters, uint32_t cnt0, uint32_t n_shards,
               uint32_t heads_region, uint32_t* pool, const uint32_t* slot_rec, uint16_t* slot_res, int K, long long Rfix,
               double min_ratio, uint32_t grid, bool fp, double rate) {
    if (fp)
        k_dp<true><<<grid, DP_T, 0, st>>>(heads, counters, cnt0, n_shards, heads_region, pool, slot_rec, slot_res, K, Rfix, rate, min_ratio,
                                          &counters[CNT_ERR]);
    else
        k_dp<false><<<grid, DP_T, 0, st>>>(heads, counters, cnt0, n_shards, heads_region, pool, slot_rec, slot_res, K, Rfix, rate, min_ratio,
                                           &counters[CNT_ERR]);
}

void launch_fixfirst(hipStream_t st, const uint32_t* ctg_off, uint32_t nc, const uint32_t* soff, const uint8_t* slot_info,
                     uint16_t* slot_res) {
    if (nc == 0) return;
    k_fixfirst<<<nblk(nc, 64), 64, 0, st>>>(ctg_off, nc, soff, slot_info, slot_res);
}

void launch_emit(hipStream_t st, const uint16_t* slot_res, const uint8_t* slot_info, const uint32_t* opos, uint32_t S,
                 uint32_t mask, uint8_t* out) {
    if (S == 0) return;
    k_emit<<<nblk(S, 256), 256, 0, st>>>(slot_res, slot_info, opos, S, mask, out);
}

void launch_contig_bounds(hipStream_t st, const uint32_t* ctg_off, uint32_t nc, const uint32_t* soff, const uint32_t* opos,
                          uint32_t* out_bounds) {
    k_contig_bounds<<<nblk(nc + 1, 64), 64, 0, st>>>(ctg_off, nc, soff, opos, out_bounds);
}

}  // namespace np1k
