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
#include "np1_events.h"
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
    // thread t owns SCAN_ITEMS consecutive elements
    uint64_t first = base + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = first + k;
        v[k] = i < n ? f(i) : 0;
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
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = first + k;
        if (i < n) out[i] = (OutT)run;
        run += v[k];
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

// the draft's symbols as 4-bit codes, two per byte, first base in the high nibble (the layout of the reads' bases): one lane per byte
__global__ __launch_bounds__(256) void k_dpack(const uint8_t* __restrict__ draft, uint32_t G, uint8_t* __restrict__ dpack) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * (uint64_t)i >= G) return;
    uint32_t a = draft[2 * i], b = 2 * i + 1 < G ? draft[2 * i + 1] : (uint32_t)'=';
    if (a >= 97 && a <= 122) a -= 32;
    if (b >= 97 && b <= 122) b -= 32;
    dpack[i] = (uint8_t)(draft_code(a) << 4 | draft_code(b));
}

// ------------------------------------------------------------------------------------------------
// k_desc (fused pipeline, default): one lane per record -> its descriptor (np1_core.h build_desc) and the
// candidate record range of every vote chunk (wave-aggregated min/max instead of per-record atomics)
__global__ __launch_bounds__(256) void k_desc(ReadsDev R, int64_t n_reads, const uint32_t* __restrict__ ctg_off,
                                              const uint32_t* __restrict__ soff, const int32_t* __restrict__ qs,
                                              const int32_t* __restrict__ qe, uint32_t* __restrict__ desc,
                                              uint32_t* __restrict__ ovf_pool, uint32_t ovf_cap,
                                              uint32_t* __restrict__ chunk_first, uint32_t* __restrict__ chunk_last,
                                              uint32_t* __restrict__ counters, const uint8_t* __restrict__ dpack,
                                              uint32_t* __restrict__ dirty) {
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
        // where the record disagrees with the draft (k_tile8 evaluates symbols only there)
        if (dirty) dirty[r] = desc_dirty_chunks(mine, R.seq + R.seq_off[r], dpack, SoGlobal{soff});
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
                                          uint32_t* L, uint32_t& nvotes, uint32_t ablate = 0) {
    const bool cov = valid && s >= d[0] && s <= d[1];
    if (cov) rsym = (ablate & 8u) ? (g & 0xfu) : desc_symbol(d, g, jj, sq);
    const uint32_t p1 = wave_shr1(rsym), p2 = wave_shr1(p1);
    if (cov) {
        basemask |= 1u << rsym;
        if (lane >= 2 && !(ablate & 4u)) { vl.tally(p2 << 8 | p1 << 4 | rsym, L, lane); ++nvotes; }
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
    uint32_t nvotes = 0;
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
                            nvotes += vote ? 1u : 0u;
                            const bool rest = vote && !m0 && !m1;
                            if (__ballot(rest) != 0ull) {
                                if (rest) vl.tally(k, L, lane);   // a context seen for the first time, or one kept in the LDS list
                            }
                        } else {
                            const uint32_t* d = dsc + i * DESC_WORDS;   // LDS: every access below is a ds_read broadcast
                            const SeqLds sq{seqb + (h.w - sq0_lo)};
                            uint32_t rsym = 0;   // this record's symbol at my slot (kept across the parts of a chained record)
                            vote_part<E>(d, sq, valid, s, g, jj, lane, rsym, basemask, vl, L, nvotes);
                            uint32_t nx = d[DESC_NEXT];
                            while (nx) {   // rare: record with more indel operations than one descriptor holds; parts live in HBM
                                const uint32_t* dg = ovf_pool + (uint64_t)(nx - 1) * DESC_WORDS;
                                vote_part<E>(dg, sq, valid, s, g, jj, lane, rsym, basemask, vl, L, nvotes);
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
    for (int o = 32; o > 0; o >>= 1) nvotes += __shfl_down(nvotes, o);
    const bool ovf_any = chunk_ok && __ballot(vl.ovf) != 0ull;
    const bool single = __popc(basemask) == 1;
    const uint32_t psingle = wave_shr1((uint32_t)single);   // every lane must execute the DPP move: keep it out of the || below
    const bool prev_is_single = first || psingle != 0;
    // flag_single bit 8 (FLAG_ALL_RECORDS): general-rate path, every slot spills a record (np1_core.h:dp_run<true>)
    tile_epilogue<E, NW>(vl, L, tid, c, chunk_ok && !ovf_any, ovf_any, valid, s, info, dsym, first, prev_dsym, single, prev_is_single,
                         vl.total(L, lane), slot_res, slot_rec, pool, pool_cap, counters, heads, heads_cap, redo_out, redo_ci,
                         flag_single & 0xffu, nvotes, votes, (flag_single & FLAG_ALL_RECORDS) != 0);
}

// ------------------------------------------------------------------------------------------------
// k_tile7 (default since round 3): the per-vote kernel with the bookkeeping of a (record, 64-slot chunk) step moved to the SCALAR unit.
// k_tile3 spends ~75 vector and ~60 scalar instructions per step and is bound by vector issue.  What a record does to a chunk is
// nearly always the same thing in every lane -- "covered, and the same three bases as the draft" -- so here it is kept as 64-bit lane
// masks in scalar registers:
//   C   lanes the record covers      = one bit field (s_bfm_b64) from its slot run [sfirst, slast]: no per-lane compares
//   A   lanes where it votes the draft's own symbol (one v_cmp that writes the mask)
//   M0  lanes whose whole 3-base context is the draft's = A & (A << 1 | F1) & (A << 2 | F2)   (F1 / F2: lanes at a contig start, where
//       the context has no first / second predecessor)
// count(draft context) += M0 is one v_addc with the mask as carry-in; "did anything but the draft's base vote here" (which decides
// whether the slot needs the chain DP at all) is one s_or per record.  Only when some lane's context differs (R = votes & ~M0, a few
// steps in ten: draft errors, read errors) do the lanes build their contexts with the two DPP shifts and tally them -- in record
// order, so first-seen order is untouched.  The descriptors are not staged through LDS any more: the step's record is wave-uniform,
// so its descriptor comes through the scalar data cache (s_load) and every field is born in a scalar register; the candidate
// records of a chunk are the range k_desc left in chunk_first / chunk_last.  LDS holds the packed bases only.
__device__ __forceinline__ void add_lane_mask(uint32_t& c, unsigned long long m) {   // c += bit `lane` of m
    unsigned long long carry_out;
    asm volatile("v_addc_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(c), "=s"(carry_out) : "s"(m));
}
__device__ __forceinline__ uint32_t sel_lane_mask(uint32_t if0, uint32_t if1, unsigned long long m) {   // bit `lane` of m ? if1 : if0
    uint32_t r;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(m));
    return r;
}
__device__ __forceinline__ unsigned long long lane_field(int32_t lo, int32_t hi) {   // bits lo..hi (0 <= lo <= hi <= 63)
    const uint32_t n = (uint32_t)(hi - lo + 1);
    return (n >= 64u ? ~0ull : ((1ull << n) - 1ull)) << lo;
}

template <int E, int NW>
__global__ __launch_bounds__(NW * 64) void k_tile7(ReadsDev R, const uint32_t* __restrict__ soff,
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
    uint32_t* dsc = lists + NW * (E - 2) * 64;         // nb_max * DESC_WORDS (+ one spare descriptor slot for the prefetch)
    uint32_t* seqst = dsc + (nb_max + 1) * DESC_WORDS; // nb_max * seq_w + 8
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t item = blockIdx.x;
    if (item >= n_items) return;
    const uint32_t cbase = redo_in ? redo_in[item] : item * NW;
    const uint32_t c = cbase + (uint32_t)wave;
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
    const bool is_ins = valid && (info & SI_INSERT);
    const uint32_t jju = is_ins ? (s - soff[g]) - 1u : 0u;   // insertion column of an insertion slot
    const uint32_t dsym = info & 0xf;
    const bool first = (info & SI_FIRST) != 0;
    uint32_t d1 = wave_shr1(dsym), d2 = wave_shr1(d1);
    const uint32_t f1 = wave_shr1((uint32_t)first);
    const uint32_t prev_dsym = d1;
    if (first) { d1 = 0; d2 = 0; }
    else if (f1) d2 = 0;
    VoteLane<E> vl;
    vl.init(d2 << 8 | d1 << 4 | dsym);
    // wave-uniform lane masks of this chunk
    const unsigned long long INS = __ballot(is_ins);   // (lanes 0 and 1 are the halo: they never vote)
    const unsigned long long F1 = __ballot(first), F2 = __ballot(first || f1 != 0);
    unsigned long long NS = 0;          // lanes where something other than the draft's own symbol voted
    uint32_t basemask = 0;              // the same per lane, from the general path (chained / oversize records)
    const int32_t cs = (int32_t)((int64_t)c * VOTE_CH - 2);   // slot of lane 0 (may be -2 for chunk 0)
    uint32_t wf = 0xffffffffu, wl = 0;
    if (chunk_ok) { wf = chunk_first[c]; wl = chunk_last[c]; }
    wf = (uint32_t)__builtin_amdgcn_readfirstlane((int)wf);
    wl = (uint32_t)__builtin_amdgcn_readfirstlane((int)wl);
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
            const uint32_t sq_fit = sq_quads * 4 <= nb_max * seq_w + 8 ? sq_quads : (nb_max * seq_w + 8) / 4;
            if (sq_fit != sq_quads && tid == 0) atomicOr(&counters[CNT_ERR], ERR_BAD_RECORD);
            {
                const uint4* src = reinterpret_cast<const uint4*>(R.seq + sq0);
                uint4* dst = reinterpret_cast<uint4*>(seqst);
                for (uint32_t i = tid; i < sq_fit; i += NW * 64) dst[i] = src[i];
            }
            const uint32_t sq0_lo = (uint32_t)sq0;   // descriptors carry the low word of their record's pool offset
            __syncthreads();
            // ---- this wave's chunk votes over its candidate records of the batch, in record order
            if (wf != 0xffffffffu) {
                const uint8_t* seqb = reinterpret_cast<const uint8_t*>(seqst);
                const uint64_t ia = wf > rb ? wf : rb, ib = (uint64_t)wl < rb + nb - 1 ? (uint64_t)wl : rb + nb - 1;
                const int32_t lb = ia <= ib ? (int32_t)(ib - rb) : -1;
                const uint32_t la = lb >= 0 ? (uint32_t)(ia - rb) : 0u;
                // descriptor head (sfirst, slast, counts, base offset) and first segment of the next record are fetched while this one
                // votes (LDS broadcast reads; the spare descriptor slot keeps the last prefetch in range); every field becomes scalar
                uint4 hv = *reinterpret_cast<const uint4*>(dsc + la * DESC_WORDS);
                uint2 sv = *reinterpret_cast<const uint2*>(dsc + la * DESC_WORDS + DESC_SEG0);
                for (int32_t li = (int32_t)la; li <= lb; ++li) {
                    const uint32_t* d = dsc + (uint32_t)li * DESC_WORDS;
                    const uint32_t sfirst = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.x), slast_part = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.y);
                    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.z), boff = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.w);
                    const uint32_t seg0_g = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv.x), seg0_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv.y);
                    hv = *reinterpret_cast<const uint4*>(dsc + (uint32_t)(li + 1) * DESC_WORDS);
                    sv = *reinterpret_cast<const uint2*>(dsc + (uint32_t)(li + 1) * DESC_WORDS + DESC_SEG0);
                    if (ablate & 4u) continue;
                    uint32_t sym;
                    unsigned long long C;
                    if (cnt & DESC_SIMPLE) {
                        // ---- the common shape, one matched segment: lanes lo..hi of the slot run, base q = q_lo + (g - g_lo), DEL on the
                        //      insertion columns it passes.  (An empty interval also covers "not in this chunk".)
                        int32_t lo = (int32_t)sfirst - cs, hi = (int32_t)slast_part - cs;
                        lo = lo < 0 ? 0 : lo;
                        hi = hi > 63 ? 63 : hi;
                        if (lo > hi) continue;
                        C = (~0ull >> (63 - hi)) & (~0ull << lo);       // (slots of a run exist: C lies inside VALID)
                        const unsigned long long BASE = C & ~INS;
                        const uint32_t q = sel_lane_mask(0u, g + ((seg0_w >> 16) - seg0_g), BASE);
                        const uint32_t byte = seqb[(boff - sq0_lo) + (q >> 1)];
                        sym = sel_lane_mask(3u, (byte >> ((~q & 1u) << 2)) & 0xfu, BASE);
                    } else {
                        // a record that continues in further parts (DESC_CHAIN) is tested with the end of its whole slot run
                        const uint32_t slast = (cnt & DESC_CHAIN) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)d[DESC_NEXT + 1]) : slast_part;
                        int32_t lo = (int32_t)sfirst - cs, hi = (int32_t)slast - cs;
                        lo = lo < 0 ? 0 : lo;
                        hi = hi > 63 ? 63 : hi;
                        if ((int32_t)(slast - sfirst) < 0 || lo > hi) continue;      // empty run, or not in this chunk
                        C = (~0ull >> (63 - hi)) & (~0ull << lo);
                        const uint32_t rbase = boff - sq0_lo;
                        if (cnt & DESC_CHAIN) {
                            if (ablate & 2u) continue;
                            // indel operations that fill more than one descriptor: the general walk over the parts (they live in HBM)
                            const SeqLds sq{seqb + rbase};
                            const int32_t jj = is_ins ? (int32_t)jju : -1;
                            uint32_t rsym = 0, nv = 0;
                            vote_part<E>(d, sq, valid, s, g, jj, lane, rsym, basemask, vl, L, nv);
                            uint32_t nx = d[DESC_NEXT];
                            while (nx) {
                                const uint32_t* dg = ovf_pool + (uint64_t)(nx - 1) * DESC_WORDS;
                                vote_part<E>(dg, sq, valid, s, g, jj, lane, rsym, basemask, vl, L, nv);
                                nx = dg[DESC_NEXT];
                            }
                            continue;
                        }
                        // ---- segments and insertions of one descriptor: which lanes vote a base of the read, and which one
                        const uint32_t nseg = cnt & 0xffu, nins = (cnt >> 8) & 0xffu;
                        uint32_t q = 0;
                        unsigned long long BASE = 0ull;
                        for (uint32_t k2 = 0; k2 < nseg; ++k2) {
                            const uint2 sk = *reinterpret_cast<const uint2*>(d + DESC_SEG0 + 2 * k2);
                            const uint32_t glo = k2 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)sk.x) : seg0_g;
                            const uint32_t w = k2 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)sk.y) : seg0_w;
                            const uint32_t off = g - glo;
                            const unsigned long long in = __ballot(off < (w & 0xffffu)) & C & ~INS;
                            if ((w >> 16) != 0xffffu) {
                                q = sel_lane_mask(q, (w >> 16) + off, in);
                                BASE |= in;
                            }
                        }
                        for (uint32_t k2 = 0; k2 < nins; ++k2) {
                            const uint2 ik = *reinterpret_cast<const uint2*>(d + DESC_INS0 + 2 * k2);
                            const uint32_t pp = (uint32_t)__builtin_amdgcn_readfirstlane((int)ik.x), w = (uint32_t)__builtin_amdgcn_readfirstlane((int)ik.y);
                            const unsigned long long in = __ballot(g == pp && jju < (w & 0xffffu)) & C & INS;
                            q = sel_lane_mask(q, (w >> 16) + jju, in);
                            BASE |= in;
                        }
                        q = sel_lane_mask(0u, q, BASE);
                        const uint32_t byte = seqb[rbase + (q >> 1)];
                        sym = sel_lane_mask(3u, (byte >> ((~q & 1u) << 2)) & 0xfu, BASE);
                    }
                    // ---- agreement with the draft, all of it on masks
                    const unsigned long long A = __ballot(sym == dsym) & C;
                    NS |= C ^ A;
                    const unsigned long long M0 = A & ((A << 1) | F1) & ((A << 2) | F2);
                    add_lane_mask(vl.c0, M0 & ~3ull);
                    const unsigned long long Rm = (C & ~3ull) & ~M0;
                    if (Rm != 0ull && !(ablate & 1u)) {      // some lane's context is not the draft's: contexts from the neighbours, tallied per lane
                        const uint32_t symc = sel_lane_mask(0u, sym, C);
                        const uint32_t p1 = wave_shr1(symc), p2 = wave_shr1(p1);
                        const uint32_t k = p2 << 8 | p1 << 4 | symc;
                        const unsigned long long M1 = __ballot(k == vl.k1) & Rm;
                        add_lane_mask(vl.c1, M1);
                        const unsigned long long rest = Rm & ~M1;
                        if (rest != 0ull) {
                            if ((rest >> lane) & 1ull) vl.tally(k, L, lane);   // a context seen for the first time, or one kept in the LDS list
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    const bool ovf_any = chunk_ok && __ballot(vl.ovf) != 0ull;
    // a slot is single-state when nothing but the draft's own symbol voted on it
    const bool single = !((NS >> lane) & 1ull) && (basemask & ~(1u << dsym)) == 0u;
    const uint32_t psingle = wave_shr1((uint32_t)single);
    const bool prev_is_single = first || psingle != 0;
    // vote statistic: every tally of an own slot but the draft's own one
    uint32_t nvotes = 0;
    if (valid && lane >= 2) {
        nvotes = vl.c0 + vl.c1 - 1u;
        for (uint32_t e = 2; e < vl.n; ++e) nvotes += L[(e - 2) * 64 + lane] & 0xffffu;
    }
    for (int o = 32; o > 0; o >>= 1) nvotes += __shfl_down(nvotes, o);
    tile_epilogue<E, NW>(vl, L, tid, c, chunk_ok && !ovf_any, ovf_any, valid, s, info, dsym, first, prev_dsym, single, prev_is_single,
                         vl.total(L, lane), slot_res, slot_rec, pool, pool_cap, counters, heads, heads_cap, redo_out, redo_ci,
                         flag_single & 0xffu, nvotes, votes, (flag_single & FLAG_ALL_RECORDS) != 0);
}

// ------------------------------------------------------------------------------------------------
// k_tile8 (default since round 3).  k_tile3 evaluates, for every (record, 64-slot chunk) pair, the record's symbol at every lane and
// compares contexts -- 75 vector + 60 scalar instructions per pair, and the kernel is bound by exactly that instruction issue (one
// vector and one scalar instruction per cycle and CU).  But most pairs are PLAIN: the record repeats the draft on every slot of the
// chunk, so all it does is add 1 to the count of the draft's own context on the lanes it covers (minus the first two slots of its run,
// whose contexts lack predecessors).  Which pairs are plain is known per record before the kernel starts: k_desc leaves the record's
// dirty hull (np1_desc.h: from its first disagreeing vote to two slots behind its last one).  So:
//   phase 1, lanes = 64 candidate records at a time: slot run -> covered lanes C (a 64-bit field per record, built with vector shifts),
//            plain or dirty from the hull, and for plain pairs the finished mask M0 = C & (C << 1 | F1) & (C << 2 | F2) of the lanes
//            whose whole context is the draft's (F1 / F2: lanes at a contig start);
//   phase 2, records in file order (first-seen order of the contexts is the order of this loop): a plain pair is two v_readlane and one
//            v_addc (count += M0); if the record's run starts inside the chunk its first two lanes tally the contexts (0, 0, d) and
//            (0, d', d) straight from the draft's symbols; a dirty pair takes the per-lane evaluation (the body of k_tile7: symbols
//            from the staged bases, agreement as scalar lane masks, DPP contexts only where they differ from the draft's).
// Same staging, same LDS lists, same epilogue, same results as k_tile3.
template <int E, int NW>
__global__ __launch_bounds__(NW * 64) void k_tile8(ReadsDev R, const uint32_t* __restrict__ soff,
                                                   const uint32_t* __restrict__ desc,
                                                   const uint32_t* __restrict__ dirty,
                                                   const uint32_t* __restrict__ ovf_pool,
                                                   const uint32_t* __restrict__ chunk_first,
                                                   const uint32_t* __restrict__ chunk_last, uint32_t n_chunks,
                                                   uint32_t n_items,
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
    uint32_t* dty = dsc + nb_max * DESC_WORDS;         // nb_max (rounded up to 4)
    uint32_t* seqst = dty + ((nb_max + 3u) & ~3u);     // nb_max * seq_w + 8
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t item = blockIdx.x;
    if (item >= n_items) return;
    const uint32_t cbase = item * NW;
    const uint32_t c = cbase + (uint32_t)wave;
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
    const bool is_ins = valid && (info & SI_INSERT);
    const uint32_t jju = is_ins ? (s - soff[g]) - 1u : 0u;   // insertion column of an insertion slot
    const uint32_t dsym = info & 0xf;
    const bool first = (info & SI_FIRST) != 0;
    uint32_t d1 = wave_shr1(dsym), d2 = wave_shr1(d1);
    const uint32_t f1 = wave_shr1((uint32_t)first);
    const uint32_t prev_dsym = d1;
    if (first) { d1 = 0; d2 = 0; }
    else if (f1) d2 = 0;
    VoteLane<E> vl;
    vl.init(d2 << 8 | d1 << 4 | dsym);
    const uint32_t kstart1 = d1 << 4 | dsym;            // context of the second slot of a run that repeats the draft: (0, d', d)
    const unsigned long long F1 = __ballot(first), F2 = __ballot(first || f1 != 0);
    const uint32_t F1lo = (uint32_t)F1, F1hi = (uint32_t)(F1 >> 32), F2lo = (uint32_t)F2, F2hi = (uint32_t)(F2 >> 32);
    uint32_t basemask = 0;              // symbols that voted on my slot in dirty pairs (plain pairs only ever vote the draft's)
    const uint32_t sv = valid ? s : 0xffffffffu;   // slot for coverage tests (never covered when invalid)
    const int32_t cs = (int32_t)((int64_t)c * VOTE_CH - 2);   // slot of lane 0 (may be -2 for chunk 0)
    uint32_t wf = 0xffffffffu, wl = 0;
    if (chunk_ok) { wf = chunk_first[c]; wl = chunk_last[c]; }
    wf = (uint32_t)__builtin_amdgcn_readfirstlane((int)wf);
    wl = (uint32_t)__builtin_amdgcn_readfirstlane((int)wl);
    __syncthreads();
    const uint32_t r0 = sh_r[0], r1 = sh_r[1];
    if (r0 != 0xffffffffu) {
        for (uint64_t rb = r0; rb <= r1; rb += nb_max) {
            const uint32_t nb = (uint32_t)((r1 - rb + 1 < nb_max) ? (r1 - rb + 1) : nb_max);
            // ---- stage descriptors, dirty hulls and packed bases of the batch (all contiguous in HBM): coalesced
            {
                const uint4* dsrc = reinterpret_cast<const uint4*>(desc + rb * DESC_WORDS);
                uint4* ddst = reinterpret_cast<uint4*>(dsc);
                for (uint32_t i = tid; i < nb * (DESC_WORDS / 4); i += NW * 64) ddst[i] = dsrc[i];
                for (uint32_t i = tid; i < nb; i += NW * 64) dty[i] = dirty[rb + i];
            }
            const uint64_t sq0 = R.seq_off[rb] & ~15ull;
            const uint64_t sq1 = R.seq_off[rb + nb - 1] + (((uint64_t)R.l_qseq[rb + nb - 1] + 1) >> 1);
            const uint32_t sq_quads = (uint32_t)((sq1 - sq0 + 15) >> 4);
            const uint32_t sq_fit = sq_quads * 4 <= nb_max * seq_w + 8 ? sq_quads : (nb_max * seq_w + 8) / 4;
            if (sq_fit != sq_quads && tid == 0) atomicOr(&counters[CNT_ERR], ERR_BAD_RECORD);
            {
                const uint4* src = reinterpret_cast<const uint4*>(R.seq + sq0);
                uint4* dst = reinterpret_cast<uint4*>(seqst);
                for (uint32_t i = tid; i < sq_fit; i += NW * 64) dst[i] = src[i];
            }
            const uint32_t sq0_lo = (uint32_t)sq0;   // descriptors carry the low word of their record's pool offset
            __syncthreads();
            if (wf != 0xffffffffu && !(ablate & 16u)) {
                const uint8_t* seqb = reinterpret_cast<const uint8_t*>(seqst);
                const uint64_t ia = wf > rb ? wf : rb, ib = (uint64_t)wl < rb + nb - 1 ? (uint64_t)wl : rb + nb - 1;
                const int32_t lb = ia <= ib ? (int32_t)(ib - rb) : -1;
                for (int32_t gb = lb >= 0 ? (int32_t)(ia - rb) : 0; gb <= lb; gb += 64) {
                    // ---- phase 1: my record of this group of 64
                    const int32_t li = gb + lane;
                    const bool act = li <= lb;
                    uint4 hv = make_uint4(1u, 0u, 0u, 0u);
                    uint2 sg = make_uint2(0u, 0u);
                    uint32_t dw = DIRTY_NONE, slast = 0;
                    if (act) {
                        hv = *reinterpret_cast<const uint4*>(dsc + (uint32_t)li * DESC_WORDS);
                        sg = *reinterpret_cast<const uint2*>(dsc + (uint32_t)li * DESC_WORDS + DESC_SEG0);
                        dw = dty[li];
                        slast = (hv.z & DESC_CHAIN) ? dsc[(uint32_t)li * DESC_WORDS + DESC_NEXT + 1] : hv.y;
                    }
                    int32_t lo = (int32_t)hv.x - cs, hi = (int32_t)slast - cs;
                    lo = lo < 0 ? 0 : lo;
                    hi = hi > 63 ? 63 : hi;
                    const bool live = act && (int32_t)(slast - hv.x) >= 0 && lo <= hi;
                    unsigned long long Cm = live ? (~0ull >> (63 - hi)) & (~0ull << lo) : 0ull;
                    // dirty = the record disagrees with the draft somewhere in reach of this chunk (k_desc's chunk mask, np1_desc.h)
                    const uint32_t cj = c - hv.x / VOTE_CH;
                    const bool isdirty = (hv.z & DESC_CHAIN) != 0 || ((dw >> (cj < 31u ? cj : 31u)) & 1u) != 0;
                    const uint32_t Clo = (uint32_t)Cm, Chi = (uint32_t)(Cm >> 32);
                    // lanes whose whole context is the draft's, if the record repeats the draft on all of C
                    const unsigned long long C1 = Cm << 1, C2 = Cm << 2;
                    const uint32_t M0lo = Clo & ((uint32_t)C1 | F1lo) & ((uint32_t)C2 | F2lo) & ~3u;
                    const uint32_t M0hi = Chi & ((uint32_t)(C1 >> 32) | F1hi) & ((uint32_t)(C2 >> 32) | F2hi);
                    const bool starts = ((Clo & ~3u) & ~M0lo) != 0u || (Chi & ~M0hi) != 0u;   // covered voting lanes without the full context
                    unsigned long long todo = __ballot(live);
                    const unsigned long long DIRTY = __ballot(live && isdirty), START = __ballot(live && !isdirty && starts);
                    if (ablate & 4u) todo = 0ull;
                    // ---- phase 2: the records of the group in file order
                    while (todo != 0ull) {
                        const int r = __builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        if (!((DIRTY >> r) & 1ull)) {
                            // plain pair: count(draft's context) += 1 on the lanes with the full context
                            const unsigned long long M0 = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)M0lo, r) |
                                                          (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)M0hi, r) << 32;
                            add_lane_mask(vl.c0, M0);
                            if ((START >> r) & 1ull) {
                                // its run starts inside the chunk: the first two voting lanes see (0, 0, d) and (0, d', d)
                                const unsigned long long C = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)Clo, r) |
                                                             (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)Chi, r) << 32;
                                const unsigned long long Rm = (C & ~3ull) & ~M0;
                                const uint32_t k = ((C << 1) >> lane) & 1ull ? kstart1 : dsym;   // lane-1 covered: second slot of the run
                                const unsigned long long M1 = __ballot(k == vl.k1) & Rm;
                                add_lane_mask(vl.c1, M1);
                                const unsigned long long rest = Rm & ~M1;
                                if (rest != 0ull && !(ablate & 1u)) {
                                    if ((rest >> lane) & 1ull) vl.tally(k, L, lane);
                                }
                            }
                            continue;
                        }
                        if (ablate & 8u) continue;
                        // ---- dirty pair: every lane evaluates the record's symbol at its slot and its context (the body of k_tile3;
                        //      the descriptor comes back from LDS as broadcast reads)
                        const uint32_t* d = dsc + (uint32_t)(gb + r) * DESC_WORDS;
                        const uint4 h = *reinterpret_cast<const uint4*>(d);
                        if (!(h.z & DESC_CHAIN)) {
                            const uint32_t nseg = h.z & 0xffu, nins = (h.z >> 8) & 0xffu;
                            const bool cov = sv >= h.x && sv <= h.y;
                            uint32_t q = 0;
                            bool isdel = true;   // covered insertion column the record merely passes (or pads): DEL
                            for (uint32_t k2 = 0; k2 < nseg; ++k2) {
                                const uint2 sk = *reinterpret_cast<const uint2*>(d + DESC_SEG0 + 2 * k2);
                                const uint32_t off = g - sk.x;
                                const bool in = !is_ins && off < (sk.y & 0xffffu);
                                isdel = in ? (sk.y >> 16) == 0xffffu : isdel;
                                q = in ? (sk.y >> 16) + off : q;
                            }
                            for (uint32_t k2 = 0; k2 < nins; ++k2) {
                                const uint2 ik = *reinterpret_cast<const uint2*>(d + DESC_INS0 + 2 * k2);
                                const bool in = is_ins && ik.x == g && jju < (ik.y & 0xffffu);
                                isdel = in ? false : isdel;
                                q = in ? (ik.y >> 16) + jju : q;
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
                            if (__ballot(rest) != 0ull && !(ablate & 1u)) {
                                if (rest) vl.tally(k, L, lane);   // a context seen for the first time, or one kept in the LDS list
                            }
                        } else if (!(ablate & 2u)) {
                            // indel operations that fill more than one descriptor: the general walk over the parts (they live in HBM)
                            const SeqLds sq{seqb + (h.w - sq0_lo)};
                            const int32_t jj = is_ins ? (int32_t)jju : -1;
                            uint32_t rsym = 0, nv = 0;
                            vote_part<E>(d, sq, valid, s, g, jj, lane, rsym, basemask, vl, L, nv);
                            uint32_t nx = d[DESC_NEXT];
                            while (nx) {
                                const uint32_t* dg = ovf_pool + (uint64_t)(nx - 1) * DESC_WORDS;
                                vote_part<E>(dg, sq, valid, s, g, jj, lane, rsym, basemask, vl, L, nv);
                                nx = dg[DESC_NEXT];
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    const bool ovf_any = chunk_ok && __ballot(vl.ovf) != 0ull;
    // a slot is single-state when nothing but the draft's own symbol voted on it (plain pairs never vote anything else)
    const bool single = (basemask & ~(1u << dsym)) == 0u;
    const uint32_t psingle = wave_shr1((uint32_t)single);
    const bool prev_is_single = first || psingle != 0;
    // vote statistic: every tally of an own slot but the draft's own one
    uint32_t nvotes = 0;
    if (valid && lane >= 2) {
        nvotes = vl.c0 + vl.c1 - 1u;
        for (uint32_t e = 2; e < vl.n; ++e) nvotes += L[(e - 2) * 64 + lane] & 0xffffu;
    }
    for (int o = 32; o > 0; o >>= 1) nvotes += __shfl_down(nvotes, o);
    tile_epilogue<E, NW>(vl, L, tid, c, chunk_ok && !ovf_any, ovf_any, valid, s, info, dsym, first, prev_dsym, single, prev_is_single,
                         vl.total(L, lane), slot_res, slot_rec, pool, pool_cap, counters, heads, heads_cap, redo_out, redo_ci,
                         flag_single & 0xffu, nvotes, votes, (flag_single & FLAG_ALL_RECORDS) != 0);
}

// ------------------------------------------------------------------------------------------------
// k_tile5 (default): event form of the pileup vote (np1_events.h).  Same tile geometry, staging and epilogue as
// k_tile3, but the per-(record, slot) vote loop is gone:
//   phase R  one lane per candidate record: +1/-1 into the tile's coverage difference array, and the record's
//            EVENTS (votes whose 3-base context differs from the draft's) found by XOR-ing packed bases eight at a
//            time against the draft's packed symbols, exact per-slot evaluation only around disagreements
//   phase S  block scan -> coverage per slot; counting sort of the events by slot; one lane per slot orders its few
//            events by record (= first-seen order) and tallies them; count(k0) = 1 + coverage - #events
// Instruction count per record drops from ~475 wave-instructions (k_tile3) to a few dozen.
template <int NW>
struct Tile5Lds {
    static constexpr uint32_t NWIN = NW * VOTE_CH + 2;
};

struct Tile5Sink {
    uint32_t* ev_total;
    uint32_t* ev_unsorted;
    uint32_t* evcnt;
    uint32_t ev_max, w0, rec_local;
    uint32_t* overflow;
    __device__ __forceinline__ void event(uint32_t slot, uint32_t ctx) {
        const uint32_t idx = atomicAdd(ev_total, 1u);
        if (idx < ev_max) {
            ev_unsorted[idx] = (slot - w0) << 23 | rec_local << 12 | (ctx & 0xfffu);
            atomicAdd(&evcnt[slot - w0], 1u);
        } else {
            *overflow = 1u;
        }
    }
};

template <int E, int NW>
__global__ __launch_bounds__(NW * 64) void k_tile5(ReadsDev R, const uint32_t* __restrict__ soff,
                                                   const uint32_t* __restrict__ desc,
                                                   const uint32_t* __restrict__ ovf_pool,
                                                   const uint32_t* __restrict__ chunk_first,
                                                   const uint32_t* __restrict__ chunk_last, uint32_t n_chunks,
                                                   const uint8_t* __restrict__ slot_info,
                                                   const uint32_t* __restrict__ slot_g, uint32_t S, uint32_t seq_w,
                                                   uint32_t nb_max, uint32_t ev_max, uint16_t* __restrict__ slot_res,
                                                   uint32_t* __restrict__ slot_rec, uint32_t* __restrict__ pool,
                                                   uint32_t pool_cap, uint32_t* __restrict__ counters,
                                                   uint32_t* __restrict__ heads, uint32_t heads_cap,
                                                   uint32_t* __restrict__ redo_out, uint32_t redo_ci, uint32_t flag_single,
                                                   unsigned long long* __restrict__ votes, uint32_t ablate) {
    constexpr uint32_t NWIN = NW * VOTE_CH + 2;
    constexpr uint32_t T = NW * 64;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ __attribute__((aligned(16))) uint32_t sh_r[8];   // r0, r1, ev_total, overflow
    uint32_t* lists = lds;                                   // NW * (E-2) * 64
    uint32_t* dsc = lists + NW * (E - 2) * 64;               // (nb_max + 1) * DESC_WORDS
    uint32_t* seqst = dsc + (nb_max + 1) * DESC_WORDS;       // nb_max * seq_w + 8
    uint32_t* win_sg = seqst + nb_max * seq_w + 8;           // NWIN
    uint32_t* cover = win_sg + NWIN;                         // NWIN + 1 : difference array, then coverage
    uint32_t* evcnt = cover + NWIN + 1;                      // NWIN
    uint32_t* evoff = evcnt + NWIN;                          // NWIN + 1
    uint32_t* evcur = evoff + NWIN + 1;                      // NWIN
    uint32_t* bmask = evcur + NWIN;                          // NWIN : base mask per slot
    uint32_t* dpk_w = bmask + NWIN;                          // NWIN / 8 + 4 words : packed draft symbols
    uint16_t* win_k0 = reinterpret_cast<uint16_t*>(dpk_w + NWIN / 8 + 4);   // NWIN (+pad)
    uint8_t* win_sinfo = reinterpret_cast<uint8_t*>(win_k0 + NWIN + 2);     // NWIN (+pad)
    uint32_t* ev_unsorted = reinterpret_cast<uint32_t*>(win_sinfo + ((NWIN + 7) & ~3u));   // ev_max
    uint32_t* ev_sorted = ev_unsorted + ev_max;              // ev_max
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t cbase = blockIdx.x * NW;
    if (cbase >= n_chunks || (uint64_t)cbase * VOTE_CH >= S) return;   // (the chunk count is rounded up by one: nothing to own here)
    const uint32_t c = cbase + wave;
    const bool chunk_ok = c < n_chunks;
    const uint32_t T0 = cbase * VOTE_CH;
    const uint32_t w0 = T0 >= 2 ? T0 - 2 : 0u;
    const uint32_t T1 = (uint64_t)T0 + NW * VOTE_CH < S ? T0 + NW * VOTE_CH : S;   // exclusive
    const uint32_t wn = T1 - w0;
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
        sh_r[2] = 0;
        sh_r[3] = 0;
    }
    // ---- window arrays
    for (uint32_t k = tid; k < NWIN + 1; k += T) {
        cover[k] = 0;
        if (k < NWIN) { evcnt[k] = 0; evcur[k] = 0; }
    }
    for (uint32_t k = tid; k < NWIN / 8 + 4; k += T) dpk_w[k] = 0;
    for (uint32_t k = tid; k < wn; k += T) {
        win_sinfo[k] = slot_info[w0 + k];
        win_sg[k] = slot_g[w0 + k];
    }
    __syncthreads();
    const uint32_t dpk_g0 = win_sg[0] & ~1u;
    for (uint32_t k = tid; k < wn; k += T) {
        const uint32_t info = win_sinfo[k];
        uint32_t d1 = 0, d2 = 0;
        if (!(info & SI_FIRST) && k >= 1) {
            d1 = win_sinfo[k - 1] & 0xfu;
            if (!(win_sinfo[k - 1] & SI_FIRST) && k >= 2) d2 = win_sinfo[k - 2] & 0xfu;
        }
        win_k0[k] = (uint16_t)(d2 << 8 | d1 << 4 | (info & 0xfu));   // valid for every owned slot (k >= 2, or contig starts)
        if (!(info & SI_INSERT)) {
            const uint32_t i = win_sg[k] - dpk_g0;   // nibble index, BAM packing: even index = high nibble
            atomicOr(&dpk_w[i >> 3], (info & 0xfu) << (((i >> 1) & 3u) * 8u + ((~i & 1u) << 2)));
        }
    }
    __syncthreads();
    const uint32_t r0 = sh_r[0], r1 = sh_r[1];
    EvWindow win{w0, wn, T0 >= 2 ? T0 : 0u, win_sinfo, win_sg, win_k0, reinterpret_cast<const uint8_t*>(dpk_w), dpk_g0, soff};
    bool bad_tile = false;   // more candidates than the event words can index: fall back to k_tile3 for this tile
    if (r0 != 0xffffffffu) {
        if ((uint64_t)r1 - r0 + 1 > 2048) bad_tile = true;
        for (uint64_t rb = r0; rb <= r1 && !bad_tile; rb += nb_max) {
            const uint32_t nb = (uint32_t)((r1 - rb + 1 < nb_max) ? (r1 - rb + 1) : nb_max);
            {
                const uint4* dsrc = reinterpret_cast<const uint4*>(desc + rb * DESC_WORDS);
                uint4* ddst = reinterpret_cast<uint4*>(dsc);
                for (uint32_t i = tid; i < nb * (DESC_WORDS / 4); i += T) ddst[i] = dsrc[i];
            }
            const uint64_t sq0 = R.seq_off[rb] & ~15ull;
            const uint64_t sq1 = R.seq_off[rb + nb - 1] + (((uint64_t)R.l_qseq[rb + nb - 1] + 1) >> 1);
            const uint32_t sq_quads = (uint32_t)((sq1 - sq0 + 15) >> 4) + 1;   // +1: nib8_be may peek 4 bytes past a record
            const uint32_t sq_fit = sq_quads * 4 <= nb_max * seq_w + 8 ? sq_quads : (nb_max * seq_w + 8) / 4;
            if (sq_fit != sq_quads && tid == 0) atomicOr(&counters[CNT_ERR], ERR_BAD_RECORD);
            {
                const uint4* src = reinterpret_cast<const uint4*>(R.seq + sq0);
                uint4* dst = reinterpret_cast<uint4*>(seqst);
                for (uint32_t i = tid; i < sq_fit; i += T) dst[i] = src[i];
            }
            const uint32_t sq0_lo = (uint32_t)sq0;
            __syncthreads();
            // ---- phase R: one lane per record
            for (uint32_t k = tid; k < nb; k += T) {
                const uint32_t* d = dsc + k * DESC_WORDS;
                const uint32_t sf = d[0], sl = d[DESC_NEXT + 1];
                if (sf <= sl && d[1] >= sf && sl >= w0 && sf < T1) {
                    const uint32_t lo = sf > w0 ? sf : w0, hi = sl < T1 - 1 ? sl : T1 - 1;
                    atomicAdd(&cover[lo - w0], 1u);
                    atomicAdd(&cover[hi - w0 + 1], 0xffffffffu);
                    Tile5Sink sink{&sh_r[2], ev_unsorted, evcnt, ev_max, w0, (uint32_t)(rb - r0) + k, &sh_r[3]};
                    const uint8_t* sqb = reinterpret_cast<const uint8_t*>(seqst) + (d[3] - sq0_lo);
                    if (ablate & 2u) continue;
                    if (d[2] & DESC_CHAIN) { if (!(ablate & 1u)) record_events<true>(d, ovf_pool, sqb, win, sink); }   // rare: parts live in HBM
                    else record_events<false>(d, nullptr, sqb, win, sink);
                }
            }
            __syncthreads();
        }
    }
    if (sh_r[3]) bad_tile = true;
    if (tid == 0) { atomicAdd(&counters[CNT_STAT_EVENTS], sh_r[2]); if (bad_tile) atomicAdd(&counters[CNT_STAT_FALLBACK], 1u); }   // statistics: events, fallback tiles
    // ---- phase S0: coverage = inclusive scan of the difference array; event offsets = exclusive scan of the counts
    {
        // NWIN <= T (498 <= 512): one element per lane, Hillis-Steele in LDS
        uint32_t cv = tid < NWIN ? cover[tid] : 0u, ec = tid < NWIN ? evcnt[tid] : 0u;
        uint32_t* sc_a = ev_sorted;          // scratch (the sorted buffer is not in use yet): 2 * T words needed
        uint32_t* sc_b = ev_sorted + T;
        sc_a[tid] = cv;
        sc_b[tid] = ec;
        __syncthreads();
        for (uint32_t o = 1; o < T; o <<= 1) {
            const uint32_t a = tid >= o ? sc_a[tid - o] : 0u, b = tid >= o ? sc_b[tid - o] : 0u;
            __syncthreads();
            sc_a[tid] += a;
            sc_b[tid] += b;
            __syncthreads();
        }
        if (tid < NWIN) { cover[tid] = sc_a[tid]; evoff[tid] = sc_b[tid] - ec; }
        __syncthreads();
    }
    const uint32_t ev_total = sh_r[2] < ev_max ? sh_r[2] : ev_max;
    for (uint32_t i = tid; i < ev_total; i += T) {   // counting sort by slot (order inside a slot fixed below)
        const uint32_t e = ev_unsorted[i];
        const uint32_t k = e >> 23;
        ev_sorted[evoff[k] + atomicAdd(&evcur[k], 1u)] = e & 0x7fffffu;
    }
    __syncthreads();
    // ---- phase S1: one lane per slot
    uint32_t* L = lists + wave * (E - 2) * 64;
    const int64_t s64 = (int64_t)c * VOTE_CH - 2 + lane;
    const bool valid = chunk_ok && s64 >= 0 && s64 < (int64_t)S;
    const uint32_t s = (uint32_t)s64;
    const uint32_t k = valid ? s - w0 : 0u;
    const uint32_t info = valid ? win_sinfo[k] : 0u;
    const uint32_t dsym = info & 0xfu;
    const bool first = (info & SI_FIRST) != 0;
    const uint32_t prev_dsym = (valid && k >= 1) ? (win_sinfo[k - 1] & 0xfu) : 0u;
    VoteLane<E> vl;
    vl.init(valid ? win_k0[k] : 0u);
    uint32_t basemask = 1u << dsym, total = 0, nvotes = 0;
    if (valid && !bad_tile && !(ablate & 4u)) {
        const uint32_t ne = evcnt[k], eb = evoff[k];
        // order this slot's events by record (insertion sort; a record votes a slot at most once), then tally
        if (lane >= 2) {
            for (uint32_t i = 1; i < ne; ++i) {
                const uint32_t key = ev_sorted[eb + i];
                uint32_t j = i;
                while (j > 0 && ev_sorted[eb + j - 1] > key) { ev_sorted[eb + j] = ev_sorted[eb + j - 1]; --j; }
                ev_sorted[eb + j] = key;
            }
            for (uint32_t i = 0; i < ne; ++i) {
                const uint32_t ctx = ev_sorted[eb + i] & 0xfffu;
                basemask |= 1u << (ctx & 0xfu);
                vl.tally(ctx, L, lane);
            }
            const uint32_t cv = cover[k];
            vl.c0 += cv - ne;   // every other covering vote carries the draft's own context
            total = (1u + cv) & 0xffffu;
            nvotes = cv;
            bmask[k] = basemask;
        } else if (wave == 0) {   // the tile's two left-context slots: only their base mask matters
            for (uint32_t i = 0; i < ne; ++i) basemask |= 1u << (ev_sorted[eb + i] & 0xfu);
            bmask[k] = basemask;
        }
    }
    __syncthreads();
    const bool single = __popc(basemask) == 1;
    const bool prev_is_single = first || (valid && k >= 1 && __popc(bmask[k - 1]) == 1) || !valid || k == 0;
    for (int o = 32; o > 0; o >>= 1) nvotes += __shfl_down(nvotes, o);
    const bool ovf_any = chunk_ok && (bad_tile || __ballot(vl.ovf) != 0ull);
    tile_epilogue<E, NW>(vl, L, tid, c, chunk_ok && !ovf_any, ovf_any, valid, s, info, dsym, first, prev_dsym, single, prev_is_single,
                         total, slot_res, slot_rec, pool, pool_cap, counters, heads, heads_cap, redo_out, redo_ci, flag_single, nvotes, votes);
}

// ------------------------------------------------------------------------------------------------
// k_tile6: the event form with independent work items (np1_events.h, group form).  k_tile5 walks each record
// sequentially in one lane, which leaves the workgroup waiting on LDS latency chains; here the unit of work is one
// (record, 8-slot group) pair, ~3000 per tile, spread over all 512 lanes:
//   phase C  clean test per item: ten packed bases of the record XOR ten packed draft symbols (three aligned LDS
//            words each); a clean item contributes nothing but the record's two start events
//   phase X  dirty items (mismatch, indel, insertion column, chained record: ~15 %) go through an LDS queue and are
//            evaluated exactly, ten lanes per item, contexts from the neighbouring lanes (DPP)
//   phase S  coverage scan, counting sort of the events by slot, rank by record inside a slot (= first-seen order),
//            one lane per slot tallies its few events; count(k0) = 1 + coverage - #events
struct RegSink2 {   // the (at most two) start events of a clean item, kept in registers
    uint32_t n, w0, rec_local, e0, e1;
    __device__ __forceinline__ void event(uint32_t slot, uint32_t ctx) {
        const uint32_t word = (slot - w0) << 23 | rec_local << 12 | (ctx & 0xfffu);
        if (n == 0) e0 = word; else e1 = word;
        ++n;
    }
};

// wave-aggregated append of one event per flagged lane (one LDS atomic per wave instead of one per event)
__device__ __forceinline__ void ev_push(bool has, uint32_t word, uint32_t* ev_total, uint32_t* ev_buf, uint32_t* evcnt,
                                        uint32_t ev_max, uint32_t* overflow, int lane) {
    const unsigned long long m = __ballot(has);
    if (m) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(ev_total, (uint32_t)__popcll(m));
        base = __shfl(base, 0);
        if (has) {
            const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (idx < ev_max) {
                ev_buf[idx] = word;
                atomicAdd(&evcnt[word >> 23], 1u);
            } else {
                *overflow = 1u;
            }
        }
    }
}

template <int E, int NW>
__global__ __launch_bounds__(NW * 64) void k_tile6(ReadsDev R, const uint32_t* __restrict__ soff,
                                                   const uint32_t* __restrict__ desc,
                                                   const uint32_t* __restrict__ ovf_pool,
                                                   const uint32_t* __restrict__ chunk_first,
                                                   const uint32_t* __restrict__ chunk_last, uint32_t n_chunks,
                                                   const uint8_t* __restrict__ slot_info,
                                                   const uint32_t* __restrict__ slot_g, uint32_t S, uint32_t seq_w,
                                                   uint32_t nb_max, uint32_t ev_max, uint32_t it_max,
                                                   uint16_t* __restrict__ slot_res, uint32_t* __restrict__ slot_rec,
                                                   uint32_t* __restrict__ pool, uint32_t pool_cap,
                                                   uint32_t* __restrict__ counters, uint32_t* __restrict__ heads,
                                                   uint32_t heads_cap, uint32_t* __restrict__ redo_out, uint32_t redo_ci,
                                                   uint32_t flag_single, unsigned long long* __restrict__ votes,
                                                   uint32_t ablate, unsigned long long* __restrict__ dbg) {
    constexpr uint32_t NWIN = NW * VOTE_CH + 2;
    constexpr uint32_t NGRP = (NWIN + EV_G - 1) / EV_G;
    constexpr uint32_t T = NW * 64;
    static_assert(NWIN <= T && NGRP <= 64, "one lane per window slot; group index in 6 bits");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ __attribute__((aligned(16))) uint32_t sh_r[8];   // r0, r1, ev_total, overflow, dirty count
    __shared__ uint32_t sh_w[2 * NW];
    uint32_t* lists = lds;                                   // NW * (E-2) * 64
    uint32_t* dsc = lists + NW * (E - 2) * 64;               // (nb_max + 1) * DESC_WORDS
    uint32_t* seqst = dsc + (nb_max + 1) * DESC_WORDS;       // nb_max * seq_w + 8
    uint32_t* win_sg = seqst + nb_max * seq_w + 8;           // NWIN
    uint32_t* cover = win_sg + NWIN;                         // NWIN + 1 : difference array, then coverage
    uint32_t* evcnt = cover + NWIN + 1;                      // NWIN
    uint32_t* evoff = evcnt + NWIN;                          // NWIN + 1
    uint32_t* bmask = evoff + NWIN + 1;                      // NWIN : base mask per slot
    uint32_t* dpk_w = bmask + NWIN;                          // NWIN / 8 + 4 words : packed draft symbols
    uint16_t* win_k0 = reinterpret_cast<uint16_t*>(dpk_w + NWIN / 8 + 4);   // NWIN (+pad)
    uint8_t* win_sinfo = reinterpret_cast<uint8_t*>(win_k0 + NWIN + 2);     // NWIN (+pad)
    uint16_t* gins = reinterpret_cast<uint16_t*>(win_sinfo + ((NWIN + 7) & ~3u));   // 64
    uint16_t* dirtyq = gins + 64;                            // it_max
    uint32_t* ev_a = reinterpret_cast<uint32_t*>(dirtyq + it_max);   // ev_max : events as produced, later ordered
    uint32_t* ev_b = ev_a + ev_max;                          // ev_max : item list (phase C/X), then events by slot
    uint16_t* items = reinterpret_cast<uint16_t*>(ev_b);     // it_max <= 2 * ev_max
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    long long tm[7] = {0, 0, 0, 0, 0, 0, 0};   // phase timestamps (diagnostics, dbg != nullptr)
    long long tc = 0, tx = 0;
    if (dbg) tm[0] = clock64();
    const uint32_t cbase = blockIdx.x * NW;
    if (cbase >= n_chunks || (uint64_t)cbase * VOTE_CH >= S) return;   // (the chunk count is rounded up by one: nothing to own here)
    const uint32_t c = cbase + wave;
    const bool chunk_ok = c < n_chunks;
    const uint32_t T0 = cbase * VOTE_CH;
    const uint32_t w0 = T0 >= 2 ? T0 - 2 : 0u;
    const uint32_t T1 = (uint64_t)T0 + NW * VOTE_CH < S ? T0 + NW * VOTE_CH : S;   // exclusive
    const uint32_t wn = T1 - w0;
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
        sh_r[2] = 0;
        sh_r[3] = 0;
    }
    // ---- window arrays
    for (uint32_t k = tid; k < NWIN + 1; k += T) {
        cover[k] = 0;
        if (k < NWIN) evcnt[k] = 0;
    }
    for (uint32_t k = tid; k < NWIN / 8 + 4; k += T) dpk_w[k] = 0;
    for (uint32_t k = tid; k < wn; k += T) {
        win_sinfo[k] = slot_info[w0 + k];
        win_sg[k] = slot_g[w0 + k];
    }
    __syncthreads();
    const uint32_t dpk_g0 = win_sg[0] & ~1u;
    for (uint32_t k = tid; k < wn; k += T) {
        const uint32_t info = win_sinfo[k];
        uint32_t d1 = 0, d2 = 0;
        if (!(info & SI_FIRST) && k >= 1) {
            d1 = win_sinfo[k - 1] & 0xfu;
            if (!(win_sinfo[k - 1] & SI_FIRST) && k >= 2) d2 = win_sinfo[k - 2] & 0xfu;
        }
        win_k0[k] = (uint16_t)(d2 << 8 | d1 << 4 | (info & 0xfu));   // valid for every owned slot (k >= 2, or contig starts)
        if (!(info & SI_INSERT)) {
            const uint32_t i = win_sg[k] - dpk_g0;   // nibble index, BAM packing: even index = high nibble
            atomicOr(&dpk_w[i >> 3], (info & 0xfu) << (((i >> 1) & 3u) * 8u + ((~i & 1u) << 2)));
        }
    }
    if (tid < 64) gins[tid] = (uint16_t)group_ins_mask(win_sinfo, wn, (uint32_t)tid);
    __syncthreads();
    if (dbg) tm[1] = clock64();
    const uint32_t r0 = sh_r[0], r1 = sh_r[1];
    const EvWindow win{w0, wn, T0 >= 2 ? T0 : 0u, win_sinfo, win_sg, win_k0, reinterpret_cast<const uint8_t*>(dpk_w), dpk_g0, soff};
    const GroupWin gw{gins};
    bool bad_tile = false;   // more candidates than the event words can index: fall back to k_tile3 for this tile
    if (r0 != 0xffffffffu) {
        if ((uint64_t)r1 - r0 + 1 > 2048) bad_tile = true;
        for (uint64_t rb = r0; rb <= r1 && !bad_tile; rb += nb_max) {
            const uint32_t nb = (uint32_t)((r1 - rb + 1 < nb_max) ? (r1 - rb + 1) : nb_max);
            {
                const uint4* dsrc = reinterpret_cast<const uint4*>(desc + rb * DESC_WORDS);
                uint4* ddst = reinterpret_cast<uint4*>(dsc);
                for (uint32_t i = tid; i < nb * (DESC_WORDS / 4); i += T) ddst[i] = dsrc[i];
            }
            const uint64_t sq0 = R.seq_off[rb] & ~15ull;
            const uint64_t sq1 = R.seq_off[rb + nb - 1] + (((uint64_t)R.l_qseq[rb + nb - 1] + 1) >> 1);
            const uint32_t sq_quads = (uint32_t)((sq1 - sq0 + 15) >> 4) + 1;   // +1: the packed compare may peek past a record
            const uint32_t sq_fit = sq_quads * 4 <= nb_max * seq_w + 8 ? sq_quads : (nb_max * seq_w + 8) / 4;
            if (sq_fit != sq_quads && tid == 0) atomicOr(&counters[CNT_ERR], ERR_BAD_RECORD);
            {
                const uint4* src = reinterpret_cast<const uint4*>(R.seq + sq0);
                uint4* dst = reinterpret_cast<uint4*>(seqst);
                for (uint32_t i = tid; i < sq_fit; i += T) dst[i] = src[i];
            }
            const uint32_t sq0_lo = (uint32_t)sq0;
            __syncthreads();
            // ---- one lane per record: coverage difference array, number of groups the record touches
            uint32_t my_ng = 0, my_g0 = 0;
            if ((uint32_t)tid < nb) {
                uint32_t lo, hi;
                bool sb;
                if (record_window_range(dsc + tid * DESC_WORDS, win, &lo, &hi, &sb)) {
                    atomicAdd(&cover[lo], 1u);
                    atomicAdd(&cover[hi + 1], 0xffffffffu);
                    my_g0 = lo / EV_G;
                    my_ng = hi / EV_G - my_g0 + 1;
                }
            }
            uint32_t incl = my_ng;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) sh_w[wave] = incl;
            __syncthreads();
            uint32_t my_off = incl - my_ng, total = 0;
            for (int w = 0; w < NW; ++w) {
                const uint32_t v = sh_w[w];
                if (w < wave) my_off += v;
                total += v;
            }
            for (uint32_t round0 = 0; round0 < total; round0 += it_max) {
                for (uint32_t t = 0; t < my_ng; ++t) {
                    const uint32_t pos = my_off + t - round0;
                    if (pos < it_max) items[pos] = (uint16_t)((uint32_t)tid << 6 | (my_g0 + t));
                }
                if (tid == 0) sh_r[4] = 0;
                __syncthreads();
                const uint32_t cnt = total - round0 < it_max ? total - round0 : it_max;
                long long ta = 0;
                if (dbg) ta = clock64();
                // ---- phase C: clean test per item
                for (uint32_t base = 0; base < cnt; base += T) {
                    const uint32_t it = base + tid;
                    bool dirty = false;
                    uint32_t code = 0;
                    RegSink2 rs{0, w0, 0, 0, 0};
                    if (it < cnt && !(ablate & 2u)) {
                        code = items[it];
                        const uint32_t i = code >> 6, j = code & 63u;
                        const uint32_t* d = dsc + i * DESC_WORDS;
                        uint32_t lo, hi;
                        bool sb;
                        (void)record_window_range(d, win, &lo, &hi, &sb);
                        rs.rec_local = (uint32_t)(rb - r0) + i;
                        if (d[2] & DESC_CHAIN) dirty = true;
                        else {
                            const uint8_t* sqb = reinterpret_cast<const uint8_t*>(seqst) + (d[3] - sq0_lo);
                            dirty = !group_clean(d, sqb, win, gw, j, lo, hi, sb, rs);
                        }
                    }
                    const unsigned long long dm = __ballot(dirty);
                    if (dm) {
                        uint32_t qb = 0;
                        if (lane == 0) qb = atomicAdd(&sh_r[4], (uint32_t)__popcll(dm));
                        qb = __shfl(qb, 0);
                        if (dirty) dirtyq[qb + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = (uint16_t)code;
                    }
                    ev_push(rs.n >= 1, rs.e0, &sh_r[2], ev_a, evcnt, ev_max, &sh_r[3], lane);
                    ev_push(rs.n >= 2, rs.e1, &sh_r[2], ev_a, evcnt, ev_max, &sh_r[3], lane);
                }
                __syncthreads();
                if (dbg) { const long long tb = clock64(); tc += tb - ta; ta = tb; }
                // ---- phase X: dirty items, ten lanes each (two context slots + the group)
                const uint32_t nd = (ablate & 1u) ? 0u : sh_r[4];
                const uint32_t e6 = (uint32_t)lane / EV_GL, t6 = (uint32_t)lane - e6 * EV_GL;
                for (uint32_t base = 0; base < nd; base += NW * 6) {
                    const uint32_t idx = base + (uint32_t)wave * 6 + e6;
                    const bool active = lane < 60 && idx < nd;
                    uint32_t sym = 0, k = 0, rec_local = 0;
                    bool cov = false;
                    if (active) {
                        const uint32_t code = dirtyq[idx];
                        const uint32_t i = code >> 6, j = code & 63u;
                        const uint32_t* d = dsc + i * DESC_WORDS;
                        uint32_t lo, hi;
                        bool sb;
                        (void)record_window_range(d, win, &lo, &hi, &sb);
                        rec_local = (uint32_t)(rb - r0) + i;
                        const int32_t kk = (int32_t)(EV_G * j) - 2 + (int32_t)t6;
                        cov = kk >= (int32_t)lo && kk <= (int32_t)hi;
                        if (cov) {
                            k = (uint32_t)kk;
                            const uint32_t s = w0 + k, info = win_sinfo[k], g = win_sg[k];
                            const int32_t jj = (info & SI_INSERT) ? (int32_t)(s - soff[g]) - 1 : -1;
                            const SeqLds sq{reinterpret_cast<const uint8_t*>(seqst) + (d[3] - sq0_lo)};
                            if (s <= d[1]) {
                                sym = desc_symbol(d, g, jj, sq);   // head part: LDS
                            } else {                               // rare: overflow parts of a chained record live in HBM
                                uint32_t nx = d[DESC_NEXT];
                                while (nx) {
                                    const uint32_t* part = ovf_pool + (uint64_t)(nx - 1) * DESC_WORDS;
                                    if (s <= part[1]) {
                                        if (s >= part[0]) sym = desc_symbol(part, g, jj, sq);
                                        break;
                                    }
                                    nx = part[DESC_NEXT];
                                }
                            }
                        }
                    }
                    const uint32_t p1 = wave_shr1(sym), p2 = wave_shr1(p1);
                    const uint32_t ctx = p2 << 8 | p1 << 4 | sym;
                    bool emit = false;
                    uint32_t val = ctx;
                    if (cov && t6 >= 2) {
                        if (w0 + k >= win.own0) emit = ctx != win_k0[k];
                        else { emit = sym != (uint32_t)(win_sinfo[k] & 0xfu); val = sym; }
                    }
                    ev_push(emit, k << 23 | rec_local << 12 | (val & 0xfffu), &sh_r[2], ev_a, evcnt, ev_max, &sh_r[3], lane);
                }
                __syncthreads();
                if (dbg) tx += clock64() - ta;
            }
        }
    }
    if (dbg) tm[2] = clock64();
    if (sh_r[3]) bad_tile = true;
    if (tid == 0) { atomicAdd(&counters[CNT_STAT_EVENTS], sh_r[2]); if (bad_tile) atomicAdd(&counters[CNT_STAT_FALLBACK], 1u); }
    // ---- phase S0: coverage = inclusive scan of the difference array; event offsets = exclusive scan of the counts
    const uint32_t ev_total = sh_r[2] < ev_max ? sh_r[2] : ev_max;
    {
        const uint32_t cv = (uint32_t)tid < NWIN ? cover[tid] : 0u, ec = (uint32_t)tid < NWIN ? evcnt[tid] : 0u;
        uint32_t ci = cv, ei = ec;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(ci, o), b = __shfl_up(ei, o);
            if (lane >= o) { ci += a; ei += b; }
        }
        __syncthreads();   // (sh_w is free again)
        if (lane == 63) { sh_w[wave] = ci; sh_w[NW + wave] = ei; }
        __syncthreads();
        for (int w = 0; w < wave; ++w) { ci += sh_w[w]; ei += sh_w[NW + w]; }
        if ((uint32_t)tid < NWIN) { cover[tid] = ci; evoff[tid] = ei - ec; evcnt[tid] = 0; }
        if (tid == 0) evoff[NWIN] = ev_total;
        __syncthreads();
    }
    for (uint32_t i = tid; i < ev_total; i += T) {   // counting sort by slot
        const uint32_t e = ev_a[i];
        const uint32_t k = e >> 23;
        ev_b[evoff[k] + atomicAdd(&evcnt[k], 1u)] = e;
    }
    __syncthreads();
    for (uint32_t i = tid; i < ev_total; i += T) {   // order inside a slot by record = first-seen order (a record votes a slot once)
        const uint32_t e = ev_b[i];
        const uint32_t k = e >> 23, eb = evoff[k], ne = evcnt[k];
        uint32_t rank = 0;
        for (uint32_t x = 0; x < ne; ++x) rank += ev_b[eb + x] < e ? 1u : 0u;
        ev_a[eb + rank] = e;
    }
    __syncthreads();
    if (dbg) tm[3] = clock64();
    // ---- phase S1: one lane per slot
    uint32_t* L = lists + wave * (E - 2) * 64;
    const int64_t s64 = (int64_t)c * VOTE_CH - 2 + lane;
    const bool valid = chunk_ok && s64 >= 0 && s64 < (int64_t)S;
    const uint32_t s = (uint32_t)s64;
    const uint32_t k = valid ? s - w0 : 0u;
    const uint32_t info = valid ? win_sinfo[k] : 0u;
    const uint32_t dsym = info & 0xfu;
    const bool first = (info & SI_FIRST) != 0;
    const uint32_t prev_dsym = (valid && k >= 1) ? (win_sinfo[k - 1] & 0xfu) : 0u;
    VoteLane<E> vl;
    vl.init(valid ? win_k0[k] : 0u);
    uint32_t basemask = 1u << dsym, total = 0, nvotes = 0;
    if (valid && !bad_tile && !(ablate & 4u)) {
        const uint32_t ne = evcnt[k], eb = evoff[k];
        if (lane >= 2) {
            for (uint32_t i = 0; i < ne; ++i) {
                const uint32_t ctx = ev_a[eb + i] & 0xfffu;
                basemask |= 1u << (ctx & 0xfu);
                vl.tally(ctx, L, lane);
            }
            const uint32_t cv = cover[k];
            vl.c0 += cv - ne;   // every other covering vote carries the draft's own context
            total = (1u + cv) & 0xffffu;
            nvotes = cv;
            bmask[k] = basemask;
        } else if (wave == 0) {   // the tile's two left-context slots: only their base mask matters
            for (uint32_t i = 0; i < ne; ++i) basemask |= 1u << (ev_a[eb + i] & 0xfu);
            bmask[k] = basemask;
        }
    }
    __syncthreads();
    if (dbg) tm[4] = clock64();
    const bool single = __popc(basemask) == 1;
    const bool prev_is_single = first || (valid && k >= 1 && __popc(bmask[k - 1]) == 1) || !valid || k == 0;
    for (int o = 32; o > 0; o >>= 1) nvotes += __shfl_down(nvotes, o);
    const bool ovf_any = chunk_ok && (bad_tile || __ballot(vl.ovf) != 0ull);
    tile_epilogue<E, NW>(vl, L, tid, c, chunk_ok && !ovf_any, ovf_any, valid, s, info, dsym, first, prev_dsym, single, prev_is_single,
                         total, slot_res, slot_rec, pool, pool_cap, counters, heads, heads_cap, redo_out, redo_ci, flag_single, nvotes, votes);
    if (dbg && tid == 0) {   // cycles: setup, record loop (of which clean tests, exact items), scans + sort, tally, epilogue; tiles
        tm[5] = clock64();
        atomicAdd(&dbg[0], (unsigned long long)(tm[1] - tm[0]));
        atomicAdd(&dbg[1], (unsigned long long)(tm[2] - tm[1]));
        atomicAdd(&dbg[2], (unsigned long long)tc);
        atomicAdd(&dbg[3], (unsigned long long)tx);
        atomicAdd(&dbg[4], (unsigned long long)(tm[3] - tm[2]));
        atomicAdd(&dbg[5], (unsigned long long)(tm[4] - tm[3]));
        atomicAdd(&dbg[6], (unsigned long long)(tm[5] - tm[4]));
        atomicAdd(&dbg[7], 1ull);
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
                 uint32_t* chunk_first, uint32_t* chunk_last, uint32_t* counters, const uint8_t* dpack, uint32_t* dirty) {
    if (n_reads == 0) return;
    k_desc<<<nblk(n_reads, 256), 256, 0, st>>>(R, n_reads, ctg_off, soff, qs, qe, desc, ovf_pool, ovf_cap, chunk_first,
                                               chunk_last, counters, dpack, dirty);
}
void launch_dpack(hipStream_t st, const uint8_t* draft, uint32_t G, uint8_t* dpack) {
    if (G) k_dpack<<<nblk(((uint64_t)G + 1) / 2, 256), 256, 0, st>>>(draft, G, dpack);
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

int launch_tile7(hipStream_t st, const ReadsDev& R, const uint32_t* soff, const uint32_t* desc, const uint32_t* ovf_pool, const uint32_t* chunk_first,
                 const uint32_t* chunk_last, uint32_t n_chunks, const uint8_t* slot_info, const uint32_t* slot_g, uint32_t S, uint32_t max_lq,
                 uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool, uint32_t pool_cap, uint32_t* counters, uint32_t* heads, uint32_t heads_cap,
                 uint32_t* redo_out, uint32_t redo_ci, uint32_t flag_single, unsigned long long* votes) {
    constexpr int E7 = 8, NW7 = 8;
    const uint32_t seq_w = (((max_lq + 1) >> 1) + 3) / 4 + 1;   // packed bases per record, in words (upper bound)
    const uint32_t fixed = (uint32_t)NW7 * (uint32_t)(E7 - 2) * 64u + 8u + (uint32_t)DESC_WORDS;
    const uint32_t per = (uint32_t)DESC_WORDS + seq_w;
    static const uint32_t budget_env = getenv("NP1_TILE7_LDS_WORDS") ? (uint32_t)atoi(getenv("NP1_TILE7_LDS_WORDS")) : 13312u;   // 52 KiB: three workgroups per CU
    uint32_t budget = budget_env;
    if (budget < fixed + per) budget = fixed + per;
    if (budget > 40960u - 64u) return -1;
    uint32_t nb_max = (budget - fixed) / per;
    if (nb_max > 512u) nb_max = 512u;
    const uint32_t bytes = (fixed + nb_max * per) * 4u;
    const uint32_t items = (n_chunks + NW7 - 1) / NW7;
    if (items == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile7<E7, NW7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        attr_set = true;
    }
    k_tile7<E7, NW7><<<items, NW7 * 64, bytes, st>>>(R, soff, desc, ovf_pool, chunk_first, chunk_last, n_chunks, nullptr, items, slot_info, slot_g, S, seq_w,
                                                      nb_max, slot_res, slot_rec, pool, pool_cap, counters, heads, heads_cap, redo_out, redo_ci, flag_single, votes, ablate_env());
    return 0;
}

int launch_tile8(hipStream_t st, const ReadsDev& R, const uint32_t* soff, const uint32_t* desc, const uint32_t* dirty, const uint32_t* ovf_pool,
                 const uint32_t* chunk_first, const uint32_t* chunk_last, uint32_t n_chunks, const uint8_t* slot_info, const uint32_t* slot_g, uint32_t S,
                 uint32_t max_lq, uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool, uint32_t pool_cap, uint32_t* counters, uint32_t* heads,
                 uint32_t heads_cap, uint32_t* redo_out, uint32_t redo_ci, uint32_t flag_single, unsigned long long* votes) {
    constexpr int E8 = 8, NW8 = 8;
    const uint32_t seq_w = (((max_lq + 1) >> 1) + 3) / 4 + 1;   // packed bases per record, in words (upper bound)
    const uint32_t fixed = (uint32_t)NW8 * (uint32_t)(E8 - 2) * 64u + 8u + 4u;
    const uint32_t per = (uint32_t)DESC_WORDS + 1u + seq_w;
    static const uint32_t budget_env = getenv("NP1_TILE8_LDS_WORDS") ? (uint32_t)atoi(getenv("NP1_TILE8_LDS_WORDS")) : 13312u;   // 52 KiB: three workgroups per CU
    uint32_t budget = budget_env;
    if (budget < fixed + per) budget = fixed + per;
    if (budget > 40960u - 64u) return -1;
    uint32_t nb_max = (budget - fixed) / per;
    if (nb_max > 512u) nb_max = 512u;
    const uint32_t bytes = (fixed + nb_max * per) * 4u;
    const uint32_t items = (n_chunks + NW8 - 1) / NW8;
    if (items == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile8<E8, NW8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        attr_set = true;
    }
    k_tile8<E8, NW8><<<items, NW8 * 64, bytes, st>>>(R, soff, desc, dirty, ovf_pool, chunk_first, chunk_last, n_chunks, items, slot_info, slot_g, S, seq_w, nb_max,
                                                      slot_res, slot_rec, pool, pool_cap, counters, heads, heads_cap, redo_out, redo_ci, flag_single, votes,
                                                      ablate_env());
    return 0;
}

int launch_tile5(hipStream_t st, const ReadsDev& R, const uint32_t* soff, const uint32_t* desc, const uint32_t* ovf_pool,
                 const uint32_t* chunk_first, const uint32_t* chunk_last, uint32_t n_chunks, const uint8_t* slot_info,
                 const uint32_t* slot_g, uint32_t S, uint32_t max_lq, uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool,
                 uint32_t pool_cap, uint32_t* counters, uint32_t* heads, uint32_t heads_cap, uint32_t* redo_out, uint32_t redo_ci,
                 uint32_t flag_single, unsigned long long* votes) {
    constexpr int E5 = 8, NW5 = 8;
    constexpr uint32_t NWIN = NW5 * VOTE_CH + 2;
    const uint32_t seq_w = (((max_lq + 1) >> 1) + 3) / 4 + 2;
    const uint32_t ev_max = 2048;
    const uint32_t fixed = (uint32_t)NW5 * (E5 - 2) * 64u + (uint32_t)DESC_WORDS + 8u + NWIN * 6u + 3u + NWIN / 8 + 4 +
                           (NWIN + 2) / 2 + 1 + ((NWIN + 7) & ~3u) / 4 + 2 * ev_max + 16;
    const uint32_t per = (uint32_t)DESC_WORDS + seq_w;
    const uint32_t budget = 20352u;   // 79.5 KiB: two workgroups per CU
    if (fixed + per > 40960u - 64u) return -1;
    uint32_t nb_max = budget > fixed + per ? (budget - fixed) / per : 1u;
    if (nb_max > 512u) nb_max = 512u;
    if (nb_max < 1u) nb_max = 1u;
    const uint32_t bytes = (fixed + nb_max * per) * 4u;
    if (bytes > 160u * 1024u - 256u) return -1;
    const uint32_t items = (n_chunks + NW5 - 1) / NW5;
    if (items == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile5<E5, NW5>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - 256);
        attr_set = true;
    }
    k_tile5<E5, NW5><<<items, NW5 * 64, bytes, st>>>(R, soff, desc, ovf_pool, chunk_first, chunk_last, n_chunks, slot_info, slot_g, S,
                                                      seq_w, nb_max, ev_max, slot_res, slot_rec, pool, pool_cap, counters, heads,
                                                      heads_cap, redo_out, redo_ci, flag_single, votes, ablate_env());
    return 0;
}

int launch_tile6(hipStream_t st, const ReadsDev& R, const uint32_t* soff, const uint32_t* desc, const uint32_t* ovf_pool,
                 const uint32_t* chunk_first, const uint32_t* chunk_last, uint32_t n_chunks, const uint8_t* slot_info,
                 const uint32_t* slot_g, uint32_t S, uint32_t max_lq, uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool,
                 uint32_t pool_cap, uint32_t* counters, uint32_t* heads, uint32_t heads_cap, uint32_t* redo_out, uint32_t redo_ci,
                 uint32_t flag_single, unsigned long long* votes, unsigned long long* dbg) {
    constexpr int E6 = 8, NW6 = 8;
    constexpr uint32_t NWIN = NW6 * VOTE_CH + 2;
    const uint32_t seq_w = (((max_lq + 1) >> 1) + 3) / 4 + 2;
    const uint32_t ev_max = 2048, it_max = 4096;   // it_max u16 item codes share the second event buffer
    const uint32_t fixed = (uint32_t)NW6 * (E6 - 2) * 64u + (uint32_t)DESC_WORDS + 8u + NWIN * 5u + 2u + NWIN / 8 + 4 +
                           (NWIN + 2) / 2 + 1 + ((NWIN + 7) & ~3u) / 4 + 32 + it_max / 2 + 2 * ev_max + 16;
    const uint32_t per = (uint32_t)DESC_WORDS + seq_w;
    const uint32_t budget = 20352u;   // 79.5 KiB: two workgroups per CU
    if (fixed + per > 40960u - 64u) return -1;
    uint32_t nb_max = budget > fixed + per ? (budget - fixed) / per : 1u;
    if (nb_max > 512u) nb_max = 512u;
    if (nb_max < 1u) nb_max = 1u;
    const uint32_t bytes = (fixed + nb_max * per) * 4u;
    if (bytes > 160u * 1024u - 256u) return -1;
    const uint32_t items = (n_chunks + NW6 - 1) / NW6;
    if (items == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile6<E6, NW6>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - 256);
        attr_set = true;
    }
    k_tile6<E6, NW6><<<items, NW6 * 64, bytes, st>>>(R, soff, desc, ovf_pool, chunk_first, chunk_last, n_chunks, slot_info, slot_g, S,
                                                      seq_w, nb_max, ev_max, it_max, slot_res, slot_rec, pool, pool_cap, counters,
                                                      heads, heads_cap, redo_out, redo_ci, flag_single, votes, ablate_env(), dbg);
    return 0;
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
