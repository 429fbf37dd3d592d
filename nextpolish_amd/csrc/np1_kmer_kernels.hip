// HIP kernels of kmer_count (task 2) for gfx950.  Bodies: np1_kmer.h (shared with the host model).
//
// kmer_count re-votes only the lowercase neighbourhoods that score_chain left behind (reference:
// source/lib/kmercount.c:93-126) -- a fraction of a percent of the draft -- so unlike score_chain this is a
// sparse, irregular job.  Mapping: dense per-base / per-record / per-slot preparation kernels, then ONE LANE PER
// REGION for the reference's per-region logic (region discovery per contig on the compacted list of lowercase
// positions, insertion columns, level-2/level-1 score chain on the no-depth regions, region splitting,
// spanning-read haplotype vote), and the shared scan + k_emit for the output.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "np1_core.h"
#include "np1_kmer.h"
#include "np1_kmer_kernels.h"

namespace np1k {

static inline unsigned kblk(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }
// active lanes per wave of the lane-per-region kernels (NP1_KC_LANES, default 4: measured 27 / 18 / 16 ms per pass at 64 / 8 / 4)
static inline unsigned kc_lanes() {
    static const unsigned v = [] { const char* e = getenv("NP1_KC_LANES"); const int x = e ? atoi(e) : 4; return (unsigned)(x < 1 ? 1 : x > 64 ? 64 : x); }();
    return v;
}

// per record: filter level, end position, longest reference span
__global__ __launch_bounds__(256) void k_kc_records(KcCtx c, int64_t n_all, uint8_t* __restrict__ level,
                                                    int32_t* __restrict__ endpos, uint32_t* __restrict__ max_span) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t span = 0;
    if (r < n_all) {
        level[r] = (uint8_t)kc_filter_level(c.R, r, c.mapq, c.isize, c.read_tlen, c.max_clip_ratio_sgs, c.min_map_quality);
        const int32_t e = kc_endpos(c.R, r);
        endpos[r] = e;
        span = (uint32_t)(e - c.R.pos[r]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = __shfl_down(span, o);
        if (t > span) span = t;
    }
    if ((threadIdx.x & 63) == 0 && span) atomicMax(max_span, span);
}

// per draft base: nt16 code + FLAG_ZERO of the input draft (contig.c:92-99)
__global__ __launch_bounds__(256) void k_kc_draft(const uint8_t* __restrict__ draft, uint32_t G, uint8_t* __restrict__ code,
                                                  uint8_t* __restrict__ flag) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    uint32_t ch = draft[g], f = 0;
    if (ch >= 97 && ch <= 122) { ch -= 32; f = KC_FLAG_ZERO; }
    code[g] = (uint8_t)draft_code(ch);
    flag[g] = (uint8_t)f;
}

// compaction of the lowercase positions (fpos = exclusive scan of flag): flagged[fpos[g]] = g
__global__ __launch_bounds__(256) void k_kc_compact(const uint8_t* __restrict__ flag, const uint32_t* __restrict__ fpos, uint32_t G,
                                                    uint32_t* __restrict__ flagged) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G && flag[g]) flagged[fpos[g]] = g;
}

// one workgroup (one wave) per contig: no-depth and k-mer regions on the compacted list, merged (kmercount.c:97-106).
// Regions of a contig land contiguously and in order in the flat arrays (block allocated per contig).
//
// The reference walks the contig base by base (np1_kmer.h kc_find_regions is its sparse, still sequential, statement).
// A region never spans two flagged positions more than `gap` apart, so the list falls into RUNS at those gaps and a
// run's region (first/last position, adjacent streak at its end, closing position, edge extension incl. the
// homopolymer walk through the draft) does not depend on the other runs -- computed one run per lane.  What couples
// runs is only the walk's cursor: an extended region can swallow the first positions of the runs behind it.  A last
// sequential pass (one lane, run records staged 64 at a time through LDS) replays exactly that: a run whose first
// position lies behind the cursor is recomputed from its first surviving position (rare), the others are taken as they
// are.  Then the literal merge.
constexpr uint32_t KC_REG_LDS_WORDS = 36864;   // 144 KiB of dynamic LDS

// contig_merge_region (np1_kmer.h kc_merge_regions is the literal statement) walked by the whole wave with uniform control
// flow: lane t preloads input region base + t, the last output region lives in registers, lane 0 stores.  Needs the
// first region to have start < end (then the output never runs ahead of the input); the caller checks.
__device__ __forceinline__ int32_t kc_merge_wave(int32_t* v, int32_t n, uint32_t lane) {
    const int32_t nreg = n / 2;
    int32_t qi = 0, qs = v[0], qe = v[1], length = 2;
    for (int32_t base = 0; base < nreg; base += 64) {
        const int32_t mine = base + (int32_t)lane < nreg ? base + (int32_t)lane : nreg - 1;
        const int32_t ms = v[2 * mine], me = v[2 * mine + 1];
        const int32_t cnt = nreg - base < 64 ? nreg - base : 64;
        for (int32_t t = 0; t < cnt; ++t) {
            const int32_t ps = __builtin_amdgcn_readlane(ms, t), pe = __builtin_amdgcn_readlane(me, t);
            if (ps >= qe) {
                ++qi;
                qs = ps;
                qe = pe;
                if (lane == 0) { v[2 * qi] = qs; v[2 * qi + 1] = qe; }
                length += 2;
            } else {
                while (ps < qs) { --qi; qs = v[2 * qi]; }
                qe = pe;
                if (lane == 0) v[2 * qi + 1] = qe;
            }
        }
    }
    return length;
}

__global__ __launch_bounds__(64) void k_kc_regions(KcCtx c, uint32_t nc, const uint32_t* __restrict__ fpos,
                                                   uint32_t* __restrict__ flagged_local, int32_t* __restrict__ work,
                                                   uint32_t* __restrict__ nd_ctg, int32_t* __restrict__ nd_se,
                                                   uint32_t* __restrict__ kr_ctg, int32_t* __restrict__ kr_se,
                                                   uint32_t reg_cap, uint32_t* __restrict__ counters) {
    extern __shared__ uint32_t sh_words[];
    __shared__ uint32_t sh_nr, sh_o, sh_fail;
    const uint32_t ct = blockIdx.x, lane = threadIdx.x;
    if (ct >= nc) return;
    const uint32_t g0 = c.ctg_off[ct], g1 = c.ctg_off[ct + 1];
    const int32_t L = (int32_t)(g1 - g0);
    if (L <= 0) return;
    const uint32_t f0 = fpos[g0], f1 = fpos[g1];
    const uint32_t m = f1 - f0;
    if (m == 0) return;
    uint32_t* fl = flagged_local + f0;
    const bool list_in_lds = m <= KC_REG_LDS_WORDS;
    for (uint32_t k = lane; k < m; k += 64) fl[k] -= g0;   // global draft index -> position inside the contig
    __threadfence_block();
    __syncthreads();
    const uint32_t* F = list_in_lds ? sh_words : fl;
    // per-contig scratch in HBM: 2m + 4 values for the walk's output, m run starts, then the run records
    int32_t* gbuf = work + 12ull * f0 + 4ull * ct;
    uint32_t* rs = reinterpret_cast<uint32_t*>(gbuf + 2ull * m + 4);
    KcRun* runs = reinterpret_cast<KcRun*>(rs + m + (m & 1u));   // 8-byte aligned: 12 f0 + 4 ct + 2m + 4 + m (+1) is even
    const uint8_t* code = c.draft_code + g0;
    const uint8_t* flag = c.draft_flag + g0;
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t gap = pass == 0 ? 0u : (uint32_t)c.min_len_inter_kmer, con = pass == 0 ? (uint32_t)c.min_len_ldr : 0u;
        const bool with_ext = pass == 1;
        if (list_in_lds) {   // the list in LDS while the runs are worked out; the pass's output overlays it afterwards
            for (uint32_t k = lane; k < m; k += 64) sh_words[k] = fl[k];
            __syncthreads();
        }
        // ---- runs: starts where the distance to the previous flagged position exceeds the gap (wave compaction)
        uint32_t n_runs = 0;
        for (uint32_t base = 0; base < m; base += 64) {
            const uint32_t k = base + lane;
            const bool st = k < m && (k == 0 || F[k] - F[k - 1] - 1u > gap);
            const unsigned long long mk = __ballot(st);
            if (st) rs[n_runs + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = k;
            n_runs += (uint32_t)__popcll(mk);
        }
        __threadfence_block();
        __syncthreads();
        // ---- every run on its own (the edge extension reads the draft: this is where the latency is, now in parallel)
        for (uint32_t r = lane; r < n_runs; r += 64) {
            const uint32_t k0 = rs[r], k1 = (r + 1 < n_runs ? rs[r + 1] : m) - 1;
            KcRun rr;
            kc_run_region(F, k0, k1, code, flag, L, gap, con, c.ext_len_edge, with_ext, &rr);
            runs[r] = rr;
        }
        __threadfence_block();
        __syncthreads();
        // ---- the cursor.  It only moves past the next run's first position when an extended region reaches that far.
        // Neighbour test in parallel: a run that starts at or behind the cursor its predecessor leaves (closing position
        // + 1, or the extended end + 1) is taken as computed.  The others start short chains that are replayed like the
        // reference's walk (recompute from the first surviving position, or drop a swallowed run) until a run is again
        // taken as computed; the chains update the run records in place.  Output = the emitted runs in order.
        uint32_t* bad_list = rs;   // (the run starts are not needed any more)
        uint32_t n_bad = 0;
        for (uint32_t base = 0; base < n_runs; base += 64) {
            const uint32_t r = base + lane;
            bool bad = false;
            if (r < n_runs && r > 0) {
                const KcRun pv = runs[r - 1];
                bad = pv.emit == 2u || (int64_t)runs[r].first_pos < kc_reach(pv);
            }
            const unsigned long long mb = __ballot(bad);
            if (bad) bad_list[n_bad + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull))] = r;
            n_bad += (uint32_t)__popcll(mb);
        }
        __threadfence_block();
        __syncthreads();
        if (n_bad) {   // wave-uniform: every lane follows the same chain, lane 0 writes
            uint32_t next_q = 0;
            bool ended = false;
            for (uint32_t b = 0; b < n_bad && !ended; ++b) {
                uint32_t q = bad_list[b];
                if (q < next_q) continue;               // inside the previous chain
                q = kc_chain(F, runs, n_runs, q, code, flag, L, gap, con, c.ext_len_edge, with_ext, lane == 0, &ended);
                next_q = q + 1;
                __threadfence_block();
            }
            __syncthreads();
        }
        // the list is not needed any more in this pass: the regions go where it was when they fit
        int32_t* buf = 2ull * n_runs + 4 <= KC_REG_LDS_WORDS ? reinterpret_cast<int32_t*>(sh_words) : gbuf;
        int32_t n_out = 0;
        for (uint32_t base = 0; base < n_runs; base += 64) {
            const uint32_t r = base + lane;
            KcRun me;
            me.s = 0; me.e = 0; me.emit = 0;
            if (r < n_runs) me = runs[r];
            const bool em = r < n_runs && me.emit != 0u;
            const unsigned long long mk = __ballot(em);
            if (em) {
                const int32_t idx = n_out + 2 * (int32_t)__popcll(mk & ((1ull << lane) - 1ull));
                buf[idx] = me.s;
                buf[idx + 1] = me.e;
            }
            n_out += 2 * (int32_t)__popcll(mk);
        }
        __threadfence_block();
        __syncthreads();
        // merge: nothing to do when no region starts before its predecessor ends (the usual case; contig.c:595-620 then copies)
        bool overlap = false;
        for (int32_t i = (int32_t)lane + 1; i < n_out / 2; i += 64) overlap = overlap || buf[2 * i] < buf[2 * (i - 1) + 1];
        const bool any_overlap = __ballot(overlap) != 0ull;
        int32_t merged = n_out;
        if (n_out && !(buf[0] < buf[1])) {   // a first region of one base sends the reference's merge into its copy-everything quirk: literal
            if (lane == 0) sh_nr = (uint32_t)kc_merge_regions(buf, n_out);
            __syncthreads();
            merged = (int32_t)sh_nr;
            __syncthreads();
        } else if (any_overlap) {
            merged = kc_merge_wave(buf, n_out, lane);   // (all lanes)
        }
        if (lane == 0) {
            uint32_t fail = 0, nr = 0, o = 0;
            const int32_t k = merged;
            nr = (uint32_t)k / 2;
            if (nr) {
                o = atomicAdd(&counters[pass == 0 ? KCC_NODEPTH : KCC_KREG], nr);
                if (o + nr > reg_cap) fail = 1;
            }
            if (fail) atomicOr(c.err, ERR_KC_REGIONS);
            sh_nr = nr; sh_o = o; sh_fail = fail;
            __threadfence_block();
        }
        __syncthreads();
        if (sh_fail) return;
        const uint32_t nr = sh_nr, o = sh_o;
        uint32_t* dc = pass == 0 ? nd_ctg : kr_ctg;
        int32_t* ds = pass == 0 ? nd_se : kr_se;
        unsigned long long len_sum = 0;
        for (uint32_t i = lane; i < nr; i += 64) {
            const int32_t a = buf[2 * i], b = buf[2 * i + 1];
            dc[o + i] = ct;
            ds[2 * (o + i)] = a;
            ds[2 * (o + i) + 1] = b;
            len_sum += (unsigned long long)(b - a + 1);
        }
        if (pass == 0) {
            for (int sft = 32; sft > 0; sft >>= 1) len_sum += __shfl_down(len_sum, sft);
            if (lane == 0 && len_sum) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[KCC_ND_LEN]), len_sum);
        }
        __syncthreads();
    }
}

// one lane per region (k-mer regions first, then no-depth regions): insertion columns (contig.c:182-245)
__global__ __launch_bounds__(64) void k_kc_inserts(KcCtx c, const uint32_t* __restrict__ kr_ctg, const int32_t* __restrict__ kr_se,
                                                   uint32_t n_kr, const uint32_t* __restrict__ nd_ctg,
                                                   const int32_t* __restrict__ nd_se, uint32_t n_nd, uint32_t* __restrict__ ins) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_kr) kc_insert_region(c, kr_ctg[i], kr_se[2 * i], kr_se[2 * i + 1], ins);
    else if (i < n_kr + n_nd) kc_insert_region(c, nd_ctg[i - n_kr], nd_se[2 * (i - n_kr)], nd_se[2 * (i - n_kr) + 1], ins);
}

// per slot: working base / flag from slot_info; empty context lists
__global__ __launch_bounds__(256) void k_kc_slots(const uint8_t* __restrict__ slot_info, uint32_t S, uint8_t* __restrict__ sbase,
                                                  uint8_t* __restrict__ sflag, uint16_t* __restrict__ scount,
                                                  uint32_t* __restrict__ lhead) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const uint32_t info = slot_info[s];
    sbase[s] = (uint8_t)(info & 0xf);
    sflag[s] = (uint8_t)((info & SI_LOWER) ? KC_FLAG_ZERO : 0);
    scount[s] = 0;
    lhead[s] = 0;
}

// one lane per chain of no-depth regions (regions sharing an end point are solved in order by the same lane,
// like the reference's sequential loop, kmercount.c:107-114)
__global__ __launch_bounds__(64) void k_kc_nodepth(KcCtx c, const uint32_t* __restrict__ nd_ctg, const int32_t* __restrict__ nd_se,
                                                   uint32_t n_nd, uint32_t lanes) {
    // every lane walks one region serially (records, slots, DP): few regions per wave keep the waves short and spread
    // them over all SIMDs (`lanes` active lanes per wave)
    const uint32_t i = blockIdx.x * lanes + threadIdx.x;
    if (threadIdx.x >= lanes || i >= n_nd) return;
    const uint32_t ct = nd_ctg[i];
    if (i > 0 && nd_ctg[i - 1] == ct && nd_se[2 * (i - 1) + 1] == nd_se[2 * i]) return;   // not a chain head
    for (uint32_t k = i;; ++k) {
        kc_score_correct_level2(c, ct, nd_se[2 * k], nd_se[2 * k + 1]);
        if (k + 1 >= n_nd || nd_ctg[k + 1] != ct || nd_se[2 * (k + 1)] != nd_se[2 * k + 1]) break;
    }
}

// split pass 1/2: parts per k-mer region (count), then fill at scanned offsets
__global__ __launch_bounds__(64) void k_kc_split(KcCtx c, const uint32_t* __restrict__ kr_ctg, const int32_t* __restrict__ kr_se,
                                                 uint32_t n_kr, int32_t* __restrict__ work, const uint32_t* __restrict__ work_off,
                                                 uint32_t* __restrict__ n_parts, const uint32_t* __restrict__ part_off,
                                                 uint32_t* __restrict__ pt_ctg, int32_t* __restrict__ pt_se,
                                                 uint32_t* __restrict__ pt_len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_kr) return;
    const uint32_t ct = kr_ctg[i];
    const int32_t rs = kr_se[2 * i], re = kr_se[2 * i + 1];
    if (part_off == nullptr) {   // count: an upper bound without walking (one cut per 2 positions) is too loose; walk
        int32_t* buf = work + work_off[i];
        int32_t np = kc_split_region(c, ct, rs, re, buf, (int32_t)(work_off[i + 1] - work_off[i]));
        if (np < 0) { atomicOr(c.err, ERR_KC_REGIONS); np = 0; }
        n_parts[i] = (uint32_t)np / 2;
        return;
    }
    const int32_t* buf = work + work_off[i];
    const uint32_t o = part_off[i], np = n_parts[i];
    const uint32_t g0 = c.ctg_off[ct];
    for (uint32_t k = 0; k < np; ++k) {
        pt_ctg[o + k] = ct;
        pt_se[2 * (o + k)] = buf[2 * k];
        pt_se[2 * (o + k) + 1] = buf[2 * k + 1];
        pt_len[o + k] = c.soff[g0 + (uint32_t)buf[2 * k + 1]] - c.soff[g0 + (uint32_t)buf[2 * k]] + 1;
    }
}

// one lane per part: winner haplotype into wpool[woff[p] ..) (kmercount.c:175-261)
__global__ __launch_bounds__(64) void k_kc_winner(KcCtx c, const uint32_t* __restrict__ pt_ctg, const int32_t* __restrict__ pt_se,
                                                  const uint32_t* __restrict__ pt_len, const uint32_t* __restrict__ woff,
                                                  uint32_t n_parts, int64_t n_all, uint8_t* __restrict__ wpool,
                                                  uint8_t* __restrict__ has_winner, uint32_t lanes, const uint32_t* __restrict__ rp_first,
                                                  const uint32_t* __restrict__ rp_list, const long long* __restrict__ rp_stale,
                                                  const int32_t* __restrict__ rp_n2, uint32_t* __restrict__ rp_brk) {
    const uint32_t p = blockIdx.x * lanes + threadIdx.x;   // `lanes` parts per wave (see k_kc_nodepth)
    if (threadIdx.x >= lanes || p >= n_parts) return;
    const uint32_t ct = pt_ctg[p];
    if (ct & 0x80000000u) {   // a slot of snp_valid's second round that is unused (0xffffffff) or holds an inverted pair (contig | 1 << 31)
        has_winner[p] = 0;
        if (rp_brk) rp_brk[p] = 0;
        return;
    }
    const bool has_next = (int64_t)c.read_begin[ct + 1] < n_all;
    if (rp_first) {   // the records of this part as the replayed iterator hands them out (np1_replay.h); result 2 = second loop not known yet
        const KcReplay rp{rp_list + rp_first[p], rp_first[p + 1] - rp_first[p], (int64_t)rp_stale[p], rp_n2 ? rp_n2[p] : -1, rp_brk ? rp_brk + p : nullptr};
        has_winner[p] = (uint8_t)kc_part_winner(c, ct, pt_se[2 * p], pt_se[2 * p + 1], has_next, wpool + woff[p], (int32_t)pt_len[p], &rp);
        return;
    }
    has_winner[p] = (uint8_t)kc_part_winner(c, ct, pt_se[2 * p], pt_se[2 * p + 1], has_next, wpool + woff[p], (int32_t)pt_len[p]);
}

// one lane per part: contig_update_contig (contig.c:811-821).  Parts are processed in order by the reference and
// consecutive parts share their boundary position: the later part's value stays, so a part leaves its last slot to
// its successor when that one has a winner.
__global__ __launch_bounds__(64) void k_kc_apply(KcCtx c, const uint32_t* __restrict__ pt_ctg, const int32_t* __restrict__ pt_se,
                                                 const uint32_t* __restrict__ pt_len, const uint32_t* __restrict__ woff,
                                                 uint32_t n_parts, const uint8_t* __restrict__ wpool,
                                                 const uint8_t* __restrict__ has_winner) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parts || !has_winner[p]) return;
    const uint32_t ct = pt_ctg[p];
    uint32_t n = pt_len[p];
    // successors starting exactly at my end (a chain of zero-length... parts [k,k] can repeat): any winner among them wins
    for (uint32_t q = p + 1; q < n_parts && pt_ctg[q] == ct && pt_se[2 * q] == pt_se[2 * p + 1]; ++q) {
        if (has_winner[q]) { --n; break; }
        if (pt_se[2 * q + 1] != pt_se[2 * q]) break;   // only single-position parts keep sharing the same start
    }
    const uint32_t s0 = c.soff[c.ctg_off[ct] + (uint32_t)pt_se[2 * p]];
    const uint8_t* w = wpool + woff[p];
    for (uint32_t t = 0; t < n; ++t) c.sbase[s0 + t] = w[t];
}

__global__ __launch_bounds__(256) void k_kc_result(const uint8_t* __restrict__ sbase, const uint8_t* __restrict__ sflag, uint32_t S,
                                                   uint16_t* __restrict__ slot_res) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) slot_res[s] = (uint16_t)(sbase[s] | (uint32_t)sflag[s] << 8);
}


// ---- snp_valid (task 4, snpvalid.c:3-36): the parts are the ones of kmer_count; what differs is what happens around the votes --
// The parts of one contig are one contiguous run of pt_ctg[] (its regions come out of k_kc_regions as one block, the blocks of
// different contigs in any order): range[2 ct], range[2 ct + 1] = first part and one past the last (both 0: no parts).
__global__ __launch_bounds__(256) void k_sv_ranges(const uint32_t* __restrict__ pt_ctg, uint32_t n_parts, uint32_t* __restrict__ range) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parts) return;
    const uint32_t ct = pt_ctg[p];
    if (p == 0 || pt_ctg[p - 1] != ct) range[2 * ct] = p;
    if (p + 1 == n_parts || pt_ctg[p + 1] != ct) range[2 * ct + 1] = p + 1;
}
// Round 1, after the votes (the marks were left alone): one lane per contig takes its parts in order like ss_kmer_correct does
// (kmercount.c:221-249 with flagzero = 1): a part with a winner loses its FLAG_ZERO marks and takes the winner's bases, a part
// nothing spanned goes on the contig's list for round 2 (fail_se at the slots of the contig's own parts).
__global__ __launch_bounds__(64) void k_sv_round1(KcCtx c, uint32_t nc, const uint32_t* __restrict__ range, const int32_t* __restrict__ pt_se,
                                                  const uint32_t* __restrict__ pt_len, const uint32_t* __restrict__ woff, uint32_t n_parts,
                                                  const uint8_t* __restrict__ wpool, const uint8_t* __restrict__ has_winner,
                                                  int32_t* __restrict__ fail_se, uint32_t* __restrict__ fail_cnt) {
    const uint32_t ct = blockIdx.x * blockDim.x + threadIdx.x;
    if (ct >= nc) return;
    const uint32_t p0 = range[2 * ct], p1 = range[2 * ct + 1];
    const uint32_t g0 = c.ctg_off[ct];
    uint32_t nf = 0;
    for (uint32_t p = p0; p < p1; ++p) {
        if (has_winner[p]) {
            const uint32_t s0 = c.soff[g0 + (uint32_t)pt_se[2 * p]], n = pt_len[p];
            const uint8_t* w = wpool + woff[p];
            for (uint32_t t = 0; t < n; ++t) { c.sflag[s0 + t] = (uint8_t)(c.sflag[s0 + t] & ~KC_FLAG_ZERO); c.sbase[s0 + t] = w[t]; }
        } else {
            fail_se[2 * (p0 + nf)] = pt_se[2 * p];
            fail_se[2 * (p0 + nf) + 1] = pt_se[2 * p + 1];
            ++nf;
        }
    }
    fail_cnt[ct] = nf;
}
// Round 2 parts: fts_spilt_region over the contig's failed parts appends to ONE list -- the reference re-uses the list object that
// held the round-1 parts (start0, end0, start1, end1 ...) with its length set to 0 and its contents left in place -- and
// ss_kmer_correct then reads that list pairwise: an odd number of values makes the last pair end at whatever the list held at that
// index before (a round-1 value, or zero-filled memory behind it).  Pairs with start > end do nothing in the reference (every loop
// over them is empty) and are dropped here.  val[] / the round-2 part slots of a contig start at voff[first part of the contig].
__global__ __launch_bounds__(64) void k_sv_round2_parts(KcCtx c, uint32_t nc, const uint32_t* __restrict__ range, const int32_t* __restrict__ pt_se,
                                                        uint32_t n_parts, const uint32_t* __restrict__ voff, const int32_t* __restrict__ fail_se,
                                                        const uint32_t* __restrict__ fail_cnt, int32_t* __restrict__ val,
                                                        uint32_t* __restrict__ p2_ctg, int32_t* __restrict__ p2_se, uint32_t* __restrict__ p2_len) {
    const uint32_t ct = blockIdx.x * blockDim.x + threadIdx.x;
    if (ct >= nc) return;
    const uint32_t p0 = range[2 * ct], p1 = range[2 * ct + 1];
    if (p0 == p1) return;
    const uint32_t v0 = voff[p0], cap = voff[p1] - v0;
    int32_t* out = val + v0;
    int32_t n = 0;
    for (uint32_t k = 0; k < fail_cnt[ct] && n >= 0; ++k) n = kc_fts_split(c, ct, fail_se[2 * (p0 + k)], fail_se[2 * (p0 + k) + 1], out, n, (int32_t)cap - 1);
    if (n < 0) { atomicOr(c.err, ERR_KC_REGIONS); return; }
    if (n & 1) {
        const uint32_t n_old = 2 * (p1 - p0);
        out[n] = (uint32_t)n < n_old ? pt_se[2 * p0 + (uint32_t)n] : 0;
        ++n;
    }
    const uint32_t g0 = c.ctg_off[ct];
    for (int32_t k = 0; k < n / 2; ++k) {
        const int32_t a = out[2 * k], b = out[2 * k + 1];
        const uint32_t q = v0 + (uint32_t)k;
        p2_se[2 * q] = a;
        p2_se[2 * q + 1] = b;
        if (a > b) { p2_ctg[q] = ct | 0x80000000u; continue; }   // an inverted pair: no votes (p2_len stays 0); the iterator replay still reads its end as the nextposend of the pair before
        p2_ctg[q] = ct;
        p2_len[q] = c.soff[g0 + (uint32_t)b] - c.soff[g0 + (uint32_t)a] + 1;
    }
}
// Round 2, after the votes: the winners' bases in list order (contig_update_contig; later parts overwrite earlier ones)
__global__ __launch_bounds__(64) void k_sv_round2_apply(KcCtx c, uint32_t nc, const uint32_t* __restrict__ range, uint32_t n_parts,
                                                        const uint32_t* __restrict__ voff, const uint32_t* __restrict__ p2_ctg,
                                                        const int32_t* __restrict__ p2_se, const uint32_t* __restrict__ p2_len,
                                                        const uint32_t* __restrict__ woff2, const uint8_t* __restrict__ wpool,
                                                        const uint8_t* __restrict__ has_winner) {
    const uint32_t ct = blockIdx.x * blockDim.x + threadIdx.x;
    if (ct >= nc) return;
    const uint32_t p0 = range[2 * ct], p1 = range[2 * ct + 1];
    if (p0 == p1) return;
    const uint32_t g0 = c.ctg_off[ct];
    for (uint32_t q = voff[p0]; q < voff[p1]; ++q) {
        if (p2_ctg[q] != ct || !has_winner[q]) continue;
        const uint32_t s0 = c.soff[g0 + (uint32_t)p2_se[2 * q]], n = p2_len[q];
        const uint8_t* w = wpool + woff2[q];
        for (uint32_t t = 0; t < n; ++t) c.sbase[s0 + t] = w[t];
    }
}
__global__ __launch_bounds__(256) void k_sv_val_sizes(const uint32_t* __restrict__ pt_len, uint32_t n_parts, uint32_t* __restrict__ vsz) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_parts) vsz[p] = pt_len[p] + 3u;      // values one failed part can append (two per unflagged run + its end) + the odd tail
}

// ---- launchers ----------------------------------------------------------------------------------
void kc_launch_records(hipStream_t st, const KcCtx& c, int64_t n_all, uint8_t* level, int32_t* endpos, uint32_t* max_span) {
    if (n_all > 0) k_kc_records<<<kblk(n_all, 256), 256, 0, st>>>(c, n_all, level, endpos, max_span);
}
void kc_launch_draft(hipStream_t st, const uint8_t* draft, uint32_t G, uint8_t* code, uint8_t* flag) {
    if (G) k_kc_draft<<<kblk(G, 256), 256, 0, st>>>(draft, G, code, flag);
}
void kc_launch_compact(hipStream_t st, const uint8_t* flag, const uint32_t* fpos, uint32_t G, uint32_t* flagged) {
    if (G) k_kc_compact<<<kblk(G, 256), 256, 0, st>>>(flag, fpos, G, flagged);
}
void kc_launch_regions(hipStream_t st, const KcCtx& c, uint32_t nc, const uint32_t* fpos, uint32_t* flagged, int32_t* work,
                       uint32_t* nd_ctg, int32_t* nd_se, uint32_t* kr_ctg, int32_t* kr_se, uint32_t reg_cap, uint32_t* counters) {
    if (nc) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kc_regions), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KC_REG_LDS_WORDS * 4));
            attr_set = true;
        }
        k_kc_regions<<<nc, 64, KC_REG_LDS_WORDS * 4, st>>>(c, nc, fpos, flagged, work, nd_ctg, nd_se, kr_ctg, kr_se, reg_cap, counters);
    }
}
void kc_launch_inserts(hipStream_t st, const KcCtx& c, const uint32_t* kr_ctg, const int32_t* kr_se, uint32_t n_kr,
                       const uint32_t* nd_ctg, const int32_t* nd_se, uint32_t n_nd, uint32_t* ins) {
    if (n_kr + n_nd) k_kc_inserts<<<kblk(n_kr + n_nd, 64), 64, 0, st>>>(c, kr_ctg, kr_se, n_kr, nd_ctg, nd_se, n_nd, ins);
}
void kc_launch_slots(hipStream_t st, const uint8_t* slot_info, uint32_t S, uint8_t* sbase, uint8_t* sflag, uint16_t* scount,
                     uint32_t* lhead) {
    if (S) k_kc_slots<<<kblk(S, 256), 256, 0, st>>>(slot_info, S, sbase, sflag, scount, lhead);
}
void kc_launch_nodepth(hipStream_t st, const KcCtx& c, const uint32_t* nd_ctg, const int32_t* nd_se, uint32_t n_nd) {
    if (n_nd) k_kc_nodepth<<<kblk(n_nd, kc_lanes()), 64, 0, st>>>(c, nd_ctg, nd_se, n_nd, kc_lanes());
}
void kc_launch_split(hipStream_t st, const KcCtx& c, const uint32_t* kr_ctg, const int32_t* kr_se, uint32_t n_kr, int32_t* work,
                     const uint32_t* work_off, uint32_t* n_parts, const uint32_t* part_off, uint32_t* pt_ctg, int32_t* pt_se,
                     uint32_t* pt_len) {
    if (n_kr) k_kc_split<<<kblk(n_kr, 64), 64, 0, st>>>(c, kr_ctg, kr_se, n_kr, work, work_off, n_parts, part_off, pt_ctg, pt_se, pt_len);
}
void kc_launch_winner(hipStream_t st, const KcCtx& c, const uint32_t* pt_ctg, const int32_t* pt_se, const uint32_t* pt_len,
                      const uint32_t* woff, uint32_t n_parts, int64_t n_all, uint8_t* wpool, uint8_t* has_winner) {
    if (n_parts) k_kc_winner<<<kblk(n_parts, kc_lanes()), 64, 0, st>>>(c, pt_ctg, pt_se, pt_len, woff, n_parts, n_all, wpool, has_winner, kc_lanes(), nullptr, nullptr, nullptr, nullptr, nullptr);
}
void kc_launch_winner_replay(hipStream_t st, const KcCtx& c, const uint32_t* pt_ctg, const int32_t* pt_se, const uint32_t* pt_len, const uint32_t* woff, uint32_t n_parts,
                             int64_t n_all, uint8_t* wpool, uint8_t* has_winner, const uint32_t* rp_first, const uint32_t* rp_list, const long long* rp_stale,
                             const int32_t* rp_n2, uint32_t* rp_brk) {
    if (n_parts)
        k_kc_winner<<<kblk(n_parts, kc_lanes()), 64, 0, st>>>(c, pt_ctg, pt_se, pt_len, woff, n_parts, n_all, wpool, has_winner, kc_lanes(), rp_first, rp_list, rp_stale, rp_n2,
                                                             rp_brk);
}
void kc_launch_apply(hipStream_t st, const KcCtx& c, const uint32_t* pt_ctg, const int32_t* pt_se, const uint32_t* pt_len,
                     const uint32_t* woff, uint32_t n_parts, const uint8_t* wpool, const uint8_t* has_winner) {
    if (n_parts) k_kc_apply<<<kblk(n_parts, 64), 64, 0, st>>>(c, pt_ctg, pt_se, pt_len, woff, n_parts, wpool, has_winner);
}
void kc_launch_result(hipStream_t st, const uint8_t* sbase, const uint8_t* sflag, uint32_t S, uint16_t* slot_res) {
    if (S) k_kc_result<<<kblk(S, 256), 256, 0, st>>>(sbase, sflag, S, slot_res);
}

void sv_launch_val_sizes(hipStream_t st, const uint32_t* pt_len, uint32_t n_parts, uint32_t* vsz) {
    if (n_parts) k_sv_val_sizes<<<kblk(n_parts, 256), 256, 0, st>>>(pt_len, n_parts, vsz);
}
void sv_launch_ranges(hipStream_t st, const uint32_t* pt_ctg, uint32_t n_parts, uint32_t* range) {
    if (n_parts) k_sv_ranges<<<kblk(n_parts, 256), 256, 0, st>>>(pt_ctg, n_parts, range);
}
void sv_launch_round1(hipStream_t st, const KcCtx& c, uint32_t nc, const uint32_t* range, const int32_t* pt_se, const uint32_t* pt_len,
                      const uint32_t* woff, uint32_t n_parts, const uint8_t* wpool, const uint8_t* has_winner, int32_t* fail_se,
                      uint32_t* fail_cnt) {
    if (nc) k_sv_round1<<<kblk(nc, 64), 64, 0, st>>>(c, nc, range, pt_se, pt_len, woff, n_parts, wpool, has_winner, fail_se, fail_cnt);
}
void sv_launch_round2_parts(hipStream_t st, const KcCtx& c, uint32_t nc, const uint32_t* range, const int32_t* pt_se, uint32_t n_parts,
                            const uint32_t* voff, const int32_t* fail_se, const uint32_t* fail_cnt, int32_t* val, uint32_t* p2_ctg,
                            int32_t* p2_se, uint32_t* p2_len) {
    if (nc) k_sv_round2_parts<<<kblk(nc, 64), 64, 0, st>>>(c, nc, range, pt_se, n_parts, voff, fail_se, fail_cnt, val, p2_ctg, p2_se, p2_len);
}
void sv_launch_round2_apply(hipStream_t st, const KcCtx& c, uint32_t nc, const uint32_t* range, uint32_t n_parts, const uint32_t* voff,
                            const uint32_t* p2_ctg, const int32_t* p2_se, const uint32_t* p2_len, const uint32_t* woff2,
                            const uint8_t* wpool, const uint8_t* has_winner) {
    if (nc) k_sv_round2_apply<<<kblk(nc, 64), 64, 0, st>>>(c, nc, range, n_parts, voff, p2_ctg, p2_se, p2_len, woff2, wpool, has_winner);
}

}  // namespace np1k
