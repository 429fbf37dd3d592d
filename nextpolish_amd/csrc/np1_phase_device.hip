// snp_phase (task 3) on the GPU: kernels around the stage bodies of np1_phase.h and the launch sequence np1_batch_snp_phase.
// Reference: source/lib/snpphase.c:87-134 (snp_phase) and what it calls.  Two resident batches: the short reads and the long
// reads of the same contigs.  The host only lays out lists (regions, kept sites) and runs the chain over the sites with its own
// libm (np1_phase_host.h); every pass over records, slots and sites is a kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np1_batch_priv.h"
#include "np1_kmer_kernels.h"
#include "np1_phase.h"
#include "np1_phase_host.h"

using namespace np1dev;
using namespace np1k;
using namespace np1p;

namespace {

inline unsigned nblk(uint64_t n, unsigned per) { const uint64_t k = (n + per - 1) / per; return (unsigned)(k ? k : 1); }

__global__ __launch_bounds__(256) void k_sp_lr_records(ReadsDev R, int64_t n, double max_clip, uint8_t* __restrict__ level, int32_t* __restrict__ endpos,
                                                       uint32_t* __restrict__ max_span) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t span = 0;
    if (r < n) {
        level[r] = (uint8_t)sp_lr_level(R, r, max_clip);
        const int32_t e = kc_endpos(R, r);
        endpos[r] = e;
        span = (uint32_t)(e - R.pos[r]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = __shfl_down(span, o);
        if (t > span) span = t;
    }
    if ((threadIdx.x & 63) == 0 && span) atomicMax(max_span, span);
}

__global__ __launch_bounds__(256) void k_sp_insert(ReadsDev R, int64_t n, const uint8_t* __restrict__ level, const uint32_t* __restrict__ ctg_off,
                                                   uint32_t* __restrict__ ins, uint32_t mask, const uint32_t* __restrict__ soff, const uint8_t* __restrict__ sflag) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && level[r] >= 1) sp_insert_record(R, r, ctg_off, ins, mask, soff, sflag);
}

__global__ __launch_bounds__(256) void k_sp_hist(KcCtx c, int64_t n, uint32_t* __restrict__ cnt, uint32_t* __restrict__ first) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) sp_hist_record(c, r, cnt, first);
}

// The same histogram through LDS.  Records are position sorted, so the 128 records of a workgroup vote into a short run of slots
// (128 read starts + one read length); those votes go to a window of HW slots x 6 common symbols (A C G T, deletion, N) in LDS and
// reach HBM once per (slot, symbol) and workgroup instead of once per vote -- at 30x that is ~20x fewer global atomics.  Votes outside
// the window or with another symbol take the direct path; the result is the same whatever the window is.
constexpr uint32_t HW = 1024;
typedef __attribute__((address_space(3))) uint32_t sp_lds_u32;
struct SpHistLdsSink {
    const uint32_t* soff;
    uint32_t g0, slot_lo;
    sp_lds_u32* l_cnt;      // typed LDS pointers: the window's atomics must come out as ds_ operations, not flat ones
    sp_lds_u32* l_first;
    uint32_t* cnt;
    uint32_t* first;
    uint32_t r;
    __device__ __forceinline__ void vote(int32_t pos, uint32_t col, uint32_t sym, int32_t, bool) {
        const uint32_t s = soff[g0 + (uint32_t)pos] + col;
        const uint32_t idx = s - slot_lo;   // (wraps to a huge value below the window)
        const uint32_t m = sym == 1 ? 0u : sym == 2 ? 1u : sym == 4 ? 2u : sym == 8 ? 3u : sym == 3 ? 4u : sym == 15 ? 5u : 6u;
        if (idx < HW && m < 6) {
            __hip_atomic_fetch_add(l_cnt + (idx * 6 + m), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_min(l_first + (idx * 6 + m), r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            const uint64_t k = (uint64_t)s * 16 + sym;
            atomicAdd(&cnt[k], 1u);
            atomicMin(&first[k], r);
        }
    }
};
__global__ __launch_bounds__(128) void k_sp_hist_lds(KcCtx c, int64_t n, uint32_t* __restrict__ cnt, uint32_t* __restrict__ first) {
    __shared__ uint32_t l_cnt[HW * 6];
    __shared__ uint32_t l_first[HW * 6];
    __shared__ uint32_t s_lo;
    const int64_t r0 = (int64_t)blockIdx.x * 128;
    for (uint32_t i = threadIdx.x; i < HW * 6; i += 128) { l_cnt[i] = 0; l_first[i] = 0xffffffffu; }
    if (threadIdx.x == 0) {
        const uint32_t ct = c.R.ctg[r0];
        const int32_t p = c.R.pos[r0];
        s_lo = c.soff[c.ctg_off[ct] + (uint32_t)(p > 0 ? p : 0)];
    }
    __syncthreads();
    const int64_t r = r0 + threadIdx.x;
    if (r < n && c.level[r] == 2 && c.R.n_cigar[r] != 0) {
        const uint32_t ct = c.R.ctg[r];
        const uint32_t g0 = c.ctg_off[ct];
        SpHistLdsSink sink{c.soff, g0, s_lo, (sp_lds_u32*)l_cnt, (sp_lds_u32*)l_first, cnt, first, (uint32_t)r};
        kc_walk(c, r, g0, 0, (int32_t)(c.ctg_off[ct + 1] - g0) - 1, sink);
    }
    __syncthreads();
    static const uint32_t kSym[6] = {1, 2, 4, 8, 3, 15};
    for (uint32_t i = threadIdx.x; i < HW * 6; i += 128) {
        const uint32_t v = l_cnt[i];
        if (!v) continue;
        const uint64_t k = (uint64_t)(s_lo + i / 6) * 16 + kSym[i % 6];
        atomicAdd(&cnt[k], v);
        atomicMin(&first[k], l_first[i]);
    }
}

__global__ __launch_bounds__(256) void k_sp_decide(SpParams P, uint32_t S, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ first, uint8_t* sbase,
                                                   uint8_t* sflag, uint16_t* scount, uint8_t* dec, uint8_t* top, uint32_t* err) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) sp_slot_decide(P, s, cnt, first, sbase, sflag, scount, dec, top, err);
}

__device__ __forceinline__ uint32_t ctg_of(const uint32_t* ctg_off, uint32_t nc, uint32_t g) {   // contig of global base g
    uint32_t lo = 0, hi = nc;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ctg_off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_sp_sites(uint32_t G, const uint32_t* __restrict__ ctg_off, uint32_t nc, const uint32_t* __restrict__ soff,
                                                  const uint8_t* __restrict__ dec, const uint8_t* __restrict__ top, uint8_t* __restrict__ sflag,
                                                  uint8_t* __restrict__ dirty, uint8_t* __restrict__ alle) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint32_t ct = ctg_of(ctg_off, nc, g);
    uint8_t a = 0;
    const uint32_t d = sp_base_site(g, (int32_t)(g - ctg_off[ct]), (int32_t)(ctg_off[ct + 1] - ctg_off[ct]), soff, dec, top, &a);
    dirty[g] = (uint8_t)d;
    alle[g] = a;
    if (d) sflag[soff[g]] = (uint8_t)(sflag[soff[g]] | F_SNP);
}

__global__ __launch_bounds__(256) void k_sp_site_compact(uint32_t G, const uint32_t* __restrict__ ctg_off, uint32_t nc, const uint8_t* __restrict__ dirty,
                                                         const uint32_t* __restrict__ dpos, uint32_t* __restrict__ site_g, uint32_t* __restrict__ site_ctg) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G || !dirty[g]) return;
    site_g[dpos[g]] = g;
    site_ctg[dpos[g]] = ctg_of(ctg_off, nc, g);
}

__global__ __launch_bounds__(256) void k_sp_anchors(uint32_t NS, const uint32_t* __restrict__ site_g, const uint32_t* __restrict__ site_ctg,
                                                    const uint32_t* __restrict__ ctg_off, const uint8_t* __restrict__ dirty, int32_t* __restrict__ pos,
                                                    int32_t* __restrict__ left, int32_t* __restrict__ right) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= NS) return;
    const uint32_t g0 = ctg_off[site_ctg[k]];
    const int32_t i = (int32_t)(site_g[k] - g0);
    pos[k] = i;
    sp_site_anchors(dirty + g0, i, (int32_t)(ctg_off[site_ctg[k] + 1] - g0), &left[k], &right[k]);
}

__global__ __launch_bounds__(256) void k_sp_depthmark(uint32_t S, const uint8_t* __restrict__ sflag, uint8_t* __restrict__ mark) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) mark[s] = (sflag[s] & F_DEPTH) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_sp_compact_slots(uint32_t S, const uint8_t* __restrict__ mark, const uint32_t* __restrict__ mpos, uint32_t* __restrict__ F) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S && mark[s]) F[mpos[s]] = s;
}

// one lane per contig: the sparse region walk over its marked slots, then the literal merge; regions at out + 2 * (mpos[first slot] + ct)
__global__ __launch_bounds__(64) void k_sp_depth(uint32_t nc, const uint32_t* __restrict__ ctg_off, const uint32_t* __restrict__ soff, const uint32_t* __restrict__ sown,
                                                 const uint32_t* __restrict__ mpos, const uint32_t* __restrict__ F, uint32_t gap, int32_t ext, int32_t* __restrict__ out,
                                                 uint32_t* __restrict__ n_out, uint32_t* __restrict__ f0_out, uint32_t* err) {
    const uint32_t ct = blockIdx.x * blockDim.x + threadIdx.x;
    if (ct >= nc) return;
    const uint32_t g0 = ctg_off[ct], g1 = ctg_off[ct + 1];
    n_out[ct] = 0;
    f0_out[ct] = 0;
    if (g1 == g0) return;
    const uint32_t f0 = mpos[soff[g0]], f1 = mpos[soff[g1 - 1] + 1];
    f0_out[ct] = f0;
    int32_t* o = out + 2ull * ((uint64_t)f0 + ct);
    int32_t n = sp_depth_regions(F + f0, f1 - f0, soff, sown, g0, (int32_t)(g1 - g0), gap, ext, o, (int32_t)(2 * (f1 - f0) + 2));
    if (n < 0) { atomicOr(err, ERR_KC_REGIONS); return; }
    n = kc_merge_regions(o, n);
    n_out[ct] = (uint32_t)n;
}

__global__ __launch_bounds__(256) void k_sp_mark_regions(uint32_t n_reg, const uint32_t* __restrict__ reg_ctg, const int32_t* __restrict__ reg_se,
                                                         const uint32_t* __restrict__ ctg_off, const uint32_t* __restrict__ soff, uint8_t* sflag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_reg) return;
    const uint32_t g0 = ctg_off[reg_ctg[k]];
    for (uint32_t s = soff[g0 + (uint32_t)reg_se[2 * k]]; s <= soff[g0 + (uint32_t)reg_se[2 * k + 1]]; ++s) sp_atomic_or8(&sflag[s], F_INSERT);
}

__global__ __launch_bounds__(256) void k_sp_reslot(uint32_t G, const uint32_t* __restrict__ soff1, const uint32_t* __restrict__ soff2, const uint8_t* __restrict__ sbase1,
                                                   const uint8_t* __restrict__ sflag1, const uint16_t* __restrict__ scount1, uint8_t* __restrict__ sbase2,
                                                   uint8_t* __restrict__ sflag2, uint16_t* __restrict__ scount2, uint32_t* __restrict__ sown2) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) sp_reslot_base(g, soff1, soff2, sbase1, sflag1, scount1, sbase2, sflag2, scount2, sown2);
}

__global__ __launch_bounds__(256) void k_sp_site_stride(uint32_t NS, const uint32_t* __restrict__ site_g, const uint32_t* __restrict__ soff, uint32_t* __restrict__ rstride,
                                                        uint32_t* __restrict__ rbytes, int32_t* __restrict__ len, uint8_t* __restrict__ keep) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= NS) return;
    const uint32_t st = soff[site_g[k] + 1] - soff[site_g[k]] + 1 + 8;
    rstride[k] = st;
    rbytes[k] = 2 * st;
    len[k] = 1;
    keep[k] = 0;
}
__global__ __launch_bounds__(256) void k_sp_site_init(uint32_t NS, const uint32_t* __restrict__ site_g, const uint8_t* __restrict__ alle, const uint32_t* __restrict__ roff,
                                                      const uint32_t* __restrict__ rstride, uint8_t* __restrict__ rpool) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= NS) return;
    rpool[roff[k]] = alle[site_g[k]] & 0xf;
    rpool[roff[k] + rstride[k]] = alle[site_g[k]] >> 4;
}

// (site and region lanes diverge completely -- each walks its own records: `lanes` active lanes per wave, the rest idle, spreads them over
// more waves; same finding as for kmer_count's region kernels, np1_kmer_kernels.hip)
__global__ __launch_bounds__(64) void k_sp_verdict(KcCtx cs, KcCtx cl, SpParams P, SpSites S, uint32_t NS, uint32_t lanes, const uint32_t* __restrict__ soff1,
                                                   const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ first) {
    const uint32_t k = blockIdx.x * lanes + threadIdx.x;
    if (threadIdx.x < lanes && k < NS) sp_site_verdict(cs, cl, P, S, k, soff1, cnt, first);
}

__global__ __launch_bounds__(256) void k_sp_region_slots(uint32_t n_reg, const uint32_t* __restrict__ reg_ctg, const int32_t* __restrict__ reg_se,
                                                         const uint32_t* __restrict__ ctg_off, const uint32_t* __restrict__ soff, unsigned long long* total) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_reg) return;
    const uint32_t g0 = ctg_off[reg_ctg[k]];
    atomicAdd(total, (unsigned long long)(soff[g0 + (uint32_t)reg_se[2 * k + 1]] - soff[g0 + (uint32_t)reg_se[2 * k]] + 1));
}

__global__ __launch_bounds__(64) void k_sp_lowdepth(KcCtx cs, KcCtx cl, uint32_t n_grp, uint32_t lanes, const uint32_t* __restrict__ grp_first,
                                                    const uint32_t* __restrict__ reg_ctg, const int32_t* __restrict__ reg_se) {
    const uint32_t k = blockIdx.x * lanes + threadIdx.x;
    if (threadIdx.x < lanes && k < n_grp) sp_lowdepth_group(cs, cl, reg_ctg, reg_se, grp_first[k], grp_first[k + 1]);
}

__global__ __launch_bounds__(256) void k_sp_gather_flags(uint32_t n, const uint32_t* __restrict__ g, const uint32_t* __restrict__ soff, const uint8_t* __restrict__ sflag,
                                                         uint8_t* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = sflag[soff[g[k]]];
}

__global__ __launch_bounds__(256) void k_sp_base_marks(uint32_t G, const uint32_t* __restrict__ soff, const uint8_t* __restrict__ sflag, uint16_t* __restrict__ bmark) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) bmark[g] = sp_base_mark(g, soff, sflag);
}

__global__ __launch_bounds__(256) void k_sp_base_bits(uint64_t G, const uint16_t* __restrict__ bmark, uint32_t mask, unsigned long long* __restrict__ bits) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < G / 64 + 1) bits[w] = sp_base_bits_word(w, G, bmark, mask);
}

// records that can overlap a link region: [r0, r1), and how many 64-record chunks that is
__global__ __launch_bounds__(256) void k_sp_link_ranges(KcCtx c, uint32_t n_reg, const uint32_t* __restrict__ reg_ctg, const int32_t* __restrict__ reg_se,
                                                        unsigned long long* __restrict__ r0_out, uint32_t* __restrict__ n_rec, uint32_t* __restrict__ n_chunk) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_reg) return;
    const uint32_t ct = reg_ctg[k];
    const int64_t rb = (int64_t)c.read_begin[ct], re = (int64_t)c.read_begin[ct + 1];
    const int64_t r0 = kc_lower_bound_pos(c.R, rb, re, reg_se[2 * k] - c.max_span), r1 = kc_lower_bound_pos(c.R, rb, re, reg_se[2 * k + 1] + 1);
    r0_out[k] = (unsigned long long)r0;
    n_rec[k] = (uint32_t)(r1 - r0);
    n_chunk[k] = (uint32_t)((r1 - r0 + 63) / 64);
}

// link regions of all contigs in one flat list, cut into chunks of 64 candidate records; every workgroup takes chunks in turn, one
// record per lane; each lane owns a scratch row (entries + haplotype bytes of one record)
__global__ __launch_bounds__(64) void k_sp_links(KcCtx c, SpParams P, SpLinks L, uint32_t n_reg, uint32_t n_chunks, const uint32_t* __restrict__ reg_ctg,
                                                 const int32_t* __restrict__ reg_se, const uint32_t* __restrict__ reg_idx, const unsigned long long* __restrict__ reg_r0,
                                                 const uint32_t* __restrict__ reg_nrec, const uint32_t* __restrict__ chunk_off, uint32_t level, uint32_t flagbrim, SpEntry* ents,
                                                 uint32_t ecap, uint8_t* bytes, uint32_t bcap) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    SpEntry* my_e = ents + (uint64_t)lane * ecap;
    uint8_t* my_b = bytes + (uint64_t)lane * bcap;
    for (uint32_t w = blockIdx.x; w < n_chunks; w += gridDim.x) {
        uint32_t lo = 0, hi = n_reg;   // last region whose first chunk is <= w
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (chunk_off[mid] <= w) lo = mid; else hi = mid;
        }
        const uint32_t k = lo, t = (w - chunk_off[k]) * 64 + threadIdx.x;
        if (t >= reg_nrec[k]) continue;
        const int64_t r = (int64_t)reg_r0[k] + t;
        const uint32_t ct = reg_ctg[k];
        const int32_t s = reg_se[2 * k], e = reg_se[2 * k + 1];
        if (c.endpos[r] <= s || c.level[r] != level) continue;
        const unsigned long long order = (unsigned long long)flagbrim << 63 | (unsigned long long)reg_idx[k] << 32 | (unsigned long long)(r - (int64_t)c.read_begin[ct]);
        sp_link_record(c, P, L, r, ct, s, e, flagbrim, order, my_e, ecap, my_b, (int32_t)bcap);
    }
}

__global__ __launch_bounds__(256) void k_sp_marks(uint32_t n, const uint32_t* __restrict__ g, const uint8_t* __restrict__ bits, const uint32_t* __restrict__ soff, uint8_t* sflag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) sp_atomic_or8(&sflag[soff[g[k]]], bits[k]);
}

__global__ __launch_bounds__(256) void k_sp_apply(uint32_t n, const uint32_t* __restrict__ g, const int32_t* __restrict__ len, const uint32_t* __restrict__ roff,
                                                  const uint32_t* __restrict__ rstride, const uint8_t* __restrict__ rpool, const int8_t* __restrict__ choice,
                                                  const uint32_t* __restrict__ soff, uint8_t* __restrict__ sbase) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || choice[k] < 0) return;
    const uint8_t* reg = rpool + roff[k] + (uint32_t)choice[k] * rstride[k];
    const uint32_t sb = soff[g[k]];
    if (len[k] == 1) sbase[sb] = reg[0];
    else for (uint32_t s = sb; s < soff[g[k] + 1]; ++s) sbase[s] = reg[s - sb];   // contig_update_contig(pos, pos + 1, region, -1)
}

template <class T>
int upload_vec(DevBuf& b, const std::vector<T>& v, hipStream_t q) {
    if (b.ensure(sizeof(T) * (v.size() + 1))) return -1;
    if (!v.empty()) HIPCHK(npcopy::h2d(b.p, v.data(), sizeof(T) * v.size(), q));
    return 0;
}
template <class T>
int download_vec(std::vector<T>& v, const void* p, size_t n, hipStream_t q) {
    v.resize(n);
    if (n) HIPCHK(npcopy::d2h(v.data(), p, sizeof(T) * n, q));
    HIPCHK(hipStreamSynchronize(q));
    return 0;
}

}  // namespace

// work buffers of the pass (owned by the short-read batch)
enum { W_SOFF1, W_INFO1, W_SBASE1, W_SFLAG1, W_SCOUNT1, W_SOWN1, W_CNT, W_FIRST, W_DEC, W_TOP, W_DIRTY, W_ALLE, W_DPOS, W_SITE_G, W_SITE_CTG, W_SITE_POS,
       W_SITE_LEFT, W_SITE_RIGHT, W_SITE_LEN, W_KEEP, W_RSTRIDE, W_RBYTES, W_ROFF, W_RPOOL, W_MARK, W_MPOS, W_F, W_DOUT, W_DCNT, W_REG_CTG, W_REG_SE,
       W_K_FIRST, W_K_G, W_K_POS, W_K_LEN, W_K_ROFF, W_K_RSTRIDE, W_K_FLAG, W_LK_NUM, W_LK_MQ, W_LK_Q, W_LK_FIRST, W_LK_TOTAL, W_LREG_CTG, W_LREG_SE, W_LREG_IDX,
       W_ENTS, W_BYTES, W_MARK_G, W_MARK_B, W_CHOICE, W_LR_CNT, W_GRP, W_BMARK, W_LREG_R0, W_LREG_NREC, W_LREG_NCH, W_LREG_CHOFF, W_BBITS, W_COUNT };

extern "C" int np1_batch_snp_phase(np1_batch* b, np1_batch* l, const Configure* cfg) {
    if (!b || !l || !cfg) { np1_set_error("snp_phase: null argument"); return -1; }
    if (!b->has_qual || !l->has_qual) { np1_set_error("snp_phase needs both streams loaded with base qualities"); return -1; }
    if (b->nc != l->nc || b->G != l->G || b->ctx != l->ctx) { np1_set_error("snp_phase: the short-read and the long-read batch must hold the same contigs on the same device"); return -1; }
    np1_ctx* ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    hipStream_t q = ctx->stream;
    b->ran = false;
    b->out_cached = false;
    b->out_pinned = false;
    const uint64_t G = b->G;
    const uint32_t nc = b->nc;
    const int64_t n = b->n_reads, nl = l->n_reads;
    if (nc == 0) { b->h_bounds.assign(1, 0); b->S = 0; b->ran = true; return 0; }
    if (b->spw.size() < (size_t)W_COUNT) b->spw.resize(W_COUNT);
    DevBuf* W = b->spw.data();
    const size_t nn = (size_t)(n > 0 ? n : 1), nnl = (size_t)(nl > 0 ? nl : 1);
    if (b->kc_level.ensure(nn) || b->kc_endpos.ensure(4 * nn) || l->kc_level.ensure(nnl) || l->kc_endpos.ensure(4 * nnl) || b->kc_cnt.ensure(4 * KCC_WORDS) ||
        W[W_LR_CNT].ensure(64) || b->ins.ensure(4 * (G + 1)) || b->soff.ensure(4 * (G + 2)) || W[W_SOFF1].ensure(4 * (G + 2)) || b->totals.ensure(64) ||
        b->bounds.ensure(4 * ((size_t)nc + 1)) || b->scan_tmp.ensure(8 * (scan_tmp_words(G + G / 8 + 1024) + scan_tmp_words(nn))) || W[W_DIRTY].ensure(G + 8) ||
        W[W_ALLE].ensure(G + 8) || W[W_DPOS].ensure(4 * (G + 2)) || W[W_DCNT].ensure(8 * ((size_t)nc + 1)))
        return -1;
    uint64_t* totals = b->totals.as<uint64_t>();
    uint32_t* kcnt = b->kc_cnt.as<uint32_t>();
    const uint32_t* ctg_off = b->ctg_off.as<uint32_t>();

    SpParams P;
    P.min_depth_snp = cfg->min_depth_snp; P.min_count_snp = cfg->min_count_snp; P.min_count_snp_link = cfg->min_count_snp_link;
    P.max_variant_count_lgs = cfg->max_variant_count_lgs; P.read_len = cfg->read_len; P.ext_len_edge = cfg->ext_len_edge;
    P.min_snp_factor_sgs = cfg->min_snp_factor_sgs; P.max_clip_ratio_lgs = cfg->max_clip_ratio_lgs; P.rate_lgs = cfg->indel_balance_factor_lgs;
    P.max_indel_factor_lgs = cfg->max_indel_factor_lgs; P.max_snp_factor_lgs = cfg->max_snp_factor_lgs; P.ploidy = cfg->ploidy;
    if (!std::isfinite(P.rate_lgs)) { np1_set_error("indel_balance_factor_lgs is not a finite number"); return -1; }

    auto make_ctx = [&](np1_batch* x) {
        KcCtx c;
        memset(&c, 0, sizeof(c));
        c.R = ReadsDev{x->pos.as<int32_t>(), x->ctg.as<uint32_t>(), x->flag.as<uint16_t>(), x->ncig.as<uint32_t>(), x->lq.as<int32_t>(),
                       x->cigoff.as<uint64_t>(), x->seqoff.as<uint64_t>(), x->cigar.as<uint32_t>(), x->seq.as<uint8_t>()};
        c.mapq = x->mapq.as<uint8_t>(); c.isize = x->isize.as<int32_t>(); c.qual_off = x->qualoff.as<uint64_t>(); c.qual = x->qual.as<uint8_t>();
        c.level = x->kc_level.as<uint8_t>(); c.endpos = x->kc_endpos.as<int32_t>();
        c.ctg_off = ctg_off; c.read_begin = x->read_begin.as<uint64_t>();
        c.trim = cfg->trim_len_edge; c.ext_len_edge = cfg->ext_len_edge; c.min_map_quality = cfg->min_map_quality; c.read_tlen = cfg->read_tlen;
        c.max_clip_ratio_sgs = cfg->max_clip_ratio_sgs; c.min_count_ratio_skip = cfg->min_count_ratio_skip;
        c.K = -1; c.Rfix = 0; c.rate = cfg->indel_balance_factor_lgs;
        c.third_rule = 1; c.max_indel_factor_lgs = cfg->max_indel_factor_lgs; c.max_snp_factor_lgs = cfg->max_snp_factor_lgs;
        c.err = &kcnt[KCC_ERR];
        return c;
    };
    KcCtx cs = make_ctx(b), cl = make_ctx(l);
    cl.keep_zero_marks = 1;
    static const bool hist_lds = [] { const char* e = getenv("NP1_SP_HIST"); return !(e && strcmp(e, "global") == 0); }();   // NP1_SP_HIST=global: every vote straight to HBM
    static const uint32_t site_lanes = [] { const char* e = getenv("NP1_SP_LANES"); const int x = e ? atoi(e) : 8; return (uint32_t)(x < 1 ? 1 : x > 64 ? 64 : x); }();

    for (int attempt = 0; attempt < 5; ++attempt) {
        const size_t scale = (size_t)1 << (2 * attempt);
        uint32_t hk[KCC_WORDS];
        HIPCHK(hipMemsetAsync(kcnt, 0, 4 * KCC_WORDS, q));
        HIPCHK(hipMemsetAsync(W[W_LR_CNT].p, 0, 64, q));
        HIPCHK(hipMemsetAsync(b->totals.p, 0, 64, q));
        // ---- records of both streams
        kc_launch_records(q, cs, n, b->kc_level.as<uint8_t>(), b->kc_endpos.as<int32_t>(), &kcnt[KCC_MAXSPAN]);
        k_sp_lr_records<<<nblk((uint64_t)nnl, 256), 256, 0, q>>>(cl.R, nl, P.max_clip_ratio_lgs, l->kc_level.as<uint8_t>(), l->kc_endpos.as<int32_t>(), W[W_LR_CNT].as<uint32_t>());
        // ---- P1: short-read columns, first slot space
        HIPCHK(hipMemsetAsync(b->ins.p, 0, 4 * (G + 1), q));
        k_sp_insert<<<nblk((uint64_t)nn, 256), 256, 0, q>>>(cs.R, n, b->kc_level.as<uint8_t>(), ctg_off, b->ins.as<uint32_t>(), 0u, nullptr, nullptr);
        uint64_t* scan_tmp = b->scan_tmp.as<uint64_t>();
        launch_scan_slots(q, b->ins.as<uint32_t>(), G, W[W_SOFF1].as<uint32_t>(), scan_tmp, &totals[1]);
        uint64_t S64 = 0;
        uint32_t lr_span = 0;
        HIPCHK(npcopy::d2h(&S64, &totals[1], 8, q));
        HIPCHK(npcopy::d2h(hk, kcnt, sizeof(hk), q));
        HIPCHK(npcopy::d2h(&lr_span, W[W_LR_CNT].p, 4, q));
        HIPCHK(hipStreamSynchronize(q));
        if (S64 >= 0x0ffffff0ull) { np1_set_error("snp_phase: batch too large (more than 2^28 slots; the histogram is 128 bytes per slot)"); return -1; }
        const uint32_t S1 = (uint32_t)S64;
        cs.max_span = (int32_t)(hk[KCC_MAXSPAN] ? hk[KCC_MAXSPAN] : 1);
        cl.max_span = (int32_t)(lr_span ? lr_span : 1);
        if (W[W_INFO1].ensure(S1 + 64) || W[W_SBASE1].ensure(S1 + 64) || W[W_SFLAG1].ensure(S1 + 64) || W[W_SCOUNT1].ensure(2 * ((size_t)S1 + 64)) ||
            W[W_SOWN1].ensure(4 * ((size_t)S1 + 64)) || W[W_CNT].ensure(64ull * ((size_t)S1 + 1)) || W[W_FIRST].ensure(64ull * ((size_t)S1 + 1)) || W[W_DEC].ensure(S1 + 64) ||
            W[W_TOP].ensure(S1 + 64) || W[W_MARK].ensure(S1 + 64) || W[W_MPOS].ensure(4 * ((size_t)S1 + 2)) || b->kc_lhead.ensure(4 * ((size_t)S1 + 64)))
            return -1;
        if (scan_tmp_words((uint64_t)S1 + 1) * 8 > b->scan_tmp.cap && b->scan_tmp.ensure(8 * (scan_tmp_words((uint64_t)S1 + 1) + scan_tmp_words(nn)))) return -1;
        scan_tmp = b->scan_tmp.as<uint64_t>();
        launch_slotinfo(q, b->draft.as<uint8_t>(), (uint32_t)G, ctg_off, nc, W[W_SOFF1].as<uint32_t>(), W[W_INFO1].as<uint8_t>(), W[W_SOWN1].as<uint32_t>());
        kc_launch_slots(q, W[W_INFO1].as<uint8_t>(), S1, W[W_SBASE1].as<uint8_t>(), W[W_SFLAG1].as<uint8_t>(), W[W_SCOUNT1].as<uint16_t>(), b->kc_lhead.as<uint32_t>());
        // ---- P2 / P3: base histogram of the level-2 short reads, per-slot verdicts
        HIPCHK(hipMemsetAsync(W[W_CNT].p, 0, 64ull * S1, q));
        HIPCHK(hipMemsetAsync(W[W_FIRST].p, 0xff, 64ull * S1, q));
        cs.soff = W[W_SOFF1].as<uint32_t>();
        if (hist_lds && n > 0) k_sp_hist_lds<<<nblk((uint64_t)n, 128), 128, 0, q>>>(cs, n, W[W_CNT].as<uint32_t>(), W[W_FIRST].as<uint32_t>());
        else k_sp_hist<<<nblk((uint64_t)nn, 256), 256, 0, q>>>(cs, n, W[W_CNT].as<uint32_t>(), W[W_FIRST].as<uint32_t>());
        k_sp_decide<<<nblk(S1, 256), 256, 0, q>>>(P, S1, W[W_CNT].as<uint32_t>(), W[W_FIRST].as<uint32_t>(), W[W_SBASE1].as<uint8_t>(), W[W_SFLAG1].as<uint8_t>(),
                                                  W[W_SCOUNT1].as<uint16_t>(), W[W_DEC].as<uint8_t>(), W[W_TOP].as<uint8_t>(), &kcnt[KCC_ERR]);
        // ---- P4: sites
        k_sp_sites<<<nblk(G, 256), 256, 0, q>>>((uint32_t)G, ctg_off, nc, W[W_SOFF1].as<uint32_t>(), W[W_DEC].as<uint8_t>(), W[W_TOP].as<uint8_t>(), W[W_SFLAG1].as<uint8_t>(),
                                                W[W_DIRTY].as<uint8_t>(), W[W_ALLE].as<uint8_t>());
        launch_scan_u8(q, W[W_DIRTY].as<uint8_t>(), G, W[W_DPOS].as<uint32_t>(), scan_tmp, &totals[2]);
        uint64_t NS64 = 0;
        HIPCHK(npcopy::d2h(&NS64, &totals[2], 8, q));
        HIPCHK(hipStreamSynchronize(q));
        const uint32_t NS = (uint32_t)NS64;
        const size_t ns1 = (size_t)NS + 1;
        if (W[W_SITE_G].ensure(4 * ns1) || W[W_SITE_CTG].ensure(4 * ns1) || W[W_SITE_POS].ensure(4 * ns1) || W[W_SITE_LEFT].ensure(4 * ns1) || W[W_SITE_RIGHT].ensure(4 * ns1) ||
            W[W_SITE_LEN].ensure(4 * ns1) || W[W_KEEP].ensure(ns1 + 8) || W[W_RSTRIDE].ensure(4 * ns1) || W[W_RBYTES].ensure(4 * (ns1 + 1)) || W[W_ROFF].ensure(4 * (ns1 + 1)))
            return -1;
        if (NS) {
            k_sp_site_compact<<<nblk(G, 256), 256, 0, q>>>((uint32_t)G, ctg_off, nc, W[W_DIRTY].as<uint8_t>(), W[W_DPOS].as<uint32_t>(), W[W_SITE_G].as<uint32_t>(),
                                                           W[W_SITE_CTG].as<uint32_t>());
            k_sp_anchors<<<nblk(NS, 256), 256, 0, q>>>(NS, W[W_SITE_G].as<uint32_t>(), W[W_SITE_CTG].as<uint32_t>(), ctg_off, W[W_DIRTY].as<uint8_t>(), W[W_SITE_POS].as<int32_t>(),
                                                       W[W_SITE_LEFT].as<int32_t>(), W[W_SITE_RIGHT].as<int32_t>());
        }
        // ---- P5: low-depth regions over the slots, INSERT marks
        k_sp_depthmark<<<nblk(S1, 256), 256, 0, q>>>(S1, W[W_SFLAG1].as<uint8_t>(), W[W_MARK].as<uint8_t>());
        launch_scan_u8(q, W[W_MARK].as<uint8_t>(), S1, W[W_MPOS].as<uint32_t>(), scan_tmp, &totals[3]);
        uint64_t M = 0;
        HIPCHK(npcopy::d2h(&M, &totals[3], 8, q));
        HIPCHK(hipStreamSynchronize(q));
        HIPCHK(hipMemcpyAsync(W[W_MPOS].as<uint32_t>() + S1, &totals[3], 4, hipMemcpyDeviceToDevice, q));   // mpos[S1] = M
        if (W[W_F].ensure(4 * (M + 4)) || W[W_DOUT].ensure(8 * (M + (size_t)nc + 4))) return -1;
        k_sp_compact_slots<<<nblk(S1, 256), 256, 0, q>>>(S1, W[W_MARK].as<uint8_t>(), W[W_MPOS].as<uint32_t>(), W[W_F].as<uint32_t>());
        k_sp_depth<<<nblk(nc, 64), 64, 0, q>>>(nc, ctg_off, W[W_SOFF1].as<uint32_t>(), W[W_SOWN1].as<uint32_t>(), W[W_MPOS].as<uint32_t>(), W[W_F].as<uint32_t>(),
                                               (uint32_t)P.ext_len_edge, P.ext_len_edge, W[W_DOUT].as<int32_t>(), W[W_DCNT].as<uint32_t>(), W[W_DCNT].as<uint32_t>() + nc, &kcnt[KCC_ERR]);
        std::vector<uint32_t> dcnt;
        if (download_vec(dcnt, W[W_DCNT].p, 2 * (size_t)nc, q)) return -1;
        b->h_ctg_off.resize((size_t)nc + 1);
        HIPCHK(npcopy::d2h(b->h_ctg_off.data(), b->ctg_off.p, 4 * ((size_t)nc + 1), q));
        HIPCHK(hipStreamSynchronize(q));
        std::vector<uint32_t> reg_ctg;
        std::vector<int32_t> reg_se;
        for (uint32_t ct = 0; ct < nc; ++ct) {
            if (!dcnt[ct]) continue;
            const size_t at = reg_se.size();
            reg_se.resize(at + dcnt[ct]);
            HIPCHK(npcopy::d2h(reg_se.data() + at, W[W_DOUT].as<int32_t>() + 2ull * ((uint64_t)dcnt[nc + ct] + ct), 4ull * dcnt[ct], q));
            for (uint32_t i = 0; i + 1 < dcnt[ct]; i += 2) reg_ctg.push_back(ct);
        }
        HIPCHK(hipStreamSynchronize(q));
        const uint32_t n_reg = (uint32_t)reg_ctg.size();
        const std::vector<uint32_t> grp_first = sp_region_groups(reg_ctg, reg_se);
        const uint32_t n_grp = (uint32_t)grp_first.size() - 1;
        if (upload_vec(W[W_REG_CTG], reg_ctg, q) || upload_vec(W[W_REG_SE], reg_se, q) || upload_vec(W[W_GRP], grp_first, q)) return -1;
        if (n_reg)
            k_sp_mark_regions<<<nblk(n_reg, 256), 256, 0, q>>>(n_reg, W[W_REG_CTG].as<uint32_t>(), W[W_REG_SE].as<int32_t>(), ctg_off, W[W_SOFF1].as<uint32_t>(), W[W_SFLAG1].as<uint8_t>());
        // ---- P6: long-read columns behind marked bases, second slot space
        k_sp_insert<<<nblk((uint64_t)nnl, 256), 256, 0, q>>>(cl.R, nl, l->kc_level.as<uint8_t>(), ctg_off, b->ins.as<uint32_t>(), F_INSERT | F_SNP, W[W_SOFF1].as<uint32_t>(),
                                                             W[W_SFLAG1].as<uint8_t>());
        launch_scan_slots(q, b->ins.as<uint32_t>(), G, b->soff.as<uint32_t>(), scan_tmp, &totals[4]);
        HIPCHK(npcopy::d2h(&S64, &totals[4], 8, q));
        HIPCHK(hipStreamSynchronize(q));
        if (S64 >= 0xfffffff0ull) { np1_set_error("batch too large: more than 2^32 slots"); return -1; }
        const uint32_t S = (uint32_t)S64;
        b->S = S;
        if (b->slot_info.ensure(S + 64) || b->slot_res.ensure(2 * ((size_t)S + 64)) || b->opos.ensure(4 * ((size_t)S + 2)) || b->out.ensure((size_t)S + 64) ||
            b->kc_sbase.ensure(S + 64) || b->kc_sflag.ensure(S + 64) || b->kc_srefk.ensure(2 * ((size_t)S + 64)) || b->kc_scount.ensure(2 * ((size_t)S + 64)) ||
            b->kc_lhead.ensure(4 * ((size_t)S + 64)) || b->slot_g.ensure(4 * ((size_t)S + 64)))
            return -1;
        if (scan_tmp_words((uint64_t)S + 1) * 8 > b->scan_tmp.cap && b->scan_tmp.ensure(8 * (scan_tmp_words((uint64_t)S + 1) + scan_tmp_words(nn)))) return -1;
        scan_tmp = b->scan_tmp.as<uint64_t>();
        k_sp_reslot<<<nblk(G, 256), 256, 0, q>>>((uint32_t)G, W[W_SOFF1].as<uint32_t>(), b->soff.as<uint32_t>(), W[W_SBASE1].as<uint8_t>(), W[W_SFLAG1].as<uint8_t>(),
                                                 W[W_SCOUNT1].as<uint16_t>(), b->kc_sbase.as<uint8_t>(), b->kc_sflag.as<uint8_t>(), b->kc_scount.as<uint16_t>(), b->slot_g.as<uint32_t>());
        launch_slotinfo(q, b->draft.as<uint8_t>(), (uint32_t)G, ctg_off, nc, b->soff.as<uint32_t>(), b->slot_info.as<uint8_t>(), nullptr);
        HIPCHK(hipMemsetAsync(b->kc_lhead.p, 0, 4 * ((size_t)S + 64), q));
        // pools of the site verdicts and of the low-depth chains
        unsigned long long nd_slots = 0;
        if (n_reg) {
            HIPCHK(hipMemsetAsync(&totals[5], 0, 8, q));
            k_sp_region_slots<<<nblk(n_reg, 256), 256, 0, q>>>(n_reg, W[W_REG_CTG].as<uint32_t>(), W[W_REG_SE].as<int32_t>(), ctg_off, b->soff.as<uint32_t>(),
                                                               reinterpret_cast<unsigned long long*>(&totals[5]));
            HIPCHK(npcopy::d2h(&nd_slots, &totals[5], 8, q));
            HIPCHK(hipStreamSynchronize(q));
        }
        const size_t lcap = std::min<size_t>(((size_t)64 * nd_slots + ((size_t)1 << 20)) * scale, (size_t)0x7ffffff0u);
        const size_t stcap = std::min<size_t>((4 * nd_slots + 64ull * n_reg + 4096) * scale, (size_t)0x0ffffff0u);
        if (W[W_BMARK].ensure(2 * (G + 2)) || W[W_BBITS].ensure(8 * (G / 64 + 2))) return -1;
        const size_t hcap = std::min<size_t>(((size_t)16384 * NS + ((size_t)64 << 20)) * scale, (size_t)0xfffffff0u);
        if (b->kc_lpool.ensure(8 * lcap) || b->kc_stsc.ensure(8 * 16 * stcap) || b->kc_stkm.ensure(2 * 16 * stcap) || b->kc_strk.ensure(16 * stcap) || b->kc_hpool.ensure(hcap)) return -1;
        for (KcCtx* c : {&cs, &cl}) {
            c->soff = b->soff.as<uint32_t>(); c->sbase = b->kc_sbase.as<uint8_t>(); c->sflag = b->kc_sflag.as<uint8_t>(); c->srefk = b->kc_srefk.as<uint16_t>();
            c->scount = b->kc_scount.as<uint16_t>(); c->lhead = b->kc_lhead.as<uint32_t>(); c->lpool = b->kc_lpool.as<uint32_t>(); c->lcap = (uint32_t)lcap;
            c->lcount = &kcnt[KCC_LCOUNT]; c->st_score = b->kc_stsc.as<long long>(); c->st_kmer = b->kc_stkm.as<uint16_t>(); c->st_rank = b->kc_strk.as<uint8_t>();
            c->st_cap = (uint32_t)stcap; c->st_count = &kcnt[KCC_STCOUNT]; c->hpool = b->kc_hpool.as<uint8_t>(); c->hcap = (uint32_t)hcap; c->hcount = &kcnt[KCC_HCOUNT];
            c->sown = b->slot_g.as<uint32_t>();
            c->bmark = W[W_BMARK].as<uint16_t>();
            c->bbits = W[W_BBITS].as<unsigned long long>();
        }
        // ---- P7: site verdicts
        uint64_t RB = 0;
        if (NS) {
            k_sp_site_stride<<<nblk(NS, 256), 256, 0, q>>>(NS, W[W_SITE_G].as<uint32_t>(), b->soff.as<uint32_t>(), W[W_RSTRIDE].as<uint32_t>(), W[W_RBYTES].as<uint32_t>(),
                                                           W[W_SITE_LEN].as<int32_t>(), W[W_KEEP].as<uint8_t>());
            launch_scan_u32(q, W[W_RBYTES].as<uint32_t>(), NS, W[W_ROFF].as<uint32_t>(), scan_tmp, &totals[6]);
            HIPCHK(npcopy::d2h(&RB, &totals[6], 8, q));
            HIPCHK(hipStreamSynchronize(q));
            if (RB >= 0xfffffff0ull) { np1_set_error("snp_phase: allele strings of the sites exceed 4 GB"); return -1; }
            if (W[W_RPOOL].ensure(RB + 64)) return -1;
            HIPCHK(hipMemsetAsync(W[W_RPOOL].p, 0, RB + 64, q));
            k_sp_site_init<<<nblk(NS, 256), 256, 0, q>>>(NS, W[W_SITE_G].as<uint32_t>(), W[W_ALLE].as<uint8_t>(), W[W_ROFF].as<uint32_t>(), W[W_RSTRIDE].as<uint32_t>(),
                                                         W[W_RPOOL].as<uint8_t>());
            SpSites SS{W[W_SITE_G].as<uint32_t>(), W[W_SITE_CTG].as<uint32_t>(), W[W_SITE_LEFT].as<int32_t>(), W[W_SITE_RIGHT].as<int32_t>(), W[W_SITE_LEN].as<int32_t>(),
                       W[W_KEEP].as<uint8_t>(), W[W_ROFF].as<uint32_t>(), W[W_RSTRIDE].as<uint32_t>(), W[W_RPOOL].as<uint8_t>()};
            k_sp_verdict<<<nblk(NS, site_lanes), 64, 0, q>>>(cs, cl, P, SS, NS, site_lanes, W[W_SOFF1].as<uint32_t>(), W[W_CNT].as<uint32_t>(), W[W_FIRST].as<uint32_t>());
        }
        // ---- P9: low-depth regions, both streams
        if (n_reg) k_sp_lowdepth<<<nblk(n_grp, site_lanes), 64, 0, q>>>(cs, cl, n_grp, site_lanes, W[W_GRP].as<uint32_t>(), W[W_REG_CTG].as<uint32_t>(), W[W_REG_SE].as<int32_t>());
        HIPCHK(npcopy::d2h(hk, kcnt, sizeof(hk), q));
        HIPCHK(hipStreamSynchronize(q));
        if (hk[KCC_ERR] & (ERR_KC_POOL | ERR_SP_POOL)) continue;
        // ---- kept sites (host lays out the list), links
        std::vector<uint8_t> keep;
        std::vector<uint32_t> site_g, site_ctg, roff, rstride;
        std::vector<int32_t> site_pos, site_left, site_right, site_len;
        if (download_vec(keep, W[W_KEEP].p, NS, q) || download_vec(site_g, W[W_SITE_G].p, NS, q) || download_vec(site_ctg, W[W_SITE_CTG].p, NS, q) ||
            download_vec(site_pos, W[W_SITE_POS].p, NS, q) || download_vec(site_left, W[W_SITE_LEFT].p, NS, q) || download_vec(site_right, W[W_SITE_RIGHT].p, NS, q) ||
            download_vec(site_len, W[W_SITE_LEN].p, NS, q) || download_vec(roff, W[W_ROFF].p, NS, q) || download_vec(rstride, W[W_RSTRIDE].p, NS, q))
            return -1;
        bool undefined = (hk[KCC_ERR] & ERR_SP_UNDEFINED) != 0;
        std::vector<uint32_t> k_first((size_t)nc + 1, 0), k_g, k_roff, k_rstride, k_ctg;
        std::vector<int32_t> k_pos, k_len, k_left, k_right;
        {
            uint32_t k = 0;
            for (uint32_t ct = 0; ct < nc; ++ct) {
                k_first[ct] = (uint32_t)k_g.size();
                for (; k < NS && site_ctg[k] == ct; ++k)
                    if (keep[k]) {
                        k_g.push_back(site_g[k]); k_pos.push_back(site_pos[k]); k_len.push_back((int32_t)(int16_t)site_len[k]); k_roff.push_back(roff[k]);
                        k_rstride.push_back(rstride[k]); k_left.push_back(site_left[k]); k_right.push_back(site_right[k]); k_ctg.push_back(ct);
                    }
            }
            k_first[nc] = (uint32_t)k_g.size();
        }
        const uint32_t NK = (uint32_t)k_g.size();
        std::vector<int8_t> choice(NK, -1);
        if (NK && !undefined) {
            if (upload_vec(W[W_K_FIRST], k_first, q) || upload_vec(W[W_K_G], k_g, q) || upload_vec(W[W_K_POS], k_pos, q) || upload_vec(W[W_K_LEN], k_len, q) ||
                upload_vec(W[W_K_ROFF], k_roff, q) || upload_vec(W[W_K_RSTRIDE], k_rstride, q) || W[W_K_FLAG].ensure(NK + 8) || W[W_LK_NUM].ensure(16ull * NK + 16) ||
                W[W_LK_MQ].ensure(16ull * NK + 16) || W[W_LK_Q].ensure(16ull * NK + 16) || W[W_LK_FIRST].ensure(32ull * NK + 32) || W[W_LK_TOTAL].ensure(4ull * NK + 16))
                return -1;
            HIPCHK(hipMemsetAsync(W[W_LK_NUM].p, 0, 16ull * NK, q));
            HIPCHK(hipMemsetAsync(W[W_LK_MQ].p, 0, 16ull * NK, q));
            HIPCHK(hipMemsetAsync(W[W_LK_Q].p, 0, 16ull * NK, q));
            HIPCHK(hipMemsetAsync(W[W_LK_FIRST].p, 0xff, 32ull * NK, q));
            HIPCHK(hipMemsetAsync(W[W_LK_TOTAL].p, 0, 4ull * NK, q));
            SpLinks LK{W[W_K_FIRST].as<uint32_t>(), W[W_K_POS].as<int32_t>(), W[W_K_LEN].as<int32_t>(), W[W_K_ROFF].as<uint32_t>(), W[W_K_RSTRIDE].as<uint32_t>(),
                       W[W_RPOOL].as<uint8_t>(), W[W_LK_NUM].as<int32_t>(), W[W_LK_MQ].as<int32_t>(), W[W_LK_Q].as<int32_t>(),
                       W[W_LK_FIRST].as<unsigned long long>(), W[W_LK_TOTAL].as<int32_t>()};
            std::vector<uint8_t> k_flag;
            std::vector<int32_t> lk_num, lk_mq, lk_q, lk_total;
            std::vector<unsigned long long> lk_first;
            auto fetch_flags = [&]() -> int {
                k_sp_gather_flags<<<nblk(NK, 256), 256, 0, q>>>(NK, W[W_K_G].as<uint32_t>(), b->soff.as<uint32_t>(), b->kc_sflag.as<uint8_t>(), W[W_K_FLAG].as<uint8_t>());
                return download_vec(k_flag, W[W_K_FLAG].p, NK, q);
            };
            auto fetch_links = [&]() -> int {
                return download_vec(lk_num, W[W_LK_NUM].p, 4ull * NK, q) || download_vec(lk_mq, W[W_LK_MQ].p, 4ull * NK, q) || download_vec(lk_q, W[W_LK_Q].p, 4ull * NK, q) ||
                       download_vec(lk_first, W[W_LK_FIRST].p, 4ull * NK, q) || download_vec(lk_total, W[W_LK_TOTAL].p, NK, q);
            };
            auto host_sites = [&](uint32_t ct) {
                std::vector<SpHostSite> v;
                for (uint32_t kk = k_first[ct]; kk < k_first[ct + 1]; ++kk) {
                    SpHostSite h;
                    memset(&h, 0, sizeof(h));
                    h.pos = k_pos[kk]; h.left = k_left[kk]; h.right = k_right[kk]; h.len = k_len[kk]; h.flag = k_flag[kk];
                    if (!lk_total.empty()) {
                        h.total = lk_total[kk];
                        for (int t = 0; t < 4; ++t) { h.num[t] = lk_num[4ull * kk + t]; h.mapqual[t] = lk_mq[4ull * kk + t]; h.qual[t] = lk_q[4ull * kk + t]; h.first[t] = lk_first[4ull * kk + t]; }
                    }
                    v.push_back(h);
                }
                return v;
            };
            auto run_links = [&](const KcCtx& c, int phase) -> int {   // phase 0: short reads over groups of near sites; 1: long reads between the marks
                std::vector<uint32_t> lctg, lidx;
                std::vector<int32_t> lse;
                for (uint32_t ct = 0; ct < nc; ++ct) {
                    if (k_first[ct + 1] - k_first[ct] <= 1) continue;
                    const std::vector<SpHostSite> h = host_sites(ct);
                    const std::vector<int32_t> reg = phase == 0 ? sp_link_regions(h, P.read_len, F_SNP) : sp_link_regions(h, P.max_variant_count_lgs, 0);
                    for (size_t i = 0; i + 1 < reg.size(); i += 2) { lctg.push_back(ct); lse.push_back(reg[i]); lse.push_back(reg[i + 1]); lidx.push_back((uint32_t)(i / 2)); }
                }
                const uint32_t nlr = (uint32_t)lctg.size();
                if (!nlr) return 0;
                // scratch row of a lane: one entry per marked base the record passes, one byte per base / column inside open strings; never more
                // than the reference's own buffer (max_variant_count_lgs); rows grow with the attempt when a record needs more
                const uint32_t lq_max = std::max<uint32_t>(phase == 0 ? b->max_lq : l->max_lq, 256u);
                const uint32_t ecap = (uint32_t)std::min<size_t>((phase == 0 ? 512u : 2048u) * scale, 65536u);
                const uint32_t bcap = (uint32_t)std::min<size_t>(((size_t)2 * lq_max + 4096u) * scale, (size_t)P.max_variant_count_lgs + 4096u);
                if (upload_vec(W[W_LREG_CTG], lctg, q) || upload_vec(W[W_LREG_SE], lse, q) || upload_vec(W[W_LREG_IDX], lidx, q) || W[W_LREG_R0].ensure(8ull * nlr + 8) ||
                    W[W_LREG_NREC].ensure(4ull * nlr + 8) || W[W_LREG_NCH].ensure(4ull * nlr + 8) || W[W_LREG_CHOFF].ensure(4ull * nlr + 8))
                    return -1;
                k_sp_base_marks<<<nblk(G, 256), 256, 0, q>>>((uint32_t)G, b->soff.as<uint32_t>(), b->kc_sflag.as<uint8_t>(), W[W_BMARK].as<uint16_t>());
                k_sp_base_bits<<<nblk(G / 64 + 1, 256), 256, 0, q>>>(G, W[W_BMARK].as<uint16_t>(), phase == 0 ? F_SNP : (F_LEFT | F_RIGHT), W[W_BBITS].as<unsigned long long>());
                k_sp_link_ranges<<<nblk(nlr, 256), 256, 0, q>>>(c, nlr, W[W_LREG_CTG].as<uint32_t>(), W[W_LREG_SE].as<int32_t>(), W[W_LREG_R0].as<unsigned long long>(),
                                                                W[W_LREG_NREC].as<uint32_t>(), W[W_LREG_NCH].as<uint32_t>());
                launch_scan_u32(q, W[W_LREG_NCH].as<uint32_t>(), nlr, W[W_LREG_CHOFF].as<uint32_t>(), scan_tmp, &totals[5]);
                uint64_t n_chunks = 0;
                HIPCHK(npcopy::d2h(&n_chunks, &totals[5], 8, q));
                HIPCHK(hipStreamSynchronize(q));
                if (!n_chunks) return 0;
                const uint32_t blocks = (uint32_t)std::min<uint64_t>(n_chunks, phase == 0 ? 1024u : 512u);
                if (W[W_ENTS].ensure(sizeof(SpEntry) * (size_t)ecap * blocks * 64) || W[W_BYTES].ensure((size_t)bcap * blocks * 64)) return -1;
                k_sp_links<<<blocks, 64, 0, q>>>(c, P, LK, nlr, (uint32_t)n_chunks, W[W_LREG_CTG].as<uint32_t>(), W[W_LREG_SE].as<int32_t>(), W[W_LREG_IDX].as<uint32_t>(),
                                                 W[W_LREG_R0].as<unsigned long long>(), W[W_LREG_NREC].as<uint32_t>(), W[W_LREG_CHOFF].as<uint32_t>(), phase == 0 ? 2u : 1u,
                                                 (uint32_t)phase, W[W_ENTS].as<SpEntry>(), ecap, W[W_BYTES].as<uint8_t>(), bcap);
                return 0;
            };
            if (fetch_flags() || run_links(cs, 0) || fetch_links()) return -1;
            // the marks between weakly linked neighbours (snpphase.c:383-393)
            std::vector<uint32_t> mark_g;
            std::vector<uint8_t> mark_b;
            for (uint32_t ct = 0; ct < nc; ++ct) {
                const std::vector<SpHostSite> h = host_sites(ct);
                for (auto& m : sp_link_marks(h, P.min_count_snp_link)) { mark_g.push_back(b->h_ctg_off[ct] + (uint32_t)m.first); mark_b.push_back(m.second); }
            }
            if (!mark_g.empty()) {
                if (upload_vec(W[W_MARK_G], mark_g, q) || upload_vec(W[W_MARK_B], mark_b, q)) return -1;
                k_sp_marks<<<nblk(mark_g.size(), 256), 256, 0, q>>>((uint32_t)mark_g.size(), W[W_MARK_G].as<uint32_t>(), W[W_MARK_B].as<uint8_t>(), b->soff.as<uint32_t>(),
                                                                    b->kc_sflag.as<uint8_t>());
            }
            if (fetch_flags() || run_links(cl, 1) || fetch_links()) return -1;
            HIPCHK(npcopy::d2h(hk, kcnt, sizeof(hk), q));
            HIPCHK(hipStreamSynchronize(q));
            undefined = undefined || (hk[KCC_ERR] & ERR_SP_UNDEFINED) != 0;
            // ---- the chain over the sites of every contig, then the writes
            for (uint32_t ct = 0; ct < nc && !undefined; ++ct) {
                const std::vector<SpHostSite> h = host_sites(ct);
                std::vector<int8_t> ch;
                if (!sp_chain(h, P.ploidy, &ch)) { undefined = true; break; }
                for (size_t t = 0; t < ch.size(); ++t) choice[k_first[ct] + t] = ch[t];
            }
            if (!undefined) {
                if (upload_vec(W[W_CHOICE], choice, q)) return -1;
                k_sp_apply<<<nblk(NK, 256), 256, 0, q>>>(NK, W[W_K_G].as<uint32_t>(), W[W_K_LEN].as<int32_t>(), W[W_K_ROFF].as<uint32_t>(), W[W_K_RSTRIDE].as<uint32_t>(),
                                                         W[W_RPOOL].as<uint8_t>(), W[W_CHOICE].as<int8_t>(), b->soff.as<uint32_t>(), b->kc_sbase.as<uint8_t>());
            }
        }
        if (undefined) {
            np1_set_error("snp_phase: the reference reads through a null or unset pointer for this input (a site with insertion columns that no filtered read spans, "
                          "or a long-read insertion behind an open site string; snpphase.c:269,750) and has no defined result");
            return -1;
        }
        // ---- emit with the FLAG_THIRD marks as lower case (snpphase.c:129)
        kc_launch_result(q, b->kc_sbase.as<uint8_t>(), b->kc_sflag.as<uint8_t>(), S, b->slot_res.as<uint16_t>());
        launch_scan_keep(q, b->slot_res.as<uint16_t>(), S, b->opos.as<uint32_t>(), scan_tmp, &totals[7]);
        launch_emit(q, b->slot_res.as<uint16_t>(), b->slot_info.as<uint8_t>(), b->opos.as<uint32_t>(), S, F_THIRD, b->out.as<uint8_t>());
        launch_contig_bounds(q, ctg_off, nc, b->soff.as<uint32_t>(), b->opos.as<uint32_t>(), b->bounds.as<uint32_t>());
        b->h_bounds.resize((size_t)nc + 1);
        HIPCHK(npcopy::d2h(b->h_bounds.data(), b->bounds.p, 4 * ((size_t)nc + 1), q));
        HIPCHK(npcopy::d2h(hk, kcnt, sizeof(hk), q));
        HIPCHK(hipStreamSynchronize(q));
        if (hk[KCC_ERR] & (ERR_KC_POOL | ERR_SP_POOL)) continue;
        if (hk[KCC_ERR] & ERR_SP_DEPTH) { np1_set_error("snp_phase: more than 65535 votes on one slot (the reference's 16-bit counters wrap there)"); return -1; }
        if (hk[KCC_ERR]) { np1_set_error("snp_phase: inconsistent pileup or region overflow on the device"); return -1; }
        b->votes = 0;
        b->ran = true;
        return 0;
    }
    np1_set_error("snp_phase: scratch pools keep overflowing");
    return -1;
}
