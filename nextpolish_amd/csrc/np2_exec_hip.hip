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
//   k2_dp        one lane                  chain DP, column after column (sequential in the reference too)  [get_cns_from_align_tags]
//   k2_backtrace one lane                  best path -> consensus bases                       [generate_cns_from_best_score]
// Bounds: tags/links/build stream every alignment column once (HBM); k2_dp is latency bound -- the (A, C)
// run decomposition described in DESIGN.md is the next step for it.
#include <hip/hip_runtime.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/nextpolish2.h"
#include "np2_exec.h"

namespace np2 {
namespace {

using namespace np2k;

#define HIPOK(x)                                                                                        \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) { *err = std::string(#x) + ": " + hipGetErrorString(e_); return false; }  \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap && p) return true;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return false; }
        cap = want;
        return true;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T* as() const { return static_cast<T*>(p); }
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

struct SpanOut {
    uint32_t col0, aln_len, aln_t_s, aln_t_e;
    uint32_t bad, pad0, pad1, pad2;
};

__global__ void k2_span(const int32_t* pos, const uint32_t* n_cigar, const uint64_t* cigar_off, const uint64_t* seq_off,
                        const uint32_t* cigar, const uint8_t* seq, uint32_t n, const char* contig_seq, int32_t s, int32_t e,
                        SpanOut* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ReadView rv{pos[i], n_cigar[i], cigar + cigar_off[i], seq + seq_off[i]};
    uint32_t N, rf_len, rd_len;
    bool bad;
    cigar_totals(rv, &N, &rf_len, &rd_len, &bad);
    SpanOut o{0, 0, 0, 0, bad ? 1u : 0u, 0, 0, 0};
    if (!bad) {
        const AlnSpan a = align_span(rv, contig_seq, s, e);
        o.col0 = a.col0; o.aln_len = a.aln_len; o.aln_t_s = a.aln_t_s; o.aln_t_e = a.aln_t_e;
    }
    out[i] = o;
}

// the seed stream: window against itself, two columns per byte; coverage +1 on every column
__global__ void k2_seed_tags(const char* contig_seq, int32_t s, uint32_t l, uint8_t* tags, uint32_t* coverage, uint32_t* max_size) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;   // tag byte
    const uint32_t nbytes = (l + 1) / 2 + 1;
    if (b >= nbytes) return;
    uint32_t v = 0;
    for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t p = 2 * b + h;
        uint32_t nib;
        if (p < l) {
            nib = base_to_int((unsigned char)contig_seq[s + (int32_t)p]);
            atomicAdd(&coverage[p], 1u);
            atomicMax(&max_size[p], 1u);
        } else {
            nib = 15;   // terminator (the reference sets the trailing nibble(s) of the last byte(s) to 15)
        }
        v |= h ? nib : nib << 4;
    }
    tags[b] = (uint8_t)v;
}

struct StreamDesc {   // one kept record
    uint32_t read;        // index into the candidate arrays
    uint32_t col0, aln_len, aln_t_s;   // contig coordinate
    uint64_t tag_off;
};

__global__ void k2_tags(const StreamDesc* sd, uint32_t n_streams, const int32_t* pos, const uint32_t* n_cigar,
                        const uint64_t* cigar_off, const uint64_t* seq_off, const uint32_t* cigar, const uint8_t* seq,
                        const char* contig_seq, int32_t s, uint32_t gap_min_len, uint8_t* tags, DevStat st, uint32_t* te_out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_streams) return;
    const StreamDesc d = sd[k];
    ReadView rv{pos[d.read], n_cigar[d.read], cigar + cigar_off[d.read], seq + seq_off[d.read]};
    AlnSpan a{d.col0, d.aln_len, d.aln_t_s, 0};
    DevStatSink sink{st};
    te_out[k] = emit_tags(rv, contig_seq, a, s, gap_min_len, tags + d.tag_off, sink);
}

// link observations of one stream; kFill = false counts per column, true scatters
template <bool kFill>
__global__ void k2_links(const uint64_t* tag_off, const uint32_t* aln_t_s, uint32_t n_streams, const uint8_t* tags,
                         uint32_t* col_cnt, const uint32_t* col_off, uint32_t* cursor, LinkObs* obs) {
    const uint32_t rd = blockIdx.x * blockDim.x + threadIdx.x;
    if (rd >= n_streams) return;
    const uint8_t* tg = tags + tag_off[rd];
    uint32_t d = 0;
    Tag p1{0, 0, 0};
    uint64_t pp = KEY_HEAD, ppp = KEY_HEAD;
    uint32_t pp_base = 0;
    while (next_tag(tg, aln_t_s[rd], &d, &p1)) {
        const uint64_t key = node_key(p1.t_pos, p1.delta, p1.q_base);
        if (p1.q_base != 6 && pp_base != 6) {
            if (!kFill) {
                atomicAdd(&col_cnt[p1.t_pos], 1u);
            } else {
                const uint32_t at = col_off[p1.t_pos] + atomicAdd(&cursor[p1.t_pos], 1u);
                LinkObs o;
                o.pp = pp; o.ppp = ppp; o.rd = rd; o.delta = (uint16_t)p1.delta; o.base = (uint8_t)p1.q_base; o.pad = 0;
                obs[at] = o;
            }
        }
        ppp = pp; pp = key; pp_base = p1.q_base;
    }
}

__global__ void k2_build(LinkObs* obs, const uint32_t* col_off, uint32_t n_cols, Entry* entries, Node* nodes, uint32_t* col_nn) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_cols) return;
    LinkObs* o = obs + col_off[p];
    const uint32_t n = col_off[p + 1] - col_off[p];
    // order by (stream, delta): the order update_msa meets the observations in (insertion sort, buckets hold ~depth items)
    for (uint32_t i = 1; i < n; ++i) {
        const LinkObs x = o[i];
        const uint64_t kx = (uint64_t)x.rd << 16 | x.delta;
        uint32_t j = i;
        while (j > 0 && ((uint64_t)o[j - 1].rd << 16 | o[j - 1].delta) > kx) { o[j] = o[j - 1]; --j; }
        o[j] = x;
    }
    col_nn[p] = build_column(o, n, entries + col_off[p], nodes + col_off[p]);
}

__global__ void k2_pack_stat(const uint32_t* coverage, const uint32_t* max_size, const uint32_t* l_ins, const uint32_t* l_del,
                             uint32_t n, ColStat* st) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    ColStat c;
    c.coverage = (uint16_t)coverage[p];
    c.max_size = (uint16_t)max_size[p];
    c.l_ins = (uint16_t)l_ins[p];
    c.l_del = (uint16_t)l_del[p];
    st[p] = c;
}

struct DpResult {
    long long gbest;
    uint64_t gkey;
    uint32_t cons_len, status;   // status: 0 ok, 1 no end column, 2 backtrace left the graph, 3 zero coverage
    unsigned long long cyc[4];   // k2_dp_wave: cycles in tile staging, entry phase, node phase, tiles
};

__global__ void k2_dp(MsaView mv, int32_t l, int read_type, DpResult* res) {
    if (blockIdx.x || threadIdx.x) return;
    long long gbest = INT64_MIN;
    uint64_t gkey = node_key(0, 0, 0xff);
    for (int32_t p = 0; p < l; ++p) {
        switch (read_type) {
            case READS_CLR: dp_column<READS_CLR>(mv, p, l, &gbest, &gkey); break;
            case READS_HIFI: dp_column<READS_HIFI>(mv, p, l, &gbest, &gkey); break;
            case READS_RS: dp_column<READS_RS>(mv, p, l, &gbest, &gkey); break;
            default: dp_column<READS_ONT>(mv, p, l, &gbest, &gkey); break;
        }
    }
    res->gbest = gbest;
    res->gkey = gkey;
    res->status = key_base(gkey) == 0xff ? 1u : 0u;
}

// writes the consensus backwards into cons[cap-1], cons[cap-2], ...; the host reads the last cons_len items
__global__ void k2_backtrace(MsaView mv, DpResult* res, ConsBase* cons, uint32_t cap) {
    if (blockIdx.x || threadIdx.x) return;
    if (res->status) return;
    uint64_t cur = res->gkey;
    uint32_t n = 0;
    for (;;) {
        const int32_t tp = key_tpos(cur);
        Node* nd = find_node(mv, tp, key_delta(cur) << 8 | key_base(cur));
        if (!nd) { res->status = 2; break; }
        const Entry& be = mv.entries[mv.col_off[tp] + nd->start + nd->best];
        if (key_base(cur) != 4) {
            const uint32_t cov = mv.stat[tp].coverage;
            if (cov == 0 || n >= cap) { res->status = 3; break; }
            ConsBase cb;
            cb.qv = (char)(100 * be.link / cov);
            const char up = int_to_base(key_base(cur));
            cb.base = (cov > 4u && cb.qv > 20) ? up : (char)(up >= 'A' && up <= 'Z' ? up + 32 : up);
            cb.pos = (uint32_t)tp;
            cons[cap - 1 - n] = cb;
            ++n;
        }
        cur = be.pp;
        if (key_tpos(cur) == -1) break;
    }
    res->cons_len = n;
}

// tag streams of gapped string pairs (the concatenated low-quality regions); one lane per pair
__global__ void k2_tags_str(const char* pool, const uint64_t* str_off, const uint32_t* str_len, uint32_t n, uint32_t gap_min_len,
                            const uint64_t* tag_off, uint8_t* tags, DevStat st, uint32_t* te_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    StrColIter f{pool + str_off[2 * i], pool + str_off[2 * i + 1], 0};
    DevStatSink sink{st};
    te_out[i] = emit_tags_from(f, str_len[i], 0u, gap_min_len, tags + tag_off[i], sink);
}

// ---- wave-per-window chain DP --------------------------------------------------------------------------------
// k2_resolve (lane per column): for every entry the position of its predecessor node's entry list and the mask of
// the entries in it whose pp equals this entry's ppp -- everything the DP needs besides the running scores.
struct EntryDp {
    uint32_t pred_first;   // global index of the predecessor node's first entry
    uint32_t pred_mask;    // bit n: predecessor entry n matches (n < 32)
    uint16_t link;
    uint16_t meta;         // bit0 head, bits1-3 pp.base, bits4-6 ppp.base, bit7 pp.delta > 0, bit8 ppp.delta > 1, bits9-15 own delta
};
__global__ void k2_resolve(MsaView mv, uint32_t n_cols, EntryDp* dp, uint32_t* deep_flag) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_cols) return;
    const uint32_t e0 = mv.col_off[p], e1 = mv.col_off[p + 1];
    // only the entries that belong to a node are live: nodes[] holds their ranges
    const Node* nd = mv.nodes + e0;
    const uint32_t nn = mv.col_nn[p];
    if (e1 - e0 > 1024u) atomicOr(deep_flag, 1u);
    for (uint32_t j = 0; j < nn; ++j) {
        for (uint32_t m = 0; m < nd[j].len; ++m) {
            const uint32_t g = e0 + nd[j].start + m;
            const Entry& em = mv.entries[g];
            EntryDp d;
            d.link = (uint16_t)em.link;
            d.meta = (uint16_t)((key_base(em.pp) & 7u) << 1 | (key_base(em.ppp) & 7u) << 4 | (key_delta(em.pp) > 0 ? 0x80u : 0u) |
                                (key_delta(em.ppp) > 1 ? 0x100u : 0u) | ((nd[j].key >> 8) < 127u ? (nd[j].key >> 8) : 127u) << 9);
            if ((nd[j].key >> 8) >= 127u) atomicOr(deep_flag, 1u);
            d.pred_first = 0;
            d.pred_mask = 0;
            if (key_tpos(em.pp) == -1) {
                d.meta |= 1u;
            } else {
                const int32_t tp = key_tpos(em.pp);
                const Node* ppn = find_node(mv, tp, key_delta(em.pp) << 8 | key_base(em.pp));
                if (ppn) {
                    d.pred_first = mv.col_off[tp] + ppn->start;
                    const Entry* PE = mv.entries + d.pred_first;
                    for (uint32_t n = 0; n < ppn->len; ++n)
                        if (PE[n].pp == em.ppp) {
                            if (n < 32) d.pred_mask |= 1u << n;
                            else atomicOr(deep_flag, 1u);
                        }
                }
            }
            dp[g] = d;
        }
    }
}

constexpr uint32_t DPW_RING = 8192, DPW_TILE = 1024;   // score ring (entries), static data staged per tile (entries)

// The per-node loop of get_cns_from_align_tags (dp_column in np2_core.h) touches the predecessor scores only through
// three aggregates per entry m over its matching predecessor entries n (in list order): whether any matched, the
// maximum score and the score of the last match.  Proof sketch (rules as in the reference, ens = predecessor score):
//   score(m) = max(0, max_n(ens) + w)                     [entries start at 0 and are only raised; w = 10 link - C cov]
//   p_pp_score_ <- max_n(ens) iff max_n(ens) + w > 0        [the last raise happens at the first n attaining the max]
//   ONT   : if (pp/ppp insertion flags && link test) every match fires: best = m, p_pp_score = ens(last match);
//           else fires iff link > link(best)/2 && base test && ens > p_pp_score, and once fired best = m keeps the
//           link test true, so the net effect is best = m iff max_n(ens) > p_pp_score, p_pp_score = max of the two
//   CLR/HiFi: ens > p || (ens == p && pp.base != '-')  ==>  best = m iff max_n(ens) > p (or >= p when pp.base != '-')
// so the aggregates are computed by one lane per ENTRY (independent LDS reads), and only the cheap best-index
// recurrence stays serial per node.
template <int TYPE>
__device__ __forceinline__ uint32_t dp_node_select(const EntryDp* E, const long long* emax, const long long* elast, const uint8_t* eany,
                                                   uint32_t len, uint32_t g0, uint32_t b, long long cov, const long long* ring) {
    uint32_t best = 0;
    long long ps_ = INT64_MIN, ps = INT64_MIN;
    constexpr long long C = TYPE == READS_HIFI ? 4 : 3;
    int tmp = 0;
    if (TYPE == READS_ONT)
        for (uint32_t mi = 0; mi < len; ++mi)
            if ((int)E[mi].link > tmp) tmp = (int)E[mi].link;
    for (uint32_t mi = 0; mi < len; ++mi) {
        const EntryDp em = E[mi];
        const uint32_t ppb = (em.meta >> 1) & 7u, pppb = (em.meta >> 4) & 7u;
        const long long score = ring[(g0 + mi) & (DPW_RING - 1)];
        if (!(em.meta & 1u) && eany[mi]) {
            const long long mx = emax[mi];
            if (mx + 10 * (long long)em.link - C * cov > 0) ps_ = mx;
            if (TYPE == READS_CLR || TYPE == READS_HIFI) {
                if (mx > ps || (mx == ps && ppb != 4)) { best = mi; ps = mx > ps ? mx : ps; }
            } else if (TYPE == READS_ONT) {
                if (((em.meta & 0x100u) || (em.meta & 0x80u)) && ((double)em.link > (double)cov * 0.2 || (int)em.link > tmp / 2)) {
                    best = mi;
                    ps = elast[mi];
                } else if ((int)em.link > (int)E[best].link / 2 && (ppb == 4 || ppb == b || pppb == b || ppb == pppb) && mx > ps) {
                    best = mi;
                    ps = mx;
                }
            }
        }
        const long long bs = ring[(g0 + best) & (DPW_RING - 1)];
        if (TYPE == READS_RS) {
            if (score >= bs) { best = mi; ps = ps_; }
        } else if (score > bs || (score == bs && ppb != 4)) {
            best = mi;
            ps = ps_;
        }
    }
    return best;
}

// One wave walks the window's columns.  Per column and insertion level: lanes = entries (scores + aggregates from the
// LDS score ring), then lanes = nodes (best-index recurrence).  Static data of up to DPW_TILE entries (<= 64
// columns) is staged through LDS per tile with coalesced loads.
template <int TYPE>
__global__ __launch_bounds__(64) void k2_dp_wave(MsaView mv, const EntryDp* dp, int32_t l, DpResult* res) {
    __shared__ long long ring[DPW_RING];
    __shared__ long long s_max[DPW_TILE], s_last[DPW_TILE];
    __shared__ EntryDp s_e[DPW_TILE];
    __shared__ Node s_n[DPW_TILE];
    __shared__ uint8_t s_any[DPW_TILE];
    __shared__ uint32_t s_off[66];
    __shared__ uint32_t s_nn[64], s_cov[64], s_lvl[64];   // per column of the tile: nodes, coverage, insertion levels
    constexpr long long C = TYPE == READS_HIFI ? 4 : 3;
    const int lane = threadIdx.x;
    long long gbest = INT64_MIN;
    uint64_t gkey = node_key(0, 0, 0xff);
    int32_t p0 = 0;
    unsigned long long cy0 = 0, cy1 = 0, cy2 = 0, ntile = 0;
    while (p0 < l) {
        long long tc = clock64();
        ++ntile;
        // ---- tile = columns [p0, p1): as many as fit DPW_TILE entries (at least one, at most 64)
        const int32_t pc = p0 + lane + 1 <= l ? p0 + lane + 1 : l;
        const uint32_t base = mv.col_off[p0];
        const uint32_t myoff = mv.col_off[pc];
        const unsigned long long fits = __ballot(myoff - base <= DPW_TILE && p0 + lane + 1 <= l);   // a prefix mask (offsets are monotone)
        int ncol = (int)__popcll(fits);
        if (ncol == 0) ncol = 1;   // (a single column never exceeds the tile: k2_resolve flags it)
        const int32_t p1 = p0 + ncol;
        if (lane < ncol) {
            s_off[lane + 1] = myoff - base;
            s_nn[lane] = mv.col_nn[p0 + lane];
            const ColStat cs = mv.stat[p0 + lane];
            s_cov[lane] = cs.coverage;
            s_lvl[lane] = cs.max_size;
        }
        if (lane == 0) s_off[0] = 0;
        const uint32_t ne = __shfl(myoff, ncol - 1) - base;
        for (uint32_t i = lane; i < ne; i += 64) {
            s_e[i] = dp[base + i];
            s_n[i] = mv.nodes[base + i];
        }
        __syncthreads();
        { const long long t = clock64(); cy0 += (unsigned long long)(t - tc); tc = t; }
        for (int32_t p = p0; p < p1; ++p) {
            const uint32_t co = s_off[p - p0], cn = s_off[p - p0 + 1] - co;   // entries of the column: [co, co + cn)
            const uint32_t nn = s_nn[p - p0];
            const long long cov = s_cov[p - p0];
            const uint32_t levels = s_lvl[p - p0];
            for (uint32_t lvl = 0; lvl < levels; ++lvl) {
                // ---- lanes = entries of this level: score + aggregates
                for (uint32_t ib = 0; ib < cn; ib += 64) {
                    const uint32_t i = co + ib + lane;
                    if (ib + lane < cn) {
                        const EntryDp em = s_e[i];
                        if ((uint32_t)(em.meta >> 9) == lvl) {
                            long long score = 0, mx = INT64_MIN, last = 0;
                            uint8_t any = 0;
                            if (em.meta & 1u) {
                                score = 10 * (long long)em.link - C * cov;
                            } else {
                                uint32_t mask = em.pred_mask;
                                while (mask) {
                                    const uint32_t n = (uint32_t)__builtin_ctz(mask);
                                    mask &= mask - 1;
                                    const long long ens = ring[(em.pred_first + n) & (DPW_RING - 1)];
                                    if (ens > mx) mx = ens;
                                    last = ens;
                                    any = 1;
                                }
                                if (any) {
                                    const long long cand = mx + 10 * (long long)em.link - C * cov;
                                    if (cand > 0) score = cand;
                                }
                            }
                            ring[(base + i) & (DPW_RING - 1)] = score;
                            s_max[i] = mx;
                            s_last[i] = last;
                            s_any[i] = any;
                        }
                    }
                }
                __syncthreads();
                { const long long t = clock64(); cy1 += (unsigned long long)(t - tc); tc = t; }
                // ---- lanes = nodes of this level: best index
                for (uint32_t jb = 0; jb < nn; jb += 64) {
                    const uint32_t j = jb + lane;
                    if (j < nn) {
                        Node& nd = s_n[co + j];
                        if ((nd.key >> 8) == lvl) {
                            const uint32_t o = co + nd.start;
                            nd.best = dp_node_select<TYPE>(s_e + o, s_max + o, s_last + o, s_any + o, nd.len, base + o, nd.key & 0xffu, cov, ring);
                        }
                    }
                }
                __syncthreads();
                { const long long t = clock64(); cy2 += (unsigned long long)(t - tc); tc = t; }
            }
            if (p == l - 1 && lane == 0) {
                for (uint32_t j = 0; j < nn; ++j) {
                    const Node& nd = s_n[co + j];
                    if (!nd.len) continue;
                    const long long bs = ring[(base + co + nd.start + nd.best) & (DPW_RING - 1)];
                    if (bs >= gbest) {
                        gkey = node_key(p, nd.key >> 8, nd.key & 0xffu);
                        if (bs > gbest) gbest = bs;
                    }
                }
            }
        }
        for (uint32_t i = lane; i < ne; i += 64) mv.nodes[base + i].best = s_n[i].best;
        __syncthreads();
        p0 = p1;
    }
    if (lane == 0) {
        res->gbest = gbest;
        res->gkey = gkey;
        res->status = key_base(gkey) == 0xff ? 1u : 0u;
        res->cyc[0] = cy0; res->cyc[1] = cy1; res->cyc[2] = cy2; res->cyc[3] = ntile;
    }
}

__global__ void k2_dp_lq(MsaView mv, int32_t len, const uint32_t* max_size, int hifi, DpResult* res) {
    if (blockIdx.x || threadIdx.x) return;
    if (hifi) for (int32_t p = 0; p < len; ++p) dp_column_lq<true>(mv, p);
    else for (int32_t p = 0; p < len; ++p) dp_column_lq<false>(mv, p);
    res->gkey = node_key(len - 1, max_size[len - 1] - 1, 5);   // last node the reference's loops visit (ctg_cns.c:1036-1038,1090-1092)
    res->status = 0;
    res->gbest = 0;
}

// backtrace of get_lqseqs_from_align_tags (ctg_cns.c:1104-1143): characters in backtrace order, no reversal
__global__ void k2_backtrace_lq(MsaView mv, DpResult* res, char* out, uint32_t cap) {
    if (blockIdx.x || threadIdx.x) return;
    uint64_t cur = res->gkey;
    uint32_t n = 0;
    for (;;) {
        const int32_t tp = key_tpos(cur);
        Node* nd = find_node(mv, tp, key_delta(cur) << 8 | key_base(cur));
        if (!nd || nd->len == 0) { res->status = 2; break; }
        const Entry& be = mv.entries[mv.col_off[tp] + nd->start + nd->best];
        if (key_base(cur) != 4) {
            if (n >= cap) { res->status = 3; break; }
            const char up = int_to_base(key_base(cur));
            const uint32_t q = be.link & 0xffffu;
            out[n++] = (q * 5 > mv.stat[tp].coverage || up == 'N') ? up : (char)(up >= 'A' && up <= 'Z' ? up + 32 : up);
        }
        cur = be.pp;
        if (key_tpos(cur) == -1) break;
    }
    res->cons_len = n;
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
    ~HipExec() override { if (stream_) (void)hipStreamDestroy(stream_); }
    bool init(std::string* err) {
        HIPOK(hipSetDevice(device_));
        HIPOK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        return true;
    }
    bool run_window(const WindowInput& in, WindowOutput* out, std::string* err) override;
    bool run_lq(const LqInput& in, std::string* cons_rev, std::string* err) override;

  private:
    // link observations -> column buckets -> nodes/entries for n_streams tag streams over n_cols columns; *total = entries
    bool build_graph(uint32_t n_streams, uint32_t n_cols, uint32_t* total, std::string* err);
    int device_;
    hipStream_t stream_ = nullptr;
    DevBuf contig_, pos_, ncig_, cigoff_, seqoff_, cigar_, seq_, spans_, sd_, tags_, tagoff_, alnts_, te_, cnt4_, stat_, colcnt_,
        coloff_, cursor_, sums_, obs_, entries_, nodes_, colnn_, res_, cons_, strpool_, stroff_, strlen_, edp_, flag_;
    uint64_t contig_serial_ = ~0ull;
    size_t contig_len_ = 0;
};

bool HipExec::run_window(const WindowInput& in, WindowOutput* out, std::string* err) {
    HIPOK(hipSetDevice(device_));
    hipStream_t q = stream_;
    const int32_t s = in.s, e = in.e, l = e - s;
    const uint32_t n = (uint32_t)in.n_reads();
    StageClock clk(q);
    out->kept.assign(n, 0);
    out->bad_cigar = false;
    out->cons.clear();
    // ---- contig characters (uploaded once per contig: the pipeline passes the same pointer for every window)
    const size_t clen = strlen(in.contig_seq);
    if (contig_serial_ != in.contig_serial || contig_len_ != clen) {
        if (!contig_.ensure(clen + 16)) { *err = "out of device memory (contig)"; return false; }
        HIPOK(hipMemcpyAsync(contig_.p, in.contig_seq, clen + 1, hipMemcpyHostToDevice, q));
        contig_serial_ = in.contig_serial;
        contig_len_ = clen;
    }
    // ---- candidate records
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> bool {
        if (!b.ensure(bytes + 16)) return false;
        return bytes == 0 || hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, q) == hipSuccess;
    };
    if (!up(pos_, in.pos.data(), 4ull * n) || !up(ncig_, in.n_cigar.data(), 4ull * n) || !up(cigoff_, in.cigar_off.data(), 8ull * n) ||
        !up(seqoff_, in.seq_off.data(), 8ull * n) || !up(cigar_, in.cigar.data(), 4ull * in.cigar.size()) ||
        !up(seq_, in.seq.data(), in.seq.size())) { *err = "out of device memory (records)"; return false; }
    std::vector<SpanOut> spans(n);
    if (n) {
        if (!spans_.ensure(sizeof(SpanOut) * (size_t)n)) { *err = "out of device memory (spans)"; return false; }
        k2_span<<<nblk(n, 64), 64, 0, q>>>(pos_.as<int32_t>(), ncig_.as<uint32_t>(), cigoff_.as<uint64_t>(), seqoff_.as<uint64_t>(),
                                           cigar_.as<uint32_t>(), seq_.as<uint8_t>(), n, contig_.as<char>(), s, e, spans_.as<SpanOut>());
        HIPOK(hipMemcpyAsync(spans.data(), spans_.p, sizeof(SpanOut) * (size_t)n, hipMemcpyDeviceToHost, q));
    }
    HIPOK(hipStreamSynchronize(q));
    clk.mark("upload+span");
    for (uint32_t i = 0; i < n; ++i)
        if (spans[i].bad) { out->bad_cigar = true; return true; }
    // ---- 500 bp rule + coverage caps (ctg_cns.c:3540-3545).  Coverage of a column = number of kept streams whose
    // [aln_t_s, aln_t_e) covers it, so the order-dependent caps can be decided here from the spans alone.
    std::vector<uint32_t> cand;
    for (uint32_t i = 0; i < n; ++i) {
        const SpanOut& a = spans[i];
        if (a.aln_t_s > a.aln_t_e - 500u) continue;
        const uint32_t ts = a.aln_t_s - (uint32_t)s, te = a.aln_t_e - (uint32_t)s;
        if (ts > (uint32_t)l || te > (uint32_t)l) { *err = "alignment outside its window"; return false; }
        cand.push_back(i);
    }
    {
        std::vector<int32_t> diff((size_t)l + 2, 0);
        diff[0] += 1; diff[(size_t)l] -= 1;   // seed
        for (uint32_t i : cand) { ++diff[spans[i].aln_t_s - (uint32_t)s]; --diff[spans[i].aln_t_e - (uint32_t)s]; }
        int32_t run = 0, mx = 0;
        for (int32_t p = 0; p <= l; ++p) { run += diff[(size_t)p]; mx = std::max(mx, run); }
        if (mx <= 500) {
            for (uint32_t i : cand) out->kept[i] = 1;
        } else {   // deep pileup: replay the reference's order-dependent decisions on a running coverage track
            std::vector<uint32_t> cov((size_t)l + 1, 0);
            for (int32_t p = 0; p < l; ++p) cov[(size_t)p] = 1;
            for (uint32_t i : cand) {
                const uint32_t ts = spans[i].aln_t_s - (uint32_t)s, te = spans[i].aln_t_e - (uint32_t)s;
                if ((cov[ts] > 3000 && cov[te] > 3000) ||
                    (cov[ts] > 500 && cov[te] > 500 && (double)in.aligned_q[i] < in.l_qseq[i] * 0.9)) continue;
                out->kept[i] = 1;
                for (uint32_t p = ts; p < te; ++p) ++cov[p];
            }
        }
    }
    // ---- stream layout: seed + kept records
    std::vector<StreamDesc> sd;
    out->tag_off.clear(); out->aln_t_s.clear(); out->aln_t_e.clear();
    uint64_t tag_bytes = 0;
    out->tag_off.push_back(0);
    out->aln_t_s.push_back(0);
    tag_bytes += ((uint64_t)l + 1) / 2 + 1;
    tag_bytes = (tag_bytes + 3) & ~3ull;
    for (uint32_t i = 0; i < n; ++i) {
        if (!out->kept[i]) continue;
        StreamDesc d;
        d.read = i; d.col0 = spans[i].col0; d.aln_len = spans[i].aln_len; d.aln_t_s = spans[i].aln_t_s; d.tag_off = tag_bytes;
        sd.push_back(d);
        out->tag_off.push_back(tag_bytes);
        out->aln_t_s.push_back(spans[i].aln_t_s - (uint32_t)s);
        tag_bytes += ((uint64_t)spans[i].aln_len + 1) / 2 + 1;
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
    k2_seed_tags<<<nblk(((uint64_t)l + 1) / 2 + 1, 256), 256, 0, q>>>(contig_.as<char>(), s, (uint32_t)l, tags_.as<uint8_t>(), st.coverage, st.max_size);
    if (!sd.empty()) {
        HIPOK(hipMemcpyAsync(sd_.p, sd.data(), sizeof(StreamDesc) * sd.size(), hipMemcpyHostToDevice, q));
        k2_tags<<<nblk(sd.size(), 64), 64, 0, q>>>(sd_.as<StreamDesc>(), (uint32_t)sd.size(), pos_.as<int32_t>(), ncig_.as<uint32_t>(),
                                                   cigoff_.as<uint64_t>(), seqoff_.as<uint64_t>(), cigar_.as<uint32_t>(), seq_.as<uint8_t>(),
                                                   contig_.as<char>(), s, in.gap_min_len, tags_.as<uint8_t>(), st, te_.as<uint32_t>() + 1);
    }
    k2_pack_stat<<<nblk(n_cols, 256), 256, 0, q>>>(st.coverage, st.max_size, st.l_ins, st.l_del, n_cols, stat_.as<ColStat>());
    HIPOK(hipMemcpyAsync(tagoff_.p, out->tag_off.data(), 8ull * n_streams, hipMemcpyHostToDevice, q));
    HIPOK(hipMemcpyAsync(alnts_.p, out->aln_t_s.data(), 4ull * n_streams, hipMemcpyHostToDevice, q));
    clk.mark("tags");
    uint32_t total = 0;
    if (!build_graph(n_streams, n_cols, &total, err)) return false;
    if (!cons_.ensure(sizeof(ConsBase) * ((size_t)total + 16))) { *err = "out of device memory (consensus)"; return false; }
    clk.mark("links+build");
    // ---- chain DP + backtrace
    MsaView mv{coloff_.as<uint32_t>(), colnn_.as<uint32_t>(), nodes_.as<Node>(), entries_.as<Entry>(), stat_.as<ColStat>()};
    const uint32_t cons_cap = total + 8;
    {
        // parallel resolve + wave DP; windows with a column too deep for the LDS staging fall back to the one-lane walk
        static const bool force_lane = getenv("NP2_DP") && strcmp(getenv("NP2_DP"), "lane") == 0;
        bool wave_ok = !force_lane;
        if (wave_ok) {
            if (!edp_.ensure(sizeof(EntryDp) * (size_t)total + 64) || !flag_.ensure(16)) { *err = "out of device memory (dp)"; return false; }
            HIPOK(hipMemsetAsync(flag_.p, 0, 4, q));
            k2_resolve<<<nblk(n_cols, 64), 64, 0, q>>>(mv, n_cols, edp_.as<EntryDp>(), flag_.as<uint32_t>());
            uint32_t deep = 0;
            HIPOK(hipMemcpyAsync(&deep, flag_.p, 4, hipMemcpyDeviceToHost, q));
            HIPOK(hipStreamSynchronize(q));
            clk.mark("resolve");
            wave_ok = deep == 0;
        }
        if (wave_ok) {
            switch (in.read_type) {
                case READS_CLR: k2_dp_wave<READS_CLR><<<1, 64, 0, q>>>(mv, edp_.as<EntryDp>(), l, res_.as<DpResult>()); break;
                case READS_HIFI: k2_dp_wave<READS_HIFI><<<1, 64, 0, q>>>(mv, edp_.as<EntryDp>(), l, res_.as<DpResult>()); break;
                case READS_RS: k2_dp_wave<READS_RS><<<1, 64, 0, q>>>(mv, edp_.as<EntryDp>(), l, res_.as<DpResult>()); break;
                default: k2_dp_wave<READS_ONT><<<1, 64, 0, q>>>(mv, edp_.as<EntryDp>(), l, res_.as<DpResult>()); break;
            }
        } else {
            k2_dp<<<1, 64, 0, q>>>(mv, l, in.read_type, res_.as<DpResult>());
        }
    }
    clk.mark("dp");
    k2_backtrace<<<1, 64, 0, q>>>(mv, res_.as<DpResult>(), cons_.as<ConsBase>(), cons_cap);
    DpResult res;
    HIPOK(hipMemcpyAsync(&res, res_.p, sizeof(res), hipMemcpyDeviceToHost, q));
    HIPOK(hipStreamSynchronize(q));
    if (clk.on) fprintf(stderr, "[np2 dp cycles] tile staging %llu, entry phase %llu, node phase %llu, tiles %llu\n", res.cyc[0], res.cyc[1], res.cyc[2], res.cyc[3]);
    if (res.status == 1) { *err = "no alignment column reaches the end of the window"; return false; }
    if (res.status == 2) { *err = "backtrace left the graph"; return false; }
    if (res.status == 3) { *err = "zero coverage on the consensus path"; return false; }
    out->cons.resize(res.cons_len);
    out->stat.resize(n_cols);
    out->tags.resize(tag_bytes);
    out->aln_t_e.assign(n_streams, 0);
    if (res.cons_len)
        HIPOK(hipMemcpyAsync(out->cons.data(), cons_.as<ConsBase>() + (cons_cap - res.cons_len), sizeof(ConsBase) * (size_t)res.cons_len,
                             hipMemcpyDeviceToHost, q));
    HIPOK(hipMemcpyAsync(out->stat.data(), stat_.p, sizeof(ColStat) * (size_t)n_cols, hipMemcpyDeviceToHost, q));
    HIPOK(hipMemcpyAsync(out->tags.data(), tags_.p, tag_bytes, hipMemcpyDeviceToHost, q));
    if (n_streams > 1) HIPOK(hipMemcpyAsync(out->aln_t_e.data() + 1, te_.as<uint32_t>() + 1, 4ull * (n_streams - 1), hipMemcpyDeviceToHost, q));
    HIPOK(hipStreamSynchronize(q));
    clk.mark("backtrace+download");
    clk.flush("window", l, n_streams, total);
    out->aln_t_e[0] = (uint32_t)l;
    return true;
}

bool HipExec::build_graph(uint32_t n_streams, uint32_t n_cols, uint32_t* total_out, std::string* err) {
    hipStream_t q = stream_;
    k2_links<false><<<nblk(n_streams, 64), 64, 0, q>>>(tagoff_.as<uint64_t>(), alnts_.as<uint32_t>(), n_streams, tags_.as<uint8_t>(),
                                                        colcnt_.as<uint32_t>(), nullptr, nullptr, nullptr);
    const uint32_t nsb = nblk(n_cols + 1, SCAN_TILE);
    k2_scan_sums<<<nsb, SCAN_T, 0, q>>>(colcnt_.as<uint32_t>(), n_cols + 1, sums_.as<uint32_t>());
    k2_scan_top<<<1, 1, 0, q>>>(sums_.as<uint32_t>(), nsb);
    k2_scan_final<<<nsb, SCAN_T, 0, q>>>(colcnt_.as<uint32_t>(), n_cols + 1, sums_.as<uint32_t>(), coloff_.as<uint32_t>());
    uint32_t total = 0;
    HIPOK(hipMemcpyAsync(&total, coloff_.as<uint32_t>() + n_cols, 4, hipMemcpyDeviceToHost, q));
    HIPOK(hipStreamSynchronize(q));
    if (!obs_.ensure(sizeof(LinkObs) * (size_t)total + 64) || !entries_.ensure(sizeof(Entry) * (size_t)total + 64) ||
        !nodes_.ensure(sizeof(Node) * (size_t)total + 64)) {
        *err = "out of device memory (link graph)";
        return false;
    }
    k2_links<true><<<nblk(n_streams, 64), 64, 0, q>>>(tagoff_.as<uint64_t>(), alnts_.as<uint32_t>(), n_streams, tags_.as<uint8_t>(),
                                                       nullptr, coloff_.as<uint32_t>(), cursor_.as<uint32_t>(), obs_.as<LinkObs>());
    k2_build<<<nblk(n_cols, 64), 64, 0, q>>>(obs_.as<LinkObs>(), coloff_.as<uint32_t>(), n_cols, entries_.as<Entry>(), nodes_.as<Node>(),
                                              colnn_.as<uint32_t>());
    *total_out = total;
    return true;
}

bool HipExec::run_lq(const LqInput& in, std::string* cons_rev, std::string* err) {
    HIPOK(hipSetDevice(device_));
    hipStream_t q = stream_;
    const uint32_t n_streams = (uint32_t)in.t.size();
    const uint32_t n_cols = in.t_len + 1 + 32;   // slack: see the fill quirk in np2_lq.cpp
    std::vector<char> pool;
    std::vector<uint64_t> str_off, tag_off;
    std::vector<uint32_t> str_len, zeros(n_streams, 0);
    uint64_t tag_bytes = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        str_off.push_back(pool.size());
        pool.insert(pool.end(), in.t[i].begin(), in.t[i].end());
        pool.push_back('\0');
        str_off.push_back(pool.size());
        pool.insert(pool.end(), in.q[i].begin(), in.q[i].end());
        pool.push_back('\0');
        str_len.push_back((uint32_t)in.t[i].size());
        tag_off.push_back(tag_bytes);
        tag_bytes += ((uint64_t)in.t[i].size() + 1) / 2 + 1;
        tag_bytes = (tag_bytes + 3) & ~3ull;
    }
    if (!strpool_.ensure(pool.size() + 16) || !stroff_.ensure(8ull * str_off.size() + 16) || !strlen_.ensure(4ull * n_streams + 16) ||
        !tags_.ensure(tag_bytes + 16) || !cnt4_.ensure(16ull * n_cols + 64) || !stat_.ensure(sizeof(ColStat) * (size_t)n_cols) ||
        !tagoff_.ensure(8ull * n_streams + 16) || !alnts_.ensure(4ull * n_streams + 16) || !te_.ensure(4ull * n_streams + 16) ||
        !colcnt_.ensure(4ull * (n_cols + 2)) || !coloff_.ensure(4ull * (n_cols + 2)) || !cursor_.ensure(4ull * (n_cols + 2)) ||
        !sums_.ensure(4ull * (nblk(n_cols + 1, SCAN_TILE) + 2)) || !colnn_.ensure(4ull * n_cols) || !res_.ensure(sizeof(DpResult))) {
        *err = "out of device memory (low-quality regions)";
        return false;
    }
    HIPOK(hipMemcpyAsync(strpool_.p, pool.data(), pool.size(), hipMemcpyHostToDevice, q));
    HIPOK(hipMemcpyAsync(stroff_.p, str_off.data(), 8ull * str_off.size(), hipMemcpyHostToDevice, q));
    HIPOK(hipMemcpyAsync(strlen_.p, str_len.data(), 4ull * n_streams, hipMemcpyHostToDevice, q));
    HIPOK(hipMemcpyAsync(tagoff_.p, tag_off.data(), 8ull * n_streams, hipMemcpyHostToDevice, q));
    HIPOK(hipMemcpyAsync(alnts_.p, zeros.data(), 4ull * n_streams, hipMemcpyHostToDevice, q));
    HIPOK(hipMemsetAsync(tags_.p, 0, tag_bytes + 16, q));
    HIPOK(hipMemsetAsync(cnt4_.p, 0, 16ull * n_cols + 64, q));
    HIPOK(hipMemsetAsync(colcnt_.p, 0, 4ull * (n_cols + 2), q));
    HIPOK(hipMemsetAsync(cursor_.p, 0, 4ull * (n_cols + 2), q));
    DevStat st{cnt4_.as<uint32_t>(), cnt4_.as<uint32_t>() + n_cols, cnt4_.as<uint32_t>() + 2ull * n_cols, cnt4_.as<uint32_t>() + 3ull * n_cols};
    k2_tags_str<<<nblk(n_streams, 64), 64, 0, q>>>(strpool_.as<char>(), stroff_.as<uint64_t>(), strlen_.as<uint32_t>(), n_streams, in.gap_min_len,
                                                   tagoff_.as<uint64_t>(), tags_.as<uint8_t>(), st, te_.as<uint32_t>());
    k2_pack_stat<<<nblk(n_cols, 256), 256, 0, q>>>(st.coverage, st.max_size, st.l_ins, st.l_del, n_cols, stat_.as<ColStat>());
    uint32_t total = 0;
    if (!build_graph(n_streams, n_cols, &total, err)) return false;
    const uint32_t cap = total + 8;
    if (!cons_.ensure((size_t)cap + 16)) { *err = "out of device memory (low-quality consensus)"; return false; }
    MsaView mv{coloff_.as<uint32_t>(), colnn_.as<uint32_t>(), nodes_.as<Node>(), entries_.as<Entry>(), stat_.as<ColStat>()};
    k2_dp_lq<<<1, 64, 0, q>>>(mv, (int32_t)in.t_len, st.max_size, in.hifi ? 1 : 0, res_.as<DpResult>());
    k2_backtrace_lq<<<1, 64, 0, q>>>(mv, res_.as<DpResult>(), cons_.as<char>(), cap);
    DpResult res;
    HIPOK(hipMemcpyAsync(&res, res_.p, sizeof(res), hipMemcpyDeviceToHost, q));
    HIPOK(hipStreamSynchronize(q));
    if (res.status) { *err = "low-quality backtrace left the graph"; return false; }
    cons_rev->resize(res.cons_len);
    if (res.cons_len) HIPOK(hipMemcpyAsync(&(*cons_rev)[0], cons_.p, res.cons_len, hipMemcpyDeviceToHost, q));
    HIPOK(hipStreamSynchronize(q));
    return true;
}

}  // namespace

Exec* make_exec(std::string* err) {
    const int d = pick_device(err);
    if (d < 0) return nullptr;
    HipExec* x = new HipExec(d);
    if (!x->init(err)) { delete x; return nullptr; }
    return x;
}

}  // namespace np2

extern "C" int np2_device_index(void) {
    std::string err;
    return np2::pick_device(&err);
}
