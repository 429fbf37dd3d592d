// snp_phase (task 3) bodies, host+device (reference: source/lib/snpphase.c:87-903 on top of contig.c / kmercount.c).
//
// Two record streams of the same contigs: the short reads (`sr`, the reference's contig->fp) and the long reads (`lr`,
// contig->tfp).  Work by stage and the shape it has here:
//   P1  insertion columns of the short reads                       lane per record, atomic max            (contig.c:170-245)
//   P2  per-slot base histogram of the level-2 short reads         lane per record, 2 atomics per vote    (contig.c:247-331, shift 16)
//   P3  per-slot verdict: settle / heterozygous candidate          lane per slot                          (snpphase.c:136-214)
//   P4  sites = bases with a candidate slot, their anchors         lane per base / per site               (snpphase.c:168-198)
//   P5  low-depth regions over the slots                           lane per contig over the marked slots  (contig.c:498-620)
//   P6  long-read insertion columns behind marked bases, re-slot   lane per record / per base             (contig.c:202-245 with a mask)
//   P7  site verdict from the spanning reads of both streams       lane per site                          (snpphase.c:216-349)
//   P9  low-depth regions: score chain over both streams           lane per region                        (snpphase.c:797-871)
//   P10 haplotype links between neighbouring sites                 lane per (region, record)              (snpphase.c:351-448,615-795)
//   the chain over the sites (log10 scores) runs on the host: it is one short sequential pass per contig and has to use the
//   host's libm to agree with the reference bit for bit (snpphase.c:450-557).
// The same bodies compile for the host model the CPU tests check against the oracle (tests/model).
#pragma once
#include "np1_kmer.h"

namespace np1p {
using namespace np1k;

constexpr uint32_t F_ZERO = 1, F_DEPTH = 4, F_SNP = 8, F_THIRD = 16, F_INSERT = 32, F_LEFT = 64, F_RIGHT = 128;
constexpr uint32_t ERR_SP_UNDEFINED = 1024, ERR_SP_POOL = 2048, ERR_SP_DEPTH = 4096;
constexpr uint32_t SYM_DEL = 3;

NP1_HD void sp_atomic_add(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
#endif
}
NP1_HD void sp_atomic_add(int32_t* p, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
#endif
}
NP1_HD void sp_atomic_min64(unsigned long long* p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (*p > v) *p = v;
#endif
}
NP1_HD void sp_atomic_and8(uint8_t* p, uint32_t keep) {   // byte-wide and through the containing word
#if defined(__HIP_DEVICE_COMPILE__)
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    uint32_t* w = reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3) * 8;
    atomicAnd(w, (keep << sh) | ~(0xffu << sh));
#else
    *p = (uint8_t)(*p & keep);
#endif
}
NP1_HD void sp_atomic_or8(uint8_t* p, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    uint32_t* w = reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3);
    atomicOr(w, bits << ((uint32_t)(a & 3) * 8));
#else
    *p = (uint8_t)(*p | bits);
#endif
}

struct SpParams {   // reference: Configure (config.h:25-67)
    int32_t min_depth_snp, min_count_snp, min_count_snp_link, max_variant_count_lgs, read_len, ext_len_edge;
    double min_snp_factor_sgs, max_clip_ratio_lgs, rate_lgs, max_indel_factor_lgs, max_snp_factor_lgs, ploidy;
};

// contig_read_fliter2 (contig.c:679-686): 1 = the long read takes part
NP1_HD uint32_t sp_lr_level(const ReadsDev& R, int64_t r, double max_clip) {
    if ((R.flag[r] & 0xD04) != 0) return 0;
    const uint32_t ncig = R.n_cigar[r];
    double cliprate = 0;
    if (ncig > 0) {
        const uint32_t* cg = R.cigar + R.cigar_off[r];
        int32_t addlen = 0;
        if (cig_op(cg[0]) == 4) addlen += cig_len(cg[0]);
        if (cig_op(cg[ncig - 1]) == 4) addlen += cig_len(cg[ncig - 1]);
        cliprate = R.l_qseq[r] > 0 ? addlen / (double)R.l_qseq[r] : 0;
    }
    return cliprate <= max_clip ? 1u : 0u;
}

// ---- P1 / P6: insertion columns of one record over the whole contig (contig.c:170-180,202-245) -------------------
// mask == 0: every insertion counts; else only those behind a base whose main slot carries one of the mask bits
NP1_HD void sp_insert_record(const ReadsDev& R, int64_t r, const uint32_t* ctg_off, uint32_t* ins, uint32_t mask, const uint32_t* soff,
                             const uint8_t* sflag) {
    const uint32_t ncig = R.n_cigar[r];
    if (!ncig) return;
    const uint32_t ct = R.ctg[r];
    const uint32_t g0 = ctg_off[ct];
    const int32_t L = (int32_t)(ctg_off[ct + 1] - g0);
    const uint32_t* cg = R.cigar + R.cigar_off[r];
    int32_t pos = R.pos[r];
    for (uint32_t i = 0; i < ncig; ++i) {
        const uint32_t op = cig_op(cg[i]);
        if (op == 0 || op == 2) pos += cig_len(cg[i]);
        else if (op == 1 && pos > 0 && pos <= L - 1) {
            const uint32_t g = g0 + (uint32_t)pos - 1;
            if (mask == 0 || (sflag[soff[g]] & mask)) np1_atomic_max(&ins[g], (uint32_t)cig_len(cg[i]));
        }
    }
}

// ---- P2: base histogram with first-seen record per (slot, symbol) (contig.c:247-331 with shift 16: the context is the base) ----
struct SpHistSink {
    const uint32_t* soff;
    uint32_t g0;
    uint32_t* cnt;     // [slot * 16 + sym]
    uint32_t* first;   // [slot * 16 + sym], 0xffffffff = never
    uint32_t r;
    NP1_HD void vote(int32_t pos, uint32_t col, uint32_t sym, int32_t, bool) {
        const uint64_t k = (uint64_t)(soff[g0 + (uint32_t)pos] + col) * 16 + sym;
        sp_atomic_add(&cnt[k], 1u);
        np1_atomic_min(&first[k], r);
    }
};
NP1_HD void sp_hist_record(const KcCtx& c, int64_t r, uint32_t* cnt, uint32_t* first) {
    if (c.level[r] != 2 || c.R.n_cigar[r] == 0) return;
    const uint32_t ct = c.R.ctg[r];
    const uint32_t g0 = c.ctg_off[ct];
    const int32_t L = (int32_t)(c.ctg_off[ct + 1] - g0);
    SpHistSink sink{c.soff, g0, cnt, first, (uint32_t)r};
    kc_walk(c, r, g0, 0, L - 1, sink);
}

// stable top two of a slot's symbols: by count, the symbol seen first wins a tie (base.c:91-121 over the first-seen list)
struct SpTop2 { uint32_t n, sym0, c0, sym1, c1, total; };
NP1_HD SpTop2 sp_top2(const uint32_t* cnt16, const uint32_t* first16) {
    SpTop2 t{0, 0, 0, 0, 0, 0};
    uint32_t last = 0;
    bool any = false;
    for (;;) {   // symbols in first-seen order
        uint32_t best = 16, bf = 0xffffffffu;
        for (uint32_t s = 0; s < 16; ++s)
            if (cnt16[s] && (!any || first16[s] > last) && first16[s] < bf) { bf = first16[s]; best = s; }
        if (best == 16) break;
        any = true;
        last = bf;
        const uint32_t cc = cnt16[best] & 0xffffu;
        t.total += cnt16[best];
        if (t.n == 0) { t.sym0 = best; t.c0 = cc; t.n = 1; }
        else if (t.n == 1) {
            if (cc > t.c0) { t.sym1 = t.sym0; t.c1 = t.c0; t.sym0 = best; t.c0 = cc; }
            else { t.sym1 = best; t.c1 = cc; }
            t.n = 2;
        } else if (cc > t.c1) {
            if (cc > t.c0) { t.sym1 = t.sym0; t.c1 = t.c0; t.sym0 = best; t.c0 = cc; }
            else { t.sym1 = best; t.c1 = cc; }
        }
    }
    return t;
}

NP1_HD int32_t sp_check(const SpParams& P, int32_t count, double rate, bool same) {   // ts_check_snps (snpphase.c:205-214)
    if (rate < P.min_snp_factor_sgs && same) return 0;
    if (rate == 0 || (count >= P.min_count_snp && !same && rate < P.min_snp_factor_sgs)) return 2;
    return 1;
}

// ---- P3: one slot of ts_find_snps (snpphase.c:146-167): marks, settle or candidate; dec[s] = 0 / 1 (candidate) / 2 (settled)
NP1_HD void sp_slot_decide(const SpParams& P, uint32_t s, const uint32_t* cnt, const uint32_t* first, uint8_t* sbase, uint8_t* sflag,
                           uint16_t* scount, uint8_t* dec, uint8_t* top, uint32_t* err) {
    const SpTop2 t = sp_top2(cnt + (uint64_t)s * 16, first + (uint64_t)s * 16);
    if (t.total > 0xffffu) np1_atomic_or(err, ERR_SP_DEPTH);   // the reference's 16-bit counters would have wrapped
    const uint32_t count = t.total & 0xffffu;
    uint32_t fl = sflag[s];
    if (count == 0) fl |= F_ZERO; else fl &= ~F_ZERO;
    if ((int32_t)count <= P.min_depth_snp) fl |= F_DEPTH; else fl &= ~F_DEPTH;
    uint32_t d = 0;
    if (count > 0) {
        const double rate = t.n == 1 ? 0 : t.c1 / (double)t.c0;
        d = (uint32_t)sp_check(P, (int32_t)count, rate, t.sym0 == sbase[s]);
        if (d == 2) sbase[s] = (uint8_t)t.sym0;
    }
    sflag[s] = (uint8_t)fl;
    scount[s] = (uint16_t)count;
    dec[s] = (uint8_t)d;
    top[s] = (uint8_t)(t.sym0 | t.sym1 << 4);
}

// ---- P4: a base is a site when one of its slots is a candidate; the first such slot names the two alleles ---------------------
// returns 1 and the allele byte pair when base g (local index i of a contig of length L) is a site
NP1_HD uint32_t sp_base_site(uint32_t g, int32_t i, int32_t L, const uint32_t* soff, const uint8_t* dec, const uint8_t* top, uint8_t* alleles) {
    const uint32_t s0 = soff[g];
    const uint32_t s1 = (i == L - 1) ? s0 + 1 : soff[g + 1];   // the walk ends on the main slot of the last base
    for (uint32_t s = s0; s < s1; ++s)
        if (dec[s] == 1) { *alleles = top[s]; return 1; }
    return 0;
}
// anchors of a site: the nearest bases without a candidate slot (snpphase.c:172,188-197); dirty[] is per base, local indices returned
NP1_HD void sp_site_anchors(const uint8_t* dirty_ctg, int32_t i, int32_t L, int32_t* left, int32_t* right) {
    int32_t a = i - 1;
    while (a >= 0 && dirty_ctg[a]) --a;
    *left = a >= 0 ? a : 0;
    int32_t b = i + 1;
    while (b < L && dirty_ctg[b]) ++b;
    *right = b < L ? b : L - 1;
}

// ---- P5: contig_get_region(0, L-1, gap, 0, FLAG_DEPTH, no extension) over the slots (contig.c:498-560), sparse -----------------
// F = ascending slot ids of the contig's marked slots (m of them); sown[s] = global base index of slot s.  Writes local (start, end)
// pairs; returns the number of values or -1 when out is too small.
NP1_HD int32_t sp_depth_regions(const uint32_t* F, uint32_t m, const uint32_t* soff, const uint32_t* sown, uint32_t g0, int32_t L, uint32_t gap,
                                int32_t ext, int32_t* out, int32_t out_cap) {
    int32_t n = 0;
    uint32_t k = 0;
    const int32_t end = L - 1;
    const int64_t last_slot = soff[g0 + (uint32_t)end];   // the walk stops on the main slot of the last base
    int64_t cursor = soff[g0];
    while (k < m) {
        if ((int64_t)F[k] < cursor || (int64_t)F[k] > last_slot) { ++k; continue; }
        int32_t qstart = (int32_t)(sown[F[k]] - g0), qend = qstart;
        int64_t last = F[k];
        ++k;
        for (;;) {
            const bool has_next = k < m && (int64_t)F[k] <= last_slot;
            const int64_t next = has_next ? (int64_t)F[k] : (int64_t)1 << 40;
            if (!has_next || next - last - 1 > (int64_t)gap) {
                const int64_t close = last + (int64_t)gap + 1;   // the slot on which the gap counter exceeds `gap`
                qstart = qstart >= ext ? qstart - ext : 0;
                qend = qend <= end - ext ? qend + ext : end;
                if (n + 2 > out_cap) return -1;
                out[n++] = qstart; out[n++] = qend;
                if (close > last_slot) return n;                  // still open when the walk ends
                cursor = close + 1;
                if (qend > (int32_t)(sown[close] - g0)) cursor = (int64_t)soff[g0 + (uint32_t)qend] + 1;
                break;
            }
            qend = (int32_t)(sown[next] - g0);
            last = next;
            ++k;
        }
    }
    return n;
}

// ---- P6: new slot space; columns that exist keep their state, new ones start as DEL with the marks of their base -----------------
NP1_HD void sp_reslot_base(uint32_t g, const uint32_t* soff1, const uint32_t* soff2, const uint8_t* sbase1, const uint8_t* sflag1, const uint16_t* scount1,
                           uint8_t* sbase2, uint8_t* sflag2, uint16_t* scount2, uint32_t* sown2) {
    const uint32_t a = soff1[g], n1 = soff1[g + 1] - a, b = soff2[g], n2 = soff2[g + 1] - b;
    for (uint32_t j = 0; j < n2; ++j) {
        if (j < n1) { sbase2[b + j] = sbase1[a + j]; sflag2[b + j] = sflag1[a + j]; scount2[b + j] = scount1[a + j]; }
        else { sbase2[b + j] = (uint8_t)SYM_DEL; sflag2[b + j] = sflag1[a]; scount2[b + j] = 0; }
        sown2[b + j] = g;
    }
}

// per base: what the link walk looks at on every step, in one load
NP1_HD uint16_t sp_base_mark(uint32_t g, const uint32_t* soff, const uint8_t* sflag) {
    return (uint16_t)(sflag[soff[g]] | (soff[g + 1] - soff[g] > 1 ? 0x100u : 0u));
}

// 64 bases of the "look here" bitmap: bit set = the base carries one of `mask`
NP1_HD unsigned long long sp_base_bits_word(uint64_t w, uint64_t G, const uint16_t* bmark, uint32_t mask) {
    unsigned long long v = 0;
    for (uint32_t t = 0; t < 64; ++t) {
        const uint64_t g = w * 64 + t;
        if (g < G && (bmark[g] & mask)) v |= 1ull << t;
    }
    return v;
}
// distance from base g to the next marked base, at most maxd
NP1_HD int32_t sp_next_marked(const unsigned long long* bits, uint64_t g, int32_t maxd) {
    int32_t d = 0;
    while (d < maxd) {
        const uint64_t idx = g + (uint64_t)d;
        const unsigned long long w = bits[idx >> 6] >> (idx & 63);
        if (w) {
#if defined(__HIP_DEVICE_COMPILE__)
            const int32_t t = __ffsll((long long)w) - 1;
#else
            const int32_t t = __builtin_ctzll(w);
#endif
            return d + t < maxd ? d + t : maxd;
        }
        d += 64 - (int32_t)(idx & 63);
    }
    return maxd;
}

}  // namespace np1p
#include "np1_phase_sites.h"
