// k_tile9 (round 4): the pileup vote with FOUR SLOTS PER LANE and the agreeing votes counted per record, not per vote.
//
// What the reference does per vote (source/lib/contig.c:247-331, base.c:60-71) -- look the 3-base context up in the slot's list, bump
// its count -- k_tile3 does per (record, 64-slot chunk) step with one lane per slot: ~75 vector instructions for 64 votes of which
// more than 99 % repeat the draft's own context.  Here a lane owns four consecutive slots plus the two in front of them (its
// WINDOW, six slots), and the record loop only asks one question per (record, lane): does the record cover the whole window inside
// one matched CIGAR segment with the draft's own six bases?
//     one unaligned 32-bit LDS read of its packed bases (eight 4-bit codes), one shift, one compare against the draft's six codes:
//     equal -> every context of the four slots is the draft's: one add to a per-lane counter.
// Every other (record, lane) pair that touches the lane's own slots -- a read start or end, a substituted base, an indel, a window
// with insertion columns -- is DEFERRED: the pair goes on the wave's list in LDS (dense: the entries of one step lie side by side),
// threaded into a chain per lane.  After the record loop of a staging round the entries are EVALUATED 64 at a time, whoever owns
// them: the record's symbol at each covered position of the owner's window, from the record's segment table held in registers
// (t9_code: the formulation k_tile3 uses for every vote, np1_desc.h: desc_symbol).  After the last round the entries are TALLIED
// in record order, which keeps every slot's contexts in first-seen order (base.c:60-71) without any sorting:
//   * a lane with a few entries walks its own chain;
//   * a lane over a draft error defers every record that covers it (they all disagree with the draft there), as many entries as the
//     pileup is deep on a few lanes of every wave.  Such a lane's entries are tallied by the WHOLE WAVE: the list is in record order,
//     so 64 entries at a time the distinct contexts of a slot come out in first-seen order as ballots (lowest set lane first) with
//     their counts as popcounts, and go into the owner's histogram with that weight.
//
// A wave owns T9_CH = 4 vote chunks = 248 slots: lane 0 carries the four slots in front of them (only "did every vote agree on the
// base" of the last one is needed, for the run structure), lanes 1..62 own four slots each, lane 63 idles.  A wave whose list
// overflows (T9_DL entries) or that meets a slot with more contexts than its lists hold sends its chunks to the redo list, i.e. to k_tile3.
//
// This header holds the per-lane logic for host and device; the kernel (np1_kernels.hip: k_tile9) adds the staging, the record
// loop and the wave-level bookkeeping, the host model (tests/model/np1_model.cpp, np1m_fused = 2) drives the same functions with
// plain loops and is compared with the oracle on the CPU.
#pragma once
#include "np1_core.h"
#include "np1_desc.h"

namespace np1k {

constexpr uint32_t T9_CH = 4;                      // vote chunks per wave
constexpr uint32_t T9_SLOTS = T9_CH * VOTE_CH;     // 248
constexpr uint32_t T9_DL = 768;                    // deferred entries per wave (8 bytes each in LDS)
constexpr uint32_t T9_HOT = 6;                     // a lane with more entries than this is tallied by the whole wave
// evaluated entry ("code"): bits 23..0 the six symbols (position 0 in bits 23..20), bits 26..24 first covered position, bits 29..27 last one

struct T9Win {
    uint32_t s0;          // first own slot (window position 2)
    uint32_t vmask;       // bit p: window position p is a slot of the batch
    uint32_t imask;       // bit p: it is an insertion column
    uint32_t g[6];        // draft index per position (of the base an insertion column follows)
    uint32_t D;           // draft symbols in slot space (3 = DEL on insertion columns), position 0 in bits 23..20
    bool plain;           // six valid positions, no insertion column: positions are consecutive draft bases from g[0] on
    bool active;          // the lane takes part at all (its own slots exist)
};

// per-position slot_info / slot_g words -> window (vmask: the positions that are slots of the batch)
NP1_HD void t9_window_from(uint32_t s0, bool active, uint32_t vmask, const uint32_t info[6], const uint32_t g[6], T9Win* w) {
    w->s0 = s0;
    w->vmask = vmask;
    w->imask = 0;
    w->D = 0;
    w->active = active;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int p = 0; p < 6; ++p) {
        const bool v = (vmask >> p) & 1u;
        w->g[p] = g[p];
        if (v && (info[p] & SI_INSERT)) w->imask |= 1u << p;
        w->D |= (info[p] & 0xfu) << (20 - 4 * p);
    }
    w->plain = vmask == 63u && w->imask == 0u;
}

// slot arrays -> window.  tile_s0 = first slot of the wave's 248; lane as in the kernel.
NP1_HD void t9_window(uint32_t tile_s0, int lane, uint32_t S, const uint8_t* slot_info, const uint32_t* slot_g, T9Win* w, uint32_t info_out[6]) {
    const int64_t s0 = (int64_t)tile_s0 + 4 * ((int64_t)lane - 1);
    uint32_t vmask = 0, g[6];
    for (int p = 0; p < 6; ++p) {
        const int64_t s = s0 - 2 + p;
        const bool v = s >= 0 && s < (int64_t)S;
        info_out[p] = v ? slot_info[s] : 0u;
        g[p] = v ? slot_g[s] : 0u;
        if (v) vmask |= 1u << p;
    }
    t9_window_from((uint32_t)s0, lane <= 62 && s0 >= 0 && s0 < (int64_t)S, vmask, info_out, g, w);
}

// eight 4-bit codes from query index q on, the first one in bits 31..28 (seven when q is odd)
NP1_HD uint32_t t9_fetch8(const uint8_t* seq, uint32_t q) {
    typedef uint32_t __attribute__((aligned(1))) u32u;
    const uint32_t v = __builtin_bswap32(*reinterpret_cast<const u32u*>(seq + (q >> 1)));
    return v << ((q & 1u) << 2);
}

// ---- the record loop's question, per (record, lane) -----------------------------------------------------------------------------
// The record's fields are wave-uniform; the kernel prefetches them into registers and precomputes per segment
//   lim = len - 5 for a matched segment of at least six bases, else 0   (a window starting at segment offset o lies inside iff o < lim)
// T9_AGREE: the record covers the lane's whole window inside one matched segment with the draft's six bases.
// T9_ENTRY: it touches the lane's own slots in any other way.  T9_SKIP: it does not touch them.
enum { T9_SKIP = 0, T9_AGREE = 1, T9_ENTRY = 2 };
constexpr int T9_NSEG_FAST = 3;     // segments the record loop looks at (a read with one deletion has three); windows in later ones are deferred

NP1_HD uint32_t t9_seg_lim(uint32_t w) {       // w = len | qcode << 16
    const uint32_t len = w & 0xffffu;
    return ((w >> 16) != 0xffffu && len >= 6u) ? len - 5u : 0u;
}
struct T9Rec {                                  // what a step needs of a record
    uint32_t sf, sl;                            // its run in slot space (whole record); sf > sl: votes on nothing
    bool chain;                                 // more parts than the head descriptor: never counted here
    uint32_t glo[T9_NSEG_FAST], lim[T9_NSEG_FAST], qc[T9_NSEG_FAST];
};
NP1_HD T9Rec t9_rec(const uint32_t* d) {
    T9Rec r;
    r.sf = d[0];
    r.sl = d[DESC_NEXT + 1];
    r.chain = (d[2] & DESC_CHAIN) != 0;
    const uint32_t nseg = d[2] & 0xffu;
    for (int k = 0; k < T9_NSEG_FAST; ++k) {
        const bool has = (uint32_t)k < nseg;
        const uint32_t w = has ? d[DESC_SEG0 + 2 * k + 1] : 0u;
        r.glo[k] = has ? d[DESC_SEG0 + 2 * k] : 0u;
        r.lim[k] = has ? t9_seg_lim(w) : 0u;
        r.qc[k] = w >> 16;
    }
    return r;
}
NP1_HD int t9_step(const T9Rec& r, const uint8_t* seq, const T9Win& w) {
    if (!w.active || r.sf > r.sl || r.sf > w.s0 + 3u || r.sl < w.s0) return T9_SKIP;
    const bool full = r.sf + 2u <= w.s0 && r.sl >= w.s0 + 3u;
    if (!full || !w.plain || r.chain) return T9_ENTRY;
    uint32_t q = 0;
    bool in = false;
    for (int k = T9_NSEG_FAST - 1; k >= 0; --k) {
        const uint32_t o = w.g[0] - r.glo[k];
        const bool ink = o < r.lim[k];
        q = ink ? r.qc[k] + o : q;
        in = in || ink;
    }
    if (!in) return T9_ENTRY;
    return (t9_fetch8(seq, q) >> 8) == w.D ? T9_AGREE : T9_ENTRY;
}

// ---- evaluation of a deferred (record, lane) pair ---------------------------------------------------------------------------------
// One part's segment and insertion tables, in registers on the device (the loops below are unrolled with predicates there).
struct T9Segs {
    uint32_t glo[DESC_NSEG], w[DESC_NSEG];      // g_lo, len | qcode << 16; unused entries have length 0
    uint32_t ip[DESC_NINS], iw[DESC_NINS];      // draft index, len | q0 << 16; unused entries have length 0
};
NP1_HD T9Segs t9_load_segs(const uint32_t* d) {
    T9Segs t;
    const uint32_t nseg = d[2] & 0xffu, nins = (d[2] >> 8) & 0xffu;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < DESC_NSEG; ++k) {
        const bool has = (uint32_t)k < nseg;
        t.glo[k] = has ? d[DESC_SEG0 + 2 * k] : 0u;
        t.w[k] = has ? d[DESC_SEG0 + 2 * k + 1] : 0u;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < DESC_NINS; ++k) {
        const bool has = (uint32_t)k < nins;
        t.ip[k] = has ? d[DESC_INS0 + 2 * k] : 0u;
        t.iw[k] = has ? d[DESC_INS0 + 2 * k + 1] : 0u;
    }
    return t;
}
// the symbol the part votes at a covered slot (draft index g, insertion column jj or -1): desc_symbol on the register tables
NP1_HD uint32_t t9_symbol(const T9Segs& t, uint32_t g, int32_t jj, const uint8_t* seq) {
    uint32_t q = 0;
    bool hit = false, del = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = DESC_NSEG - 1; k >= 0; --k) {          // (the first segment that holds g wins, as in desc_symbol)
        const uint32_t off = g - t.glo[k];
        const bool in = jj < 0 && off < (t.w[k] & 0xffffu);
        const uint32_t qc = t.w[k] >> 16;
        q = in ? qc + off : q;
        del = in ? qc == 0xffffu : del;
        hit = hit || in;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = DESC_NINS - 1; k >= 0; --k) {
        const bool in = jj >= 0 && t.ip[k] == g && (uint32_t)jj < (t.iw[k] & 0xffffu);
        q = in ? (t.iw[k] >> 16) + (uint32_t)jj : q;
        del = in ? false : del;
        hit = hit || in;
    }
    return (hit && !del) ? seq_nib(seq, (int32_t)q) : 3u;      // an insertion column the record only passes, a deletion, padding: DEL
}
NP1_HD void t9_cover(uint32_t sf, uint32_t sl, uint32_t s0, uint32_t* lo, uint32_t* hi) {
    const uint32_t a = sf + 2u;
    *lo = (a > s0 ? a : s0) - s0;                                   // first covered window position
    *hi = (sl < s0 + 3u ? sl : s0 + 3u) - s0 + 2u;                  // last one
}
// d = the record's head descriptor; ovf_pool = the parts of chained records (HBM); s0, g, jj = the owner's window (jj[p] = insertion
// column of position p, -1 for a base slot)
NP1_HD uint32_t t9_code(const uint32_t* d, const uint32_t* ovf_pool, const uint8_t* seq, uint32_t s0, const uint32_t g[6], const int32_t jj[6]) {
    uint32_t lo, hi;
    t9_cover(d[0], d[DESC_NEXT + 1], s0, &lo, &hi);
    uint32_t W = 0;
    if (!(d[2] & DESC_CHAIN)) {
        const T9Segs t = t9_load_segs(d);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t p = 0; p < 6; ++p)
            if (p >= lo && p <= hi) W |= t9_symbol(t, g[p], jj[p], seq) << (20 - 4 * p);
    } else {       // rare: the parts cover consecutive slot runs; the head lives where d points, the others in the overflow pool
        uint32_t part_slast = d[1], next = d[DESC_NEXT];
        const uint32_t* part = nullptr;
        for (uint32_t p = lo; p <= hi && p < 6; ++p) {
            const uint32_t s = s0 - 2u + p;
            while (s > part_slast && next) {
                part = ovf_pool + (uint64_t)(next - 1) * DESC_WORDS;
                part_slast = part[1];
                next = part[DESC_NEXT];
            }
            uint32_t gp = 0;
            int32_t jp = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (uint32_t t = 0; t < 6; ++t)
                if (t == p) { gp = g[t]; jp = jj[t]; }             // (no dynamic register indexing)
            const uint32_t sym = part ? desc_symbol(part, gp, jp, SeqBytes{seq}) : desc_symbol(d, gp, jp, SeqBytes{seq});
            W |= sym << (20 - 4 * p);
        }
    }
    return W | lo << 24 | hi << 27;
}

// ---- tally ----------------------------------------------------------------------------------------------------------------------
// One context into a slot's histogram, `weight` times (the entries of a hot lane arrive as (context, count) pairs).
// The two register entries of a histogram (the draft's context and the first other one) are handled without branches.
template <int E>
NP1_HD void t9_tally_ctx(uint32_t k, uint32_t weight, bool on, VoteLane<E>& v, uint32_t& basemask, uint32_t* Lj, int lane) {
    const bool m0 = on && k == v.k0, m1 = on && !m0 && k == v.k1;
    v.c0 += m0 ? weight : 0u;
    v.c1 += m1 ? weight : 0u;
    basemask |= on ? 1u << (k & 0xfu) : 0u;
    if (on && !m0 && !m1) {                     // a context seen for the first time, or one kept in the list
        if (v.n == 1) { v.k1 = k; v.c1 = weight; v.n = 2; }
        else {
            bool found = false;
            for (uint32_t e = 2; e < v.n; ++e) {
                const uint32_t x = Lj[(e - 2) * 64 + lane];
                if ((x >> 16) == k) {
                    Lj[(e - 2) * 64 + lane] = (x & 0xffff0000u) | ((x + weight) & 0xffffu);
                    found = true;
                    break;
                }
            }
            if (!found) {
                if (v.n < (uint32_t)E) { Lj[(v.n - 2) * 64 + lane] = k << 16 | (weight & 0xffffu); ++v.n; }
                else v.ovf = true;
            }
        }
    }
}
// One evaluated entry into the four slots' histograms.  vl[j], basemask[j]: own slot j = window position j + 2; L = the wave's context
// lists, slot j's at L + j * (E - 2) * 64.  Returns the number of votes (covered own slots).
template <int E>
NP1_HD uint32_t t9_tally(uint32_t code, VoteLane<E> vl[4], uint32_t basemask[4], uint32_t* L, int lane) {
    const uint32_t lo = (code >> 24) & 7u, hi = (code >> 27) & 7u;
    const uint32_t W = code & 0xffffffu;       // uncovered positions are 0: a context's missing predecessors read as 0 (contig.c:262-266)
    uint32_t n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t p = j + 2;
        const bool cov = p >= lo && p <= hi;
        t9_tally_ctx<E>((W >> (20 - 4 * p)) & 0xfffu, 1u, cov, vl[j], basemask[j], L + j * (E - 2) * 64, lane);
        n += cov ? 1u : 0u;
    }
    return n;
}

}  // namespace np1k
