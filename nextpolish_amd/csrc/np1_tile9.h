// k_tile9 (round 4): the pileup vote with FOUR SLOTS PER LANE and the agreeing votes counted per record, not per vote.
//
// What the reference does per vote (source/lib/contig.c:247-331, base.c:60-71) -- look the 3-base context up in the slot's list, bump
// its count -- k_tile3 does per (record, 64-slot chunk) step with one lane per slot: ~75 vector instructions for 64 votes of which
// more than 99 % repeat the draft's own context.  Here a lane owns four consecutive slots plus the two in front of them (its
// WINDOW, six slots), and a record that covers the whole window inside one matched CIGAR segment is handled by
//     one unaligned 32-bit LDS read of its packed bases (eight 4-bit codes), one shift, one compare against the draft's six codes:
//     equal -> every context of the four slots is the draft's: one add to a per-lane counter.
// Everything else is DEFERRED: the lane appends a 32-bit entry to the wave's list in LDS -- the six codes and the covered positions
// when the record's part of the window lies in one matched segment (a read start or end, a substituted base), or the record's index
// when it does not (an indel inside the window, chained descriptors, insertion columns with a partial cover).  The list is dense
// (entries of one step lie side by side) and every lane threads its own entries into a chain, so that after the record loop
//   * the index entries are turned into code entries 64 at a time, whoever owns them, by evaluating the record's symbol at each
//     position with desc_symbol (np1_desc.h), the formulation k_tile3 uses for every vote, and
//   * every lane walks its own chain in the order it was appended = record order, which keeps every slot's contexts in first-seen
//     order (base.c:60-71) without any sorting.
// A lane over a draft error defers every record that covers it (they all disagree with the draft there): chains are as long as the
// pileup is deep on a few lanes of every wave, which is why the conversion is dense and only the cheap tally is per lane.
//
// A wave owns T9_CH = 4 vote chunks = 248 slots: lane 0 carries the four slots in front of them (only "did every vote agree on the
// base" of the last one is needed, for the run structure), lanes 1..62 own four slots each, lane 63 idles.  A wave whose list
// overflows (T9_DL entries) or that meets a slot with more contexts than its lists hold sends its chunks to the redo list, i.e. to k_tile3.
//
// This header holds the per-lane logic for host and device; the kernel (np1_kernels.hip: k_tile9) adds the staging, the record
// loop and the wave-level bookkeeping, the host model (tests/model/np1_model.cpp, np1m_fused = 5) drives the same functions with
// plain loops and is compared with the oracle on the CPU.
#pragma once
#include "np1_core.h"
#include "np1_desc.h"

namespace np1k {

constexpr uint32_t T9_CH = 4;                      // vote chunks per wave
constexpr uint32_t T9_SLOTS = T9_CH * VOTE_CH;     // 248
constexpr uint32_t T9_DL = 512;                    // deferred entries per wave (8 bytes each in LDS: the entry and the lane's next one)
constexpr uint32_t T9_GENERAL = 1u << 31;          // entry: bit 31 set = record index (staged batch) in the low bits
// code entry: bits 23..0 the six symbols (position 0 in bits 23..20), bits 26..24 first covered position, bits 29..27 last one

struct T9Win {
    uint32_t s0;          // first own slot (window position 2)
    uint32_t vmask;       // bit p: window position p is a slot of the batch
    uint32_t imask;       // bit p: it is an insertion column
    uint32_t g[6];        // draft index per position (of the base an insertion column follows)
    uint32_t D;           // draft symbols in slot space (3 = DEL on insertion columns), position 0 in bits 23..20
    uint32_t Dc, nb;      // the base positions' symbols only, in order, right-aligned; how many
    uint32_t ga, gb, gq;  // a record covering the whole window votes the draft's symbols iff ONE matched segment holds [ga, gb] and its bases from gq on equal Dc
    bool plain;           // six valid positions, no insertion column: positions are consecutive draft bases from ga on
    bool active;          // the lane takes part at all (its own slots exist)
};

// slot arrays -> window.  tile_s0 = first slot of the wave's 248; lane as in the kernel.
NP1_HD void t9_window(uint32_t tile_s0, int lane, uint32_t S, const uint8_t* slot_info, const uint32_t* slot_g, T9Win* w, uint32_t info_out[6]) {
    const int64_t s0 = (int64_t)tile_s0 + 4 * ((int64_t)lane - 1);
    w->s0 = (uint32_t)s0;
    w->vmask = w->imask = 0;
    w->D = 0; w->Dc = 0; w->nb = 0;
    w->active = lane <= 62 && s0 >= 0 && s0 < (int64_t)S;
    uint32_t gq = 0;
    bool have_q = false;
    for (int p = 0; p < 6; ++p) {
        const int64_t s = s0 - 2 + p;
        const bool v = s >= 0 && s < (int64_t)S;
        const uint32_t info = v ? slot_info[s] : 0u;
        info_out[p] = info;
        w->g[p] = v ? slot_g[s] : 0u;
        if (v) w->vmask |= 1u << p;
        if (v && (info & SI_INSERT)) w->imask |= 1u << p;
        w->D |= (info & 0xfu) << (20 - 4 * p);
        if (v && !(info & SI_INSERT)) {
            w->Dc = w->Dc << 4 | (info & 0xfu);
            ++w->nb;
            if (!have_q) { gq = w->g[p]; have_q = true; }
        }
    }
    w->plain = w->vmask == 63u && w->imask == 0u;
    w->ga = w->g[0];
    w->gb = w->g[5] + ((w->imask >> 5) & 1u);
    w->gq = gq;
    if (w->vmask != 63u || w->nb == 0) { w->ga = 1; w->gb = 0; }     // (no segment holds an empty range's ends: such windows take the general path)
}

// eight 4-bit codes from query index q on, the first one in bits 31..28 (seven when q is odd)
NP1_HD uint32_t t9_fetch8(const uint8_t* seq, uint32_t q) {
    typedef uint32_t __attribute__((aligned(1))) u32u;
    const uint32_t v = __builtin_bswap32(*reinterpret_cast<const u32u*>(seq + (q >> 1)));
    return v << ((q & 1u) << 2);
}

enum { T9_SKIP = 0, T9_AGREE = 1, T9_ENTRY = 2 };

// One record (head descriptor d, its packed bases seq) against one lane's window.  idx = the record's index for a general entry.
NP1_HD int t9_classify(const uint32_t* d, const uint8_t* seq, const T9Win& w, uint32_t idx, uint32_t* entry) {
    const uint32_t sf = d[0], sl = d[DESC_NEXT + 1];
    if (!w.active || sf > sl || sf > w.s0 + 3u || sl < w.s0) return T9_SKIP;      // none of the lane's own slots
    const uint32_t a = sf + 2u;
    const uint32_t lo = a > w.s0 ? a - w.s0 : 0u;                                 // first covered window position (2..5 when the run starts inside)
    const uint32_t hi = sl >= w.s0 + 3u ? 5u : sl - w.s0 + 2u;                    // last one
    const bool full = lo == 0u && hi == 5u;
    *entry = T9_GENERAL | idx;
    const uint32_t cnt = d[2];
    if (cnt & DESC_CHAIN) return T9_ENTRY;
    if (!full && !w.plain) return T9_ENTRY;
    const uint32_t need_lo = full ? w.ga : w.ga + lo, need_hi = full ? w.gb : w.ga + hi, from = full ? w.gq : w.ga + lo;
    if (need_lo > need_hi) return T9_ENTRY;
    const uint32_t nseg = cnt & 0xffu;
    for (uint32_t k = 0; k < nseg; ++k) {
        const uint32_t g_lo = d[DESC_SEG0 + 2 * k], wd = d[DESC_SEG0 + 2 * k + 1], len = wd & 0xffffu, qc = wd >> 16;
        if (need_lo - g_lo < len && need_hi - g_lo < len && need_lo >= g_lo) {
            if (qc == 0xffffu) return T9_ENTRY;                                   // a deletion: DEL votes, through the general path
            const uint32_t F = t9_fetch8(seq, qc + (from - g_lo));
            if (full) {
                if ((F >> (32u - 4u * w.nb)) == w.Dc) return T9_AGREE;
                if (!w.plain) return T9_ENTRY;
                *entry = (F >> 8) | 5u << 27;
                return T9_ENTRY;
            }
            const uint32_t keep = (0xffffffu >> (4u * lo)) & ~(0xfffffu >> (4u * hi));   // positions lo..hi
            *entry = ((F >> 8) >> (4u * lo) & keep) | lo << 24 | hi << 27;
            return T9_ENTRY;
        }
    }
    return T9_ENTRY;
}

// An index entry -> a code entry: the record's symbol at every covered window position (all parts of a chained record).
template <class So>
NP1_HD uint32_t t9_general(const uint32_t* d, const uint32_t* ovf_pool, const uint8_t* seq, const T9Win& w, So so) {
    const uint32_t sf = d[0], sl = d[DESC_NEXT + 1];
    const uint32_t a = sf + 2u;
    const uint32_t lo = a > w.s0 ? a - w.s0 : 0u;
    const uint32_t hi = sl >= w.s0 + 3u ? 5u : sl - w.s0 + 2u;
    uint32_t W = 0;
    const uint32_t* part = d;
    for (uint32_t p = lo; p <= hi; ++p) {
        const uint32_t s = w.s0 - 2u + p;
        while (s > part[1] && part[DESC_NEXT]) part = ovf_pool + (uint64_t)(part[DESC_NEXT] - 1) * DESC_WORDS;   // parts cover consecutive runs
        const int32_t jj = ((w.imask >> p) & 1u) ? (int32_t)(s - so(w.g[p])) - 1 : -1;
        const uint32_t sym = desc_symbol(part, w.g[p], jj, SeqBytes{seq});
        W |= sym << (20 - 4 * p);
    }
    return W | lo << 24 | hi << 27;
}

// One code entry into the four slots' histograms.  vl[j], basemask[j]: own slot j = window position j + 2; L = the wave's context
// lists, slot j's at L + j * (E - 2) * 64.  Returns the number of votes (covered own slots).
template <int E>
NP1_HD uint32_t t9_tally(uint32_t entry, VoteLane<E> vl[4], uint32_t basemask[4], uint32_t* L, int lane) {
    const uint32_t lo = (entry >> 24) & 7u, hi = (entry >> 27) & 7u;
    const uint32_t W = entry & 0xffffffu;      // uncovered positions are 0: a context's missing predecessors read as 0 (contig.c:262-266)
    uint32_t n = 0;
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t p = j + 2;
        if (p < lo || p > hi) continue;
        const uint32_t k = (W >> (20 - 4 * p)) & 0xfffu;
        basemask[j] |= 1u << (k & 0xfu);
        vl[j].tally(k, L + j * (E - 2) * 64, lane);
        ++n;
    }
    return n;
}

}  // namespace np1k
