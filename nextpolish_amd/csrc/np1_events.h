// Event form of the pileup vote (k_tile5), host+device.
//
// At almost every slot almost every record votes the draft's own 3-base context k0 (the reads agree with the
// draft around that slot).  Those votes need no individual work: per slot
//     count(k0) = 1 (the draft) + coverage(slot) - #disagreeing votes(slot),      total = 1 + coverage(slot),
// and coverage is a difference array over the records' contiguous vote runs.  Only votes whose context differs
// from k0 -- the first two votes of a record, the three slots at/after a mismatching base, indel neighbourhoods --
// are produced individually as EVENTS (slot, record, context), a few per record instead of ~150, and tallied per
// slot in record order, which keeps the reference's first-seen list order (base.c:60-71) exactly.
//
// record_events() walks one record's vote run inside a slot window; it skips agreeing stretches eight bases at a
// time by XOR-ing the record's packed bases with the draft's packed symbols, and falls back to the exact per-slot
// symbol (np1_desc.h: desc_symbol) around every disagreement, so the emitted event set is exact, not heuristic.
#pragma once
#include "np1_core.h"
#include "np1_desc.h"

namespace np1k {

// slot window of one tile: plain arrays (LDS on the device), index = slot - w0
struct EvWindow {
    uint32_t w0, n;            // first slot, number of slots
    uint32_t own0;             // first slot whose events carry a valid context (w0 + 2, or w0 at the start of the batch)
    const uint8_t* sinfo;      // slot_info per window slot (draft symbol in the low nibble, SI_* bits)
    const uint32_t* sg;        // draft index of each window slot
    const uint16_t* k0;        // draft context of each window slot
    const uint8_t* dpk;        // draft symbols of the window's BASE positions, packed like BAM bases (high nibble = even index),
    uint32_t dpk_g0;           //   nibble i <-> draft index dpk_g0 + i
    const uint32_t* soff;      // global slot offsets (insertion column of an insertion slot = slot - soff[g] - 1)
};

// 8 nibbles starting at nibble index i of a BAM-packed stream, first nibble in the top bits
NP1_HD uint32_t nib8_be(const uint8_t* p, uint32_t i) {
    const uint8_t* b = p + (i >> 1);
    const uint32_t w = (uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | (uint32_t)b[3];
    return (i & 1u) ? (w << 4 | (uint32_t)b[4] >> 4) : w;
}
NP1_HD uint32_t clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clz((int)x);
#else
    return x ? (uint32_t)__builtin_clz(x) : 32u;
#endif
}

// Sink: void event(uint32_t slot, uint32_t rec_context /* 12 bits */)
// `d` = head descriptor of the record, parts chained through ovf_pool; sqb = its packed bases.
// kChained = false: the record is known to fit its head descriptor (keeps every descriptor access in one address space).
template <bool kChained, class Sink>
NP1_HD void record_events(const uint32_t* d, const uint32_t* ovf_pool, const uint8_t* sqb, const EvWindow& w, Sink& sink) {
    const uint32_t sf_all = d[0], sl_all = d[DESC_NEXT + 1];
    const uint32_t wend = w.w0 + w.n - 1;
    if (sf_all > sl_all || sl_all < w.w0 || sf_all > wend) return;
    uint32_t s = sf_all > w.w0 ? sf_all : w.w0;
    const uint32_t send = sl_all < wend ? sl_all : wend;
    // part of the record that covers slot s
    const uint32_t* part = d;
    if (kChained) {
        while (s > part[1]) {
            const uint32_t nx = part[DESC_NEXT];
            if (!nx) return;
            part = ovf_pool + (uint64_t)(nx - 1) * DESC_WORDS;
        }
    }
    uint32_t ctx = 0;        // the record's own symbols at the two previous slots of its run (0 before the run starts)
    uint32_t pending = 2;    // slots that must still be evaluated exactly (context may differ from the draft's)
    const SeqBytes sq{sqb};
    while (s <= send) {
        if (kChained) {
            while (s > part[1]) {
                const uint32_t nx = part[DESC_NEXT];
                if (!nx) return;
                part = ovf_pool + (uint64_t)(nx - 1) * DESC_WORDS;
            }
        }
        const uint32_t k = s - w.w0;
        const uint32_t info = w.sinfo[k];
        const uint32_t g = w.sg[k];
        if (pending == 0 && !(info & SI_INSERT)) {
            // fast path: inside an aligned segment, compare up to eight bases with the draft at once
            const uint32_t nseg = part[2] & 0xffu;
            for (uint32_t t = 0; t < nseg; ++t) {
                const uint32_t off = g - part[DESC_SEG0 + 2 * t], wd = part[DESC_SEG0 + 2 * t + 1];
                const uint32_t len = wd & 0xffffu, qc = wd >> 16;
                if (off < len && qc != 0xffffu) {
                    uint32_t n = len - off;                      // bases left in the segment
                    if (n > 8) n = 8;
                    if (n > send - s + 1) n = send - s + 1;
                    if (n > part[1] - s + 1) n = part[1] - s + 1;
                    // the n slots must be consecutive base slots (no insertion column in between)
                    uint32_t m = 1;
                    while (m < n && !(w.sinfo[k + m] & SI_INSERT)) ++m;
                    n = m;
                    const uint32_t rw = nib8_be(sqb, qc + off);
                    const uint32_t dw = nib8_be(w.dpk, g - w.dpk_g0);
                    uint32_t x = rw ^ dw;
                    if (n < 8) x &= ~(0xffffffffu >> (4 * n));
                    const uint32_t agree = x ? clz32(x) >> 2 : n;   // leading agreeing bases
                    if (agree) {
                        s += agree;
                        // the context now equals the draft's: reload it lazily when the exact path is entered again
                        ctx = 0xffffffffu;
                    }
                    break;
                }
            }
            if (ctx == 0xffffffffu) {
                if (s > send) break;
                // after a skip the two previous symbols are the draft's own
                const uint32_t kk = s - w.w0;
                ctx = (uint32_t)(w.sinfo[kk - 2] & 0xf) << 4 | (uint32_t)(w.sinfo[kk - 1] & 0xf);
                continue;
            }
        }
        // exact path: this record's symbol at slot s
        const int32_t jj = (info & SI_INSERT) ? (int32_t)(s - w.soff[g]) - 1 : -1;
        const uint32_t sym = desc_symbol(part, g, jj, sq);
        ctx = ((ctx & 0xffu) << 4) | sym;
        if (s >= w.own0) {
            if (ctx != w.k0[k]) sink.event(s, ctx);
        } else if (sym != (info & 0xfu)) {
            sink.event(s, sym);   // left-context slots: only the symbol matters (base mask of the previous slot)
        }
        if (sym != (info & 0xfu)) pending = 2;
        else if (pending) --pending;
        ++s;
    }
}

// ------------------------------------------------------------------------------------------------
// Group form of the same event set (k_tile6).  record_events() is sequential per record; here the work item is one
// (record, group of 8 consecutive window slots) pair and items are independent, so a tile's ~3000 items spread over
// all lanes.  An item is CLEAN when the record's symbols at the group's slots and at the two context slots before
// them all equal the draft's: then every vote in the group is the draft's context and nothing is emitted (apart from
// the record's first two votes, whose shortened contexts are known without looking at the bases).  Everything else
// -- mismatches, insertion columns, segment boundaries, chained records -- is DIRTY and evaluated exactly, one lane
// per slot (group_symbol), with the contexts assembled from the neighbouring lanes.
constexpr uint32_t EV_G = 8;        // slots per group
constexpr uint32_t EV_GL = 10;      // lanes per dirty item: two context slots + the group

// bit t set <=> window slot 8j-2+t is an insertion column or lies outside the window
NP1_HD uint32_t group_ins_mask(const uint8_t* sinfo, uint32_t n, uint32_t j) {
    uint32_t m = 0;
    for (uint32_t t = 0; t < EV_GL; ++t) {
        const int64_t k = (int64_t)EV_G * j - 2 + t;
        if (k < 0 || k >= (int64_t)n || (sinfo[k] & SI_INSERT)) m |= 1u << t;
    }
    return m;
}

// ten nibbles from nibble index i of a BAM-packed stream, left-aligned in a 64-bit word (top 40 bits)
NP1_HD uint64_t nib10_be(const uint8_t* p, uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    // three aligned words instead of six byte loads (the arrays are padded for the over-read)
    const uint8_t* a = p + (i >> 1);
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(a) & 3u);
    const uint32_t* wp = reinterpret_cast<const uint32_t*>(a - sh);
    const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
    const uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, sh), x1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
    const uint64_t v = (uint64_t)__builtin_bswap32(x0) << 32 | __builtin_bswap32(x1);
#else
    const uint8_t* b = p + (i >> 1);
    uint64_t v = 0;
    for (int t = 0; t < 6; ++t) v |= (uint64_t)b[t] << (56 - 8 * t);
#endif
    return (i & 1u) ? v << 4 : v;
}

struct GroupWin {            // what the group form needs on top of EvWindow
    const uint16_t* gins;    // group_ins_mask per group
};

// Window range [lo, hi] (window-relative) a record's votes cover; false when it does not touch the window.
NP1_HD bool record_window_range(const uint32_t* d, const EvWindow& w, uint32_t* lo, uint32_t* hi, bool* started_before) {
    const uint32_t sf = d[0], sl = d[DESC_NEXT + 1];
    if (sf > sl || d[1] < sf || sl < w.w0 || sf >= w.w0 + w.n) return false;
    *started_before = sf < w.w0;
    *lo = sf > w.w0 ? sf - w.w0 : 0u;
    *hi = (sl < w.w0 + w.n - 1 ? sl : w.w0 + w.n - 1) - w.w0;
    return true;
}

// true: group j of an unchained record holds no disagreeing vote beyond the record's first two (emitted here)
template <class Sink>
NP1_HD bool group_clean(const uint32_t* d, const uint8_t* sqb, const EvWindow& w, const GroupWin& gw, uint32_t j, uint32_t lo,
                        uint32_t hi, bool started_before, Sink& sink) {
    const int32_t ka = (int32_t)(EV_G * j) - 2;
    const uint32_t klast = EV_G * j + 7 < hi ? EV_G * j + 7 : hi;
    const uint32_t ca = ka > (int32_t)lo ? (uint32_t)ka : lo;      // first slot whose symbol matters here
    const uint32_t n = klast - ca + 1;                             // 1..10 symbols to compare
    const uint32_t bits = ((1u << n) - 1u) << (ca - (uint32_t)ka); // (ca >= ka always; ka < 0 only with lo = 0 > ka)
    if (gw.gins[j] & bits) return false;
    const uint32_t g = w.sg[ca];
    const uint32_t nseg = d[2] & 0xffu;
    uint32_t q = 0xffffffffu;
    for (uint32_t t = 0; t < nseg; ++t) {
        const uint32_t off = g - d[DESC_SEG0 + 2 * t], wd = d[DESC_SEG0 + 2 * t + 1];
        const uint32_t len = wd & 0xffffu, qc = wd >> 16;
        if (off < len) {
            if (qc != 0xffffu && off + n <= len) q = qc + off;
            break;
        }
    }
    if (q == 0xffffffffu) return false;
    const uint64_t x = nib10_be(sqb, q) ^ nib10_be(w.dpk, g - w.dpk_g0);
    if (x >> (64 - 4 * n)) return false;
    // clean: the record's first two votes carry shortened contexts (nothing, then one symbol, to their left)
    if (!started_before) {
        for (uint32_t k = lo; k <= lo + 1 && k <= klast; ++k) {
            if (k < EV_G * j) continue;
            const uint32_t s = w.w0 + k;
            if (s < w.own0) continue;   // left-context slot: its symbol equals the draft's, no event
            const uint32_t ctx = k == lo ? (uint32_t)(w.sinfo[k] & 0xf)
                                         : (uint32_t)(w.sinfo[k - 1] & 0xf) << 4 | (uint32_t)(w.sinfo[k] & 0xf);
            if (ctx != w.k0[k]) sink.event(s, ctx);
        }
    }
    return true;
}

// exact symbol of a record at window slot k (0 when the record does not vote there); d = head descriptor
NP1_HD uint32_t group_symbol(const uint32_t* d, const uint32_t* ovf_pool, const uint8_t* sqb, const EvWindow& w, int32_t k,
                             uint32_t lo, uint32_t hi) {
    if (k < (int32_t)lo || k > (int32_t)hi) return 0u;
    const uint32_t s = w.w0 + (uint32_t)k;
    const uint32_t* part = d;
    while (s > part[1]) {   // chained record: the part that covers s (parts are ordered; the head covers most records whole)
        const uint32_t nx = part[DESC_NEXT];
        if (!nx) return 0u;
        part = ovf_pool + (uint64_t)(nx - 1) * DESC_WORDS;
    }
    if (s < part[0]) return 0u;
    const uint32_t info = w.sinfo[k], g = w.sg[k];
    const int32_t jj = (info & SI_INSERT) ? (int32_t)(s - w.soff[g]) - 1 : -1;
    return desc_symbol(part, g, jj, SeqBytes{sqb});
}

// emission rule for an exactly evaluated vote (the same as in record_events)
template <class Sink>
NP1_HD void group_emit(const EvWindow& w, uint32_t k, uint32_t sym, uint32_t ctx, Sink& sink) {
    const uint32_t s = w.w0 + k;
    if (s >= w.own0) {
        if (ctx != w.k0[k]) sink.event(s, ctx);
    } else if (sym != (uint32_t)(w.sinfo[k] & 0xf)) {
        sink.event(s, sym);
    }
}

// sequential reference of the group form (host model): every group of the record, clean test first
template <class Sink>
NP1_HD void record_groups(const uint32_t* d, const uint32_t* ovf_pool, const uint8_t* sqb, const EvWindow& w, const GroupWin& gw,
                          Sink& sink) {
    uint32_t lo, hi;
    bool started_before;
    if (!record_window_range(d, w, &lo, &hi, &started_before)) return;
    const bool chained = (d[2] & DESC_CHAIN) != 0;
    for (uint32_t j = lo / EV_G; j <= hi / EV_G; ++j) {
        if (!chained && group_clean(d, sqb, w, gw, j, lo, hi, started_before, sink)) continue;
        uint32_t p1 = 0, p2 = 0;
        for (uint32_t t = 0; t < EV_GL; ++t) {
            const int32_t k = (int32_t)(EV_G * j) - 2 + (int32_t)t;
            const uint32_t sym = group_symbol(d, ovf_pool, sqb, w, k, lo, hi);
            const uint32_t ctx = p2 << 8 | p1 << 4 | sym;
            if (t >= 2 && k >= (int32_t)lo && k <= (int32_t)hi) group_emit(w, (uint32_t)k, sym, ctx, sink);
            p2 = p1;
            p1 = sym;
        }
    }
}

}  // namespace np1k
