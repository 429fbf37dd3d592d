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

}  // namespace np1k
