// Raw DEFLATE (RFC 1951) decoder for whole BGZF blocks, ONE LANE PER BLOCK, DECODE TABLES IN LDS (round 6).
//
// np_inflate_lane.h gave every lane a 12 KB table slice in HBM and counted on the number of lanes in flight: with 40 000 blocks per
// launch the hot entries of all lanes (~5 KB each) are 200 MB -- beyond every cache -- so each symbol costs a dependent trip to MALL or
// HBM (measured: 77 ms for 40 k blocks, ~7 000 cycles per symbol and lane, 40 GB/s).  Here the tables a symbol
// normally needs are small enough to live in LDS:
//   * one 16-bit entry per primary slot (symbol << 4 | code length), 2^LB slots for literals / lengths and 2^DB for distances, laid
//     out [slot][lane] so that the 64 lanes of a wave, each reading a slot of its own table, meet different banks (two lanes per
//     dword: at most a two-way conflict);
//   * codes longer than the primary index are decoded canonically: per code length the upper bound of its codes and an offset (a few
//     more LDS slots of the lane, read in one go), then the symbol from the symbols sorted by code, in a 1 KB per-lane slice of HBM
//     scratch that stays in L2 (a few waves per CU: tens of thousands of lanes in flight, of whose slices a lane touches a few lines);
//   * the bit buffer is refilled from a word loaded one refill ahead, literals are gathered in a register and leave as 8-byte stores;
//   * the last eight output bytes are kept in a register, and a match at a distance of 1 .. 7 -- the runs base qualities consist of -- is
//     written from there: measured (profiles/r6_inflate_lds.txt), copies that read back what the lane had just stored were 60 % of the
//     decoder's time, each one a trip to L2 and back behind the stores before it.
// A symbol then costs one LDS lookup plus ~40 instructions instead of a trip to HBM.  The code is ordinary C++ over a table policy
// (`Tab`: rd / wr of a 16-bit slot): the device instantiates it over LDS, the host over a plain array -- tests/test_inflate.py runs the
// host build against zlib over every block type, level and strategy and over damaged streams, and under AddressSanitizer with buffers
// of exactly the streams' sizes (no byte outside [src, src + src_len) is read, none outside [dst, dst + dst_len) written).
// Returns 0 when exactly dst_len bytes came out of the stream, else an error code (the caller inflates refused blocks on the host).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define NPD_HD __host__ __device__ __forceinline__
#define NPD_HD_CALL __host__ __device__ __noinline__
#define NPD_UNROLL _Pragma("unroll")
#else
#define NPD_UNROLL
#define NPD_HD inline
#define NPD_HD_CALL inline
#endif

namespace nplds {

typedef uint64_t __attribute__((aligned(1))) u64u;

// per-lane HBM scratch: code lengths while the tables are built, and the symbols of both alphabets sorted by code (read only for codes
// longer than the primary index)
struct Scratch {
    uint8_t lens[320];
    uint16_t sorted_lit[288];
    uint16_t sorted_dist[32];
};

// 16-bit slots of a lane's table: [0, 2^LB) literal / length primaries, [2^LB, 2^LB + 2^DB) distance primaries, then for every code length
// above the primary index the exclusive upper bound of its codes (15-bit, left-justified, in the order the bits arrive) and what to add
// to a code to index `sorted` -- literals / lengths first, then distances
template <int LB, int DB> struct Layout {
    static constexpr uint32_t LIT0 = 0, DIST0 = 1u << LB, NL = 15 - LB, ND = 15 - DB;
    static constexpr uint32_t LIM_LIT = DIST0 + (1u << DB), OFS_LIT = LIM_LIT + NL, LIM_DIST = OFS_LIT + NL, OFS_DIST = LIM_DIST + ND, SLOTS = OFS_DIST + ND;
};

NPD_HD uint32_t len_base(uint32_t i) {      // RFC 1951 3.2.5, computed
    if (i < 8) return 3 + i;
    if (i == 28) return 258;
    const uint32_t xb = (i - 4) >> 2;
    return 3 + ((4 + ((i - 4) & 3)) << xb);
}
NPD_HD uint32_t len_extra(uint32_t i) { return i < 8 || i == 28 ? 0 : (i - 4) >> 2; }
NPD_HD uint32_t dist_base(uint32_t i) {
    if (i < 4) return 1 + i;
    const uint32_t xb = (i - 2) >> 1;
    return 1 + ((2 + (i & 1)) << xb);
}
NPD_HD uint32_t dist_extra(uint32_t i) { return i < 4 ? 0 : (i - 2) >> 1; }

NPD_HD uint32_t rev32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v);
#else
    v = (v >> 16) | (v << 16);
    v = ((v & 0xff00ff00u) >> 8) | ((v & 0x00ff00ffu) << 8);
    v = ((v & 0xf0f0f0f0u) >> 4) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v & 0xccccccccu) >> 2) | ((v & 0x33333333u) << 2);
    v = ((v & 0xaaaaaaaau) >> 1) | ((v & 0x55555555u) << 1);
    return v;
#endif
}

// Canonical code from lens[0 .. n_sym): primary slots [base, base + 2^bits) of `tab` for the codes of up to `bits` bits; for the lengths
// above aux_bits (the width of the region's own primary index; bits <= aux_bits) the bound and offset slots at lim0 / ofs0; the sorted
// symbols.  false: over-subscribed code.
template <class Tab>
NPD_HD_CALL bool build(const uint8_t* lens, uint32_t n_sym, uint32_t bits, Tab& tab, uint32_t base, uint16_t* sorted, uint32_t aux_bits, uint32_t lim0, uint32_t ofs0) {
    uint32_t count[16];
    for (int i = 0; i < 16; ++i) count[i] = 0;
    for (uint32_t s = 0; s < n_sym; ++s) ++count[lens[s] & 15u];
    count[0] = 0;
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
        left = (left << 1) - (int)count[len];
        if (left < 0) return false;
    }
    uint32_t next_code[16], next_index[16];
    uint32_t code = 0, index = 0;
    next_code[0] = next_index[0] = 0;
    for (uint32_t len = 1; len <= 15; ++len) {
        code = (code + count[len - 1]) << 1;
        next_code[len] = code;
        next_index[len] = index;
        if (len > aux_bits) {
            // codes of this length, as their first 15 bits arrive: [code << (15 - len), (code + count) << (15 - len))
            tab.wr(lim0 + (len - aux_bits - 1), (uint16_t)((code + count[len]) << (15 - len)));      // (<= 32768)
            tab.wr(ofs0 + (len - aux_bits - 1), (uint16_t)(index - code));
        }
        index += count[len];
    }
    const uint32_t slots = 1u << bits;
    for (uint32_t i = 0; i < slots; ++i) tab.wr(base + i, 0);
    for (uint32_t s = 0; s < n_sym; ++s) {
        const uint32_t len = lens[s] & 15u;
        if (!len) continue;
        const uint32_t c = next_code[len]++;
        sorted[next_index[len]++] = (uint16_t)s;
        if (len <= bits) {
            const uint32_t r = rev32(c) >> (32 - len);
            const uint16_t e = (uint16_t)(s << 4 | len);
            for (uint32_t i = r; i < slots; i += 1u << len) tab.wr(base + i, e);
        }
    }
    return true;
}

// A code longer than the primary index of BITS bits: `peek` = the next 15 bits of the stream.  Returns symbol << 4 | length, or 0 when no
// code matches.  The bounds are read in one go (independent reads of the lane's slots), the symbol comes from the lane's scratch.
template <int BITS, class Tab>
NPD_HD uint32_t decode_long(uint32_t peek, const Tab& tab, uint32_t lim0, uint32_t ofs0, const uint16_t* sorted) {
    const uint32_t v = rev32(peek) >> 17;      // the 15 bits in the order they arrived, first bit on top
    uint32_t lim[15 - BITS];
NPD_UNROLL
    for (int k = 0; k < 15 - BITS; ++k) lim[k] = tab.rd(lim0 + k);
    uint32_t k = 0;                             // number of lengths whose codes all lie below v
NPD_UNROLL
    for (int j = 0; j < 15 - BITS; ++j) k += v >= lim[j] ? 1u : 0u;      // (the bounds never decrease with the length)
    if (k >= 15u - BITS) return 0;
    const uint32_t len = BITS + 1 + k;
    return (uint32_t)sorted[(uint16_t)(tab.rd(ofs0 + k) + (v >> (15 - len)))] << 4 | len;
}

// The bit reader.  The stream's bytes come through a WINDOW OF REGISTERS: five 8-byte words w0..w4 = the stream's bytes [base, base + 40) and
// the four words behind them (n0..n3), loaded one window ahead.  A refill takes its eight bytes out of the window with shifts and selects --
// no memory access -- and only every 32 bytes of input does the window slide (w0 = w4, w1..w4 = n0..n3, four new loads issued).  Why: on
// this hardware a wait for a load is also a wait for every store issued before it (one in-order counter), so the refill of round 6's first
// versions -- one load per ~3 symbols -- queued behind the output stores again and again (profiles/r6_inflate_lds.txt).
struct Bits {
    const uint8_t* src;      // the stream
    uint32_t len;            // its length
    uint32_t base;           // offset of w0 in the stream
    uint32_t pos;            // offset from `base` of the next byte that is not yet in `buf` (< 32 between refills)
    uint64_t w0, w1, w2, w3, w4, n0, n1, n2, n3;
    uint64_t buf;
    uint32_t cnt;
    uint32_t taken;          // bits consumed so far (to tell a stream that ran past its end)
    // The 8 bytes at offset `at` (clamped to the stream's end).  On the device the 8 bytes at any address up to the end are readable (the
    // compressed blocks of a launch lie in one buffer with a pad behind the last; what is read behind a stream's end is never used by a stream
    // that is intact, and `taken` convicts one that is not); on the host (tests: buffers of exactly the stream's size) they read as zero.
    NPD_HD uint64_t load(uint32_t at) const {
        if (at > len) at = len;
#if defined(__HIP_DEVICE_COMPILE__)
        return *reinterpret_cast<const u64u*>(src + at);
#else
        if (len - at >= 8) return *reinterpret_cast<const u64u*>(src + at);
        uint64_t v = 0;
        for (uint32_t i = 0; at + i < len; ++i) v |= (uint64_t)src[at + i] << (8 * i);
        return v;
#endif
    }
    NPD_HD void start(const uint8_t* s, uint32_t n) {
        src = s; len = n; base = 0; pos = 0; buf = 0; cnt = 0; taken = 0;
        w0 = load(0); w1 = load(8); w2 = load(16); w3 = load(24); w4 = load(32);
        n0 = load(40); n1 = load(48); n2 = load(56); n3 = load(64);
    }
    // tops the buffer up to >= 56 bits
    NPD_HD void refill() {
        const uint32_t i = pos >> 3, sh = (pos & 7u) * 8u;
        const uint64_t lo = i == 0 ? w0 : i == 1 ? w1 : i == 2 ? w2 : w3;
        const uint64_t hi = i == 0 ? w1 : i == 1 ? w2 : i == 2 ? w3 : w4;
        const uint64_t v = sh ? lo >> sh | hi << (64u - sh) : lo;
        buf |= v << cnt;
        pos += (63 - cnt) >> 3;
        cnt |= 56;
        if (pos >= 32) {      // slide: the words loaded a window ago become the window, the next four are asked for
            w0 = w4; w1 = n0; w2 = n1; w3 = n2; w4 = n3;
            base += 32;
            pos -= 32;
            n0 = load(base + 40); n1 = load(base + 48); n2 = load(base + 56); n3 = load(base + 64);
        }
    }
    NPD_HD uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    NPD_HD void drop(uint32_t n) { buf >>= n; cnt -= n; taken += n; }
    NPD_HD uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

// eight bytes of the endless repetition of the low `period` bytes of p (1 <= period < 8), starting `phase` bytes into it (phase < period)
NPD_HD uint64_t periodic(uint64_t p, uint32_t period, uint32_t phase) {
    uint64_t w = 0;
    uint32_t k = phase;
NPD_UNROLL
    for (int j = 0; j < 8; ++j) {
        w |= ((p >> (8 * k)) & 0xffull) << (8 * j);
        if (++k == period) k = 0;
    }
    return w;
}

// src[0 .. src_len): raw DEFLATE stream; dst[0 .. dst_len): its output; tab: Layout<LB, DB>::SLOTS 16-bit slots; sc: the lane's scratch.
// (DBG: timing experiments only -- bit 0 drops the literal stores, bit 1 the match copies: wrong output)
template <int LB, int DB, class Tab, int DBG = 0>
NPD_HD int inflate_block(const uint8_t* src, uint32_t src_len, uint8_t* dst, uint32_t dst_len, Tab& tab, Scratch* sc) {
    typedef Layout<LB, DB> Y;
    constexpr int CLB = 7 <= LB ? 7 : LB;      // primary bits of the code-length alphabet (its codes have up to 7 bits)
    Bits b;
    b.start(src, src_len);
    uint8_t* out = dst;                    // everything below `out` is in memory
    uint8_t* const out_end = dst + dst_len;
    uint64_t pend = 0;                     // literals not yet stored: bytes out[0 .. npend)
    uint32_t npend = 0;
    // The last bytes of the output so far, pending ones included, newest on top; the top `nvalid` bytes are known.  A match at a distance
    // of 1 .. 7 -- the runs base qualities are made of -- is written from here: no load from memory that was stored a moment ago (on the
    // device such a load waits for the stores before it, a trip to L2 and back per match).
    uint64_t last8 = 0;
    uint32_t nvalid = 0;
    const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t* const lens = sc->lens;
    // the pending literals to memory (an 8-byte store when that stays inside the block: the bytes above the pending ones are this lane's
    // own future output, written again later)
    auto flush = [&]() {
        if (!npend) return;
        if (!(DBG & 1)) {
            if ((size_t)(out_end - out) >= 8) *reinterpret_cast<u64u*>(out) = pend;
            else for (uint32_t i = 0; i < npend; ++i) out[i] = (uint8_t)(pend >> (8 * i));
        }
        out += npend;
        pend = 0;
        npend = 0;
    };
    // Copies whose source bytes have been asked for but not yet stored (at most two).  A match that reads memory costs a trip to L2 or HBM
    // and back, and 3 800 of the 5 800 symbols of a BAM block are such matches (profiles/r6_inflate_lds.txt): instead of waiting for each, the
    // loads of a short match are issued and the decoder goes on; when a second one has been issued, or something needs the bytes (a match that
    // reads them, the end of the block), ONE wait brings both in and they are stored -- exactly their bytes, because what lies behind them
    // may already be in memory.
    uint8_t *pc0_d = nullptr, *pc1_d = nullptr;
    uint32_t pc0_n = 0, pc1_n = 0, npc = 0;
    uint64_t pc0_a = 0, pc0_b = 0, pc0_c = 0, pc0_e = 0, pc1_a = 0, pc1_b = 0, pc1_c = 0, pc1_e = 0;
    auto store_exact = [&](uint8_t* d, uint32_t n, uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3) {
        if (DBG & 2) return;
        uint64_t tail = a0;
        if (n >= 8) { *reinterpret_cast<u64u*>(d) = a0; d += 8; n -= 8; tail = a1; }
        if (n >= 8) { *reinterpret_cast<u64u*>(d) = a1; d += 8; n -= 8; tail = a2; }
        if (n >= 8) { *reinterpret_cast<u64u*>(d) = a2; d += 8; n -= 8; tail = a3; }
        if (n >= 8) { *reinterpret_cast<u64u*>(d) = a3; d += 8; n -= 8; tail = 0; }
        for (uint32_t i = 0; i < n; ++i) d[i] = (uint8_t)(tail >> (8 * i));
    };
    auto drain = [&]() {
        if (npc >= 1) store_exact(pc0_d, pc0_n, pc0_a, pc0_b, pc0_c, pc0_e);
        if (npc >= 2) store_exact(pc1_d, pc1_n, pc1_a, pc1_b, pc1_c, pc1_e);
        npc = 0;
    };
    for (;;) {
        b.refill();
        const uint32_t final_block = b.take(1), type = b.take(2);
        if (type == 0) {   // stored: skip to the byte boundary, LEN / NLEN, bytes
            b.drop(b.cnt & 7u);
            b.refill();
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ 0xffffu) != nlen) return 2;
            if (b.taken > 8u * src_len) return 17;
            drain();
            flush();
            const uint32_t at = b.taken >> 3;      // (byte aligned here)
            if (src_len - at < len || (size_t)(out_end - out) < len) return 3;
            for (uint32_t i = 0; i < len; ++i) out[i] = src[at + i];
            out += len;
            if (len) nvalid = 0;
            const uint32_t done = b.taken + 8u * len;
            b.start(src + at + len, src_len - at - len);
            b.taken = done;
            if (final_block) break;
            continue;
        } else if (type == 1) {
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            if (!build(lens, 288, LB, tab, Y::LIT0, sc->sorted_lit, LB, Y::LIM_LIT, Y::OFS_LIT)) return 11;
            for (int i = 0; i < 32; ++i) lens[i] = 5;
            if (!build(lens, 32, DB, tab, Y::DIST0, sc->sorted_dist, DB, Y::LIM_DIST, Y::OFS_DIST)) return 12;
        } else if (type == 2) {
            const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
            if (hlit > 286 || hdist > 30) return 5;
            uint8_t* const cl = lens + 300;      // 19 code-length code lengths, behind the 286 + 30 lengths they describe
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            for (uint32_t i = 0; i < hclen; ++i) {
                if (b.cnt < 3) b.refill();
                cl[kClOrder[i]] = (uint8_t)b.take(3);
            }
            // the code-length alphabet borrows the head of the literal table, the literal alphabet's bound / offset slots and the distance
            // alphabet's sorted symbols (all three are built afterwards)
            if (!build(cl, 19, CLB, tab, Y::LIT0, sc->sorted_dist, LB, Y::LIM_LIT, Y::OFS_LIT)) return 6;
            uint32_t n = 0;
            while (n < hlit + hdist) {
                if (b.cnt < 32) b.refill();
                uint32_t e = tab.rd(Y::LIT0 + b.peek(CLB));
                if (!(e & 15u) && CLB < 7) e = decode_long<LB>(b.peek(15), tab, Y::LIM_LIT, Y::OFS_LIT, sc->sorted_dist);
                if (!(e & 15u)) return 7;
                b.drop(e & 15u);
                const uint32_t sym = e >> 4;
                if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16) { if (!n) return 8; val = lens[n - 1]; rep = 3 + b.take(2); }
                else if (sym == 17) rep = 3 + b.take(3);
                else rep = 11 + b.take(7);
                if (n + rep > hlit + hdist) return 9;
                for (uint32_t i = 0; i < rep; ++i) lens[n + i] = (uint8_t)val;
                n += rep;
            }
            if (b.taken > 8u * src_len || lens[256] == 0) return 10;
            if (!build(lens + hlit, hdist, DB, tab, Y::DIST0, sc->sorted_dist, DB, Y::LIM_DIST, Y::OFS_DIST)) return 12;
            if (!build(lens, hlit, LB, tab, Y::LIT0, sc->sorted_lit, LB, Y::LIM_LIT, Y::OFS_LIT)) return 11;
        } else {
            return 4;
        }
        // ---- the symbols of the block.  With >= 32 bits in the buffer a literal / length code and its extra bits (<= 20) or a distance
        // code and its extra bits (<= 28) can be taken without looking at the count again.
        for (;;) {
            if (b.cnt < 32) b.refill();
            uint32_t e = tab.rd(Y::LIT0 + b.peek(LB));
            if (!(e & 15u)) {
                e = decode_long<LB>(b.peek(15), tab, Y::LIM_LIT, Y::OFS_LIT, sc->sorted_lit);
                if (!(e & 15u)) return 14;
            }
            b.drop(e & 15u);
            const uint32_t sym = e >> 4;
            if (sym < 256) {
                if ((size_t)(out_end - out) <= npend) return 13;
                pend |= (uint64_t)sym << (8 * npend);
                last8 = last8 >> 8 | (uint64_t)sym << 56;
                nvalid = nvalid < 8 ? nvalid + 1 : 8;
                if (++npend == 8) { if (!(DBG & 1)) *reinterpret_cast<u64u*>(out) = pend; out += 8; pend = 0; npend = 0; }
                continue;
            }
            if (sym == 256) break;
            if (sym > 285) return 14;
            const uint32_t len = len_base(sym - 257) + b.take(len_extra(sym - 257));
            if (b.cnt < 32) b.refill();
            uint32_t d = tab.rd(Y::DIST0 + b.peek(DB));
            if (!(d & 15u)) {
                d = decode_long<DB>(b.peek(15), tab, Y::LIM_DIST, Y::OFS_DIST, sc->sorted_dist);
                if (!(d & 15u)) return 15;
            }
            b.drop(d & 15u);
            if ((d >> 4) > 29) return 15;
            const uint32_t off = dist_base(d >> 4) + b.take(dist_extra(d >> 4));
            if (off > (size_t)(out - dst) + npend || len > (size_t)(out_end - out) - npend) return 16;
            if (off <= nvalid && off < 8) {
                // ---- a run: the period is the top `off` bytes of last8
                flush();
                const uint64_t x = off == 1 ? (last8 >> 56) * 0x0101010101010101ull : periodic(last8 >> (8 * (8 - off)), off, 0);
                if (!(DBG & 2)) {
                    // whole words, each starting at the head of a period; what a word writes beyond the run is this lane's own future output
                    const uint32_t step = off == 3 || off == 6 ? 6u : off == 5 ? 5u : off == 7 ? 7u : 8u;
                    uint32_t done = 0;
                    for (; done < len && (size_t)(out_end - (out + done)) >= 8; done += step) *reinterpret_cast<u64u*>(out + done) = x;
                    if (done < len) {      // the end of the block: byte by byte
                        uint32_t k = done % off;
                        for (uint32_t i = done; i < len; ++i) { out[i] = (uint8_t)(x >> (8 * k)); if (++k == off) k = 0; }
                    }
                }
                out += len;
                if (len >= 8) { last8 = off == 1 ? x : periodic(x, off, (len - 8) % off); nvalid = 8; }
                else { last8 = last8 >> (8 * len) | x << (8 * (8 - len)); nvalid = nvalid + len < 8 ? nvalid + len : 8; }
                continue;
            }
            nvalid = 0;
            // ---- a copy from memory.  Short, not overlapping itself, its 32 source bytes inside the block: deferred (see above).
            {
                const uint8_t* from = out + npend - off;
                if (len <= 32 && off >= len && (size_t)(out_end - from) >= 32) {
                    bool dep = false;      // does it read bytes of a pending copy?
                    if (npc >= 1) dep = dep || (from < pc0_d + pc0_n && from + len > pc0_d);
                    if (npc >= 2) dep = dep || (from < pc1_d + pc1_n && from + len > pc1_d);
                    if (dep || npc == 2) drain();
                    if (from + len > out) flush();      // it reads pending literals: they go to memory first
                    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                    if (!(DBG & 2)) {
                        a0 = *reinterpret_cast<const u64u*>(from); a1 = *reinterpret_cast<const u64u*>(from + 8);
                        a2 = *reinterpret_cast<const u64u*>(from + 16); a3 = *reinterpret_cast<const u64u*>(from + 24);
                    }
                    flush();                              // (behind the loads: they do not queue behind this store)
                    if (npc == 0) { pc0_d = out; pc0_n = len; pc0_a = a0; pc0_b = a1; pc0_c = a2; pc0_e = a3; }
                    else { pc1_d = out; pc1_n = len; pc1_a = a0; pc1_b = a1; pc1_c = a2; pc1_e = a3; }
                    ++npc;
                    out += len;
                    continue;
                }
            }
            // ---- every other copy from memory, at once (as in np_inflate_lane.h: sources far enough away four words at a time)
            drain();
            flush();
            const uint8_t* from = out - off;
            const size_t room = (size_t)(out_end - out);
            if (DBG & 2) {
            } else if (off >= 32 && room >= (size_t)len + 32) {
                uint8_t* o = out;
                const uint8_t* const stop = out + len;
                do {
                    const uint64_t w0 = *reinterpret_cast<const u64u*>(from), w1 = *reinterpret_cast<const u64u*>(from + 8);
                    const uint64_t w2 = *reinterpret_cast<const u64u*>(from + 16), w3 = *reinterpret_cast<const u64u*>(from + 24);
                    *reinterpret_cast<u64u*>(o) = w0; *reinterpret_cast<u64u*>(o + 8) = w1;
                    *reinterpret_cast<u64u*>(o + 16) = w2; *reinterpret_cast<u64u*>(o + 24) = w3;
                    from += 32;
                    o += 32;
                } while (o < stop);
            } else if (off >= 8 && room >= (size_t)len + 8) {   // whole words; the slack bytes are overwritten by what follows
                uint8_t* o = out;
                const uint8_t* const stop = out + len;
                do {
                    *reinterpret_cast<u64u*>(o) = *reinterpret_cast<const u64u*>(from);
                    from += 8;
                    o += 8;
                } while (o < stop);
            } else if (off < 8 && (size_t)(out - dst) >= 8) {   // a period of 1 .. 7 bytes whose bytes are not in last8: the last eight bytes in memory hold it
                const uint64_t tail = *reinterpret_cast<const u64u*>(out - 8);
                const uint32_t first = 8u - off;              // byte of `tail` that is from[0]
                uint32_t k = 0;
                for (uint32_t i = 0; i < len; ++i) {
                    out[i] = (uint8_t)(tail >> (8u * (first + k)));
                    if (++k == off) k = 0;
                }
            } else {
                for (uint32_t i = 0; i < len; ++i) out[i] = from[i];   // near the start or the end of the output: forward, byte by byte
            }
            out += len;
        }
        if (b.taken > 8u * src_len) return 17;
        if (final_block) break;
    }
    drain();
    flush();
    return (b.taken <= 8u * src_len && out == out_end) ? 0 : 19;
}

// the host's table policy (tests; the device's lives in np1_ingest.hip: LDS, [slot][lane])
struct ArrayTab {
    uint16_t* t;
    NPD_HD uint16_t rd(uint32_t i) const { return t[i]; }
    NPD_HD void wr(uint32_t i, uint16_t v) { t[i] = v; }
};

}  // namespace nplds
