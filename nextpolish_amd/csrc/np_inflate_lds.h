// Raw DEFLATE (RFC 1951) decoder for whole BGZF blocks, ONE LANE PER BLOCK, written for a machine whose lanes run in lockstep (round 6).
//
// np_inflate_lane.h is a sequential decoder per lane: tables in a 12 KB slice of HBM per lane, one load per table lookup and refill, every
// match copied at once.  Measured (profiles/r6_inflate_lds.txt): with 40 000 blocks per launch a launch takes as long as ONE lane needs for
// ONE 64 KiB block, ~42 ms, i.e. ~7 us per symbol -- because the 64 lanes of a wave are each at a different kind of symbol, the wave runs
// the union of all paths in every iteration, and every path that waits for memory (table lookup, refill, each of the copy loops: 3 800 of the
// 5 800 symbols of a BAM block are matches that read back what the lane wrote) stalls all 64 lanes.  Moving the tables to LDS alone changed
// nothing, deferring copies on top of the old structure made it slower (more paths).  So the symbol loop here has a fixed shape with ONE
// place where it waits for memory:
//     (a) a lane with bytes of a match still to copy asks for the next <= 32 of them (loads, no wait);
//     (b) a lane that has no more bytes to ask for decodes its next symbol -- registers and LDS only: the primary tables are 16-bit slots in
//         LDS laid out [slot][lane] (two lanes per dword: at most a two-way bank conflict), codes longer than the primary index are decoded
//         canonically from per-length bounds and the symbols sorted by code, all of them in LDS as bytes; the stream's bytes come out of a window
//         of registers; a literal goes into a register that leaves as an 8-byte store when full (a store is not waited for); a match at a
//         distance of 1 .. 7 whose period is known is written from a register copy of the last eight bytes;
//     (c) the wait: the bytes asked for in (a) arrive -- while (b) ran -- and are stored, exactly their count (what follows them may already
//         be in memory); the input window slides when 16 of its bytes are used (the words it takes in were asked for an iteration ago).
// The code is ordinary C++ over a table policy (`Tab`: rd / wr of a 16-bit slot): the device instantiates it over LDS, the host over a plain
// array -- tests/test_inflate.py runs the host build against zlib over every block type, level and strategy and over damaged streams, and
// under AddressSanitizer with buffers of exactly the streams' sizes (no byte outside [src, src + src_len) is read, none outside
// [dst, dst + dst_len) written).  Returns 0 when exactly dst_len bytes came out of the stream, else an error code (the caller inflates
// refused blocks on the host: a refusal costs time, never correctness).
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
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint16_t __attribute__((aligned(1))) u16u;

// per-lane HBM scratch: code lengths while the tables of a block are built (never touched by the symbol loop)
struct Scratch {
    uint8_t lens[320];
};

// 16-bit slots of a lane's table.  Per alphabet: 2^bits primaries (symbol << 4 | code length, 0 = no code this short); for every code
// length above `bits` the exclusive upper bound of its codes (15-bit, left-justified, in the order the bits arrive), what to add to a code
// to index the symbols sorted by code, and (literal/length alphabet) the index in that order from which the symbols of this length are
// >= 256; and ALL symbols sorted by code, their low eight bits, two to a slot.  (Round 6, third version: only the first 64 long symbols were
// in LDS, the rest in HBM scratch -- and with 64 lanes in lockstep some lane needed one of the rest in most iterations, a trip to HBM for
// the whole wave.  The symbol loop now reads nothing but LDS.)
template <int LB, int DB> struct Layout {
    static constexpr uint32_t NL = 15 - LB, ND = 15 - DB;
    static constexpr uint32_t LIT0 = 0, DIST0 = 1u << LB;
    static constexpr uint32_t LIM_LIT = DIST0 + (1u << DB), OFS_LIT = LIM_LIT + NL, HI_LIT = OFS_LIT + NL, SORT_LIT = HI_LIT + NL;
    static constexpr uint32_t LIM_DIST = SORT_LIT + 144, OFS_DIST = LIM_DIST + ND, SORT_DIST = OFS_DIST + ND;
    static constexpr uint32_t SLOTS = SORT_DIST + 16;
};
// one alphabet's places in the table (hi = 0: no symbol of the alphabet is >= 256)
struct Alpha { uint32_t base, bits, lim, ofs, hi, sort, nsort; };

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

// Canonical code from lens[0 .. n_sym) into the alphabet's places: primaries for the codes of up to `bits` bits (bits <= A.bits: the code-length
// alphabet borrows the literal alphabet's places with 7 bits), bounds, offsets and >= 256 marks for the lengths above A.bits, and the low bytes
// of all symbols sorted by code.  false: over-subscribed code.
template <class Tab>
NPD_HD_CALL bool build(const uint8_t* lens, uint32_t n_sym, uint32_t bits, Tab& tab, const Alpha A) {
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
        if (len > A.bits) {
            // codes of this length, as their first 15 bits arrive: [code << (15 - len), (code + count) << (15 - len))
            tab.wr(A.lim + (len - A.bits - 1), (uint16_t)((code + count[len]) << (15 - len)));      // (<= 32768)
            tab.wr(A.ofs + (len - A.bits - 1), (uint16_t)(index - code));
            if (A.hi) tab.wr(A.hi + (len - A.bits - 1), (uint16_t)(index + count[len]));             // (until a symbol >= 256 of this length shows)
        }
        index += count[len];
    }
    const uint32_t slots = 1u << bits;
    for (uint32_t i = 0; i < slots; ++i) tab.wr(A.base + i, 0);
    uint32_t marked = 0;      // lengths whose first symbol >= 256 has been seen
    for (uint32_t s = 0; s < n_sym; ++s) {
        const uint32_t len = lens[s] & 15u;
        if (!len) continue;
        const uint32_t c = next_code[len]++;
        const uint32_t at = next_index[len]++;
        if (at < A.nsort) {      // (always, for a code that is not over-subscribed)
            const uint32_t w = tab.rd(A.sort + (at >> 1));
            tab.wr(A.sort + (at >> 1), (uint16_t)((at & 1u) ? (w & 0x00ffu) | (s & 0xffu) << 8 : (w & 0xff00u) | (s & 0xffu)));
        }
        if (A.hi && s >= 256 && len > A.bits && !(marked >> len & 1u)) { marked |= 1u << len; tab.wr(A.hi + (len - A.bits - 1), (uint16_t)at); }
        if (len <= bits) {
            const uint32_t r = rev32(c) >> (32 - len);
            const uint16_t e = (uint16_t)(s << 4 | len);
            for (uint32_t i = r; i < slots; i += 1u << len) tab.wr(A.base + i, e);
        }
    }
    return true;
}

// A code longer than the primary index of BITS bits: `peek` = the next 15 bits of the stream.  Returns symbol << 4 | length, or 0 when no
// code matches.  All of it LDS (the bounds are read in one go).
template <int BITS, class Tab>
NPD_HD uint32_t decode_long(uint32_t peek, const Tab& tab, const Alpha A) {
    const uint32_t v = rev32(peek) >> 17;      // the 15 bits in the order they arrived, first bit on top
    uint32_t lim[15 - BITS];
NPD_UNROLL
    for (int k = 0; k < 15 - BITS; ++k) lim[k] = tab.rd(A.lim + k);
    uint32_t k = 0;                             // number of lengths whose codes all lie below v
NPD_UNROLL
    for (int j = 0; j < 15 - BITS; ++j) k += v >= lim[j] ? 1u : 0u;      // (the bounds never decrease with the length)
    if (k >= 15u - BITS) return 0;
    const uint32_t len = BITS + 1 + k;
    const uint32_t at = (uint16_t)(tab.rd(A.ofs + k) + (v >> (15 - len)));
    if (at >= A.nsort) return 0;
    const uint32_t w = tab.rd(A.sort + (at >> 1));
    uint32_t sym = (at & 1u) ? w >> 8 : w & 0xffu;
    if (A.hi && at >= tab.rd(A.hi + k)) sym |= 256u;
    return sym << 4 | len;
}

struct __attribute__((packed, aligned(1))) Pair64 { uint64_t a, b; };
NPD_HD void store16(uint8_t* p, uint64_t a, uint64_t b) { *reinterpret_cast<Pair64*>(p) = Pair64{a, b}; }

// The bit reader.  The stream's bytes come through a WINDOW OF REGISTERS: w0..w4 = the stream's bytes [base, base + 40), n0 n1 = the 16
// behind them.  A refill takes its eight bytes out of the window with shifts and selects -- no memory access; the window slides by 16 bytes
// (slide(), called where the symbol loop waits for memory anyway) and then asks for the next 16.
struct Bits {
    const uint8_t* src;      // the stream
    uint32_t len;            // its length
    uint32_t base;           // offset of w0 in the stream
    uint32_t pos;            // offset from `base` of the next byte that is not yet in `buf` (< 32 whenever refill runs)
    uint64_t w0, w1, w2, w3, w4, n0, n1;
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
        n0 = load(40); n1 = load(48);
    }
    // tops the buffer up to >= 56 bits (pos < 32 on entry; it grows by at most 7)
    NPD_HD void refill() {
        // (the word is picked with masks, not with a chain of selects: a compiler that sees "select among five variables" turns the window
        // into an indexed array in scratch memory -- and every refill into a load the wave has to wait for)
        const uint32_t i = pos >> 3, sh = (pos & 7u) * 8u;
        const uint64_t m0 = 0ull - (uint64_t)(i == 0), m1 = 0ull - (uint64_t)(i == 1), m2 = 0ull - (uint64_t)(i == 2), m3 = 0ull - (uint64_t)(i >= 3);
        const uint64_t lo = (w0 & m0) | (w1 & m1) | (w2 & m2) | (w3 & m3);
        const uint64_t hi = (w1 & m0) | (w2 & m1) | (w3 & m2) | (w4 & m3);
        const uint64_t v = lo >> sh | (hi << 1) << (63u - sh);
        buf |= v << cnt;
        pos += (63 - cnt) >> 3;
        cnt |= 56;
    }
    // once pos >= 16: two words out, the two asked for at the previous slide in, the next two asked for.  Between two calls at most 16
    // bytes may be consumed (a symbol with its extra bits and a distance with its: <= 6 bytes, two refills of <= 7).
    // (The two words behind the window are asked for on EVERY call, slid or not -- the same addresses again when not: a load under a
    // condition has to be merged with the old value of its register, and the compiler does that with a wait right behind the load.)
    // In the symbol loop the two halves are called apart: take() right where the wave has waited for memory (it reads n0 n1), ask() behind the
    // stores of that place (a load waits for nothing, but whoever next waits for it also waits for the stores issued before it).
    NPD_HD void slide_take() {
        if (pos >= 16) {
            w0 = w2; w1 = w3; w2 = w4; w3 = n0; w4 = n1;
            base += 16;
            pos -= 16;
        }
    }
    NPD_HD void slide_ask() {
#if defined(__HIP_DEVICE_COMPILE__)
        // one 16-byte load (one transaction per lane) instead of two of eight with a clamp each; what lies behind the stream's end is never consumed
        uint32_t at = base + 40;
        if (at > len) at = len;
        const Pair64 v = *reinterpret_cast<const Pair64*>(src + at);
        n0 = v.a;
        n1 = v.b;
#else
        n0 = load(base + 40);
        n1 = load(base + 48);
#endif
    }
    NPD_HD void slide() { slide_take(); slide_ask(); }
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

// 32 bytes at p, of which the caller uses the first n <= 32.  On the device the output buffer has slack behind its last block, so whole
// words are read whatever n is (the second pair only when n > 16); the host build never reads at or beyond `limit`.
NPD_HD void load32(const uint8_t* p, const uint8_t* limit, uint32_t n, uint64_t* a0, uint64_t* a1, uint64_t* a2, uint64_t* a3) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)limit;
    *a0 = *reinterpret_cast<const u64u*>(p); *a1 = *reinterpret_cast<const u64u*>(p + 8);
    if (n > 16u) { *a2 = *reinterpret_cast<const u64u*>(p + 16); *a3 = *reinterpret_cast<const u64u*>(p + 24); }      // (a transaction per lane that asks)
#else
    (void)n;
    uint64_t w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 32 && p + i < limit; ++i) w[i >> 3] |= (uint64_t)p[i] << (8 * (i & 7));
    *a0 = w[0]; *a1 = w[1]; *a2 = w[2]; *a3 = w[3];
#endif
}

// src[0 .. src_len): raw DEFLATE stream; dst[0 .. dst_len): its output; tab: Layout<LB, DB>::SLOTS 16-bit slots; sc: the lane's scratch.
// (DBG: timing experiments only -- bit 0 drops the literal stores, bit 1 the match copies: wrong output)
template <int LB, int DB, class Tab, int DBG = 0>
NPD_HD int inflate_block(const uint8_t* src, uint32_t src_len, uint8_t* dst, uint32_t dst_len, Tab& tab, Scratch* sc) {
    typedef Layout<LB, DB> Y;
    constexpr int CLB = 7 <= LB ? 7 : LB;      // primary bits of the code-length alphabet (its codes have up to 7 bits)
    const Alpha AL{Y::LIT0, (uint32_t)LB, Y::LIM_LIT, Y::OFS_LIT, Y::HI_LIT, Y::SORT_LIT, 288};
    const Alpha AD{Y::DIST0, (uint32_t)DB, Y::LIM_DIST, Y::OFS_DIST, 0, Y::SORT_DIST, 32};
    Bits b;
    b.start(src, src_len);
    uint8_t* out = dst;                    // where the next decoded byte goes, less the pending literals
    uint8_t* const out_end = dst + dst_len;
    uint64_t pend = 0;                     // literals not yet stored: bytes out[0 .. npend)
    uint32_t npend = 0;
    // The last bytes of the output so far, pending ones included, newest on top; the top `nvalid` bytes are known.
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
    for (;;) {
        b.refill();
        const uint32_t final_block = b.take(1), type = b.take(2);
        if (type == 0) {   // stored: skip to the byte boundary, LEN / NLEN, bytes
            b.drop(b.cnt & 7u);
            b.refill();
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ 0xffffu) != nlen) return 2;
            if (b.taken > 8u * src_len) return 17;
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
            if (!build(lens, 288, LB, tab, AL)) return 11;
            for (int i = 0; i < 32; ++i) lens[i] = 5;
            if (!build(lens, 32, DB, tab, AD)) return 12;
        } else if (type == 2) {
            b.slide();
            const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
            if (hlit > 286 || hdist > 30) return 5;
            uint8_t* const cl = lens + 300;      // 19 code-length code lengths, behind the 286 + 30 lengths they describe
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            for (uint32_t i = 0; i < hclen; ++i) {
                if (b.cnt < 3) { b.refill(); b.slide(); }
                cl[kClOrder[i]] = (uint8_t)b.take(3);
            }
            // the code-length alphabet borrows the literal alphabet's places (built afterwards)
            if (!build(cl, 19, CLB, tab, AL)) return 6;
            uint32_t n = 0;
            while (n < hlit + hdist) {
                if (b.cnt < 32) { b.refill(); b.slide(); }
                uint32_t e = tab.rd(Y::LIT0 + b.peek(CLB));
                if (!(e & 15u) && CLB < 7) e = decode_long<LB>(b.peek(15), tab, AL);
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
            if (!build(lens + hlit, hdist, DB, tab, AD)) return 12;
            if (!build(lens, hlit, LB, tab, AL)) return 11;
        } else {
            return 4;
        }
        b.slide();
        // ---- the symbols of the block: (a) ask, (b) decode, (c) wait and store -- see the head of this file.
        // c_*: the match being copied (bytes not yet asked for); f_*: the piece asked for in (a), stored in (c).
        uint32_t c_rem = 0, c_off = 0, f_n = 0;
        uint8_t *c_dst = nullptr, *f_d = nullptr;
        uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        bool ended = false;
        int err = 0;
        for (;;) {
            // (a) the next piece of the match in hand: at most 32 bytes, and no more than the distance (the source of a piece must be in
            // memory when it is asked for: everything below c_dst is, after the (c) of the iteration before)
            if (c_rem) {
                uint32_t piece = c_rem < 32 ? c_rem : 32;
                if (c_off < piece) piece = c_off;
                if (!(DBG & 2)) load32(c_dst - c_off, out_end, piece, &a0, &a1, &a2, &a3);
                f_n = piece;
                f_d = c_dst;
                c_dst += piece;
                c_rem -= piece;
            }
            // (b) the next symbol, for a lane that has nothing left to ask for
            if (!c_rem && !ended) {
                if (b.cnt < 32) b.refill();
                uint32_t e = tab.rd(Y::LIT0 + b.peek(LB));
                if (!(e & 15u)) e = decode_long<LB>(b.peek(15), tab, AL);
                if (!(e & 15u)) { err = 14; break; }
                b.drop(e & 15u);
                const uint32_t sym = e >> 4;
                if (sym < 256) {
                    if ((size_t)(out_end - out) <= npend) { err = 13; break; }
                    pend |= (uint64_t)sym << (8 * npend);
                    last8 = last8 >> 8 | (uint64_t)sym << 56;
                    nvalid = nvalid < 8 ? nvalid + 1 : 8;
                    if (++npend == 8) { if (!(DBG & 1)) *reinterpret_cast<u64u*>(out) = pend; out += 8; pend = 0; npend = 0; }
                } else if (sym == 256) {
                    ended = true;
                } else {
                    if (sym > 285) { err = 14; break; }
                    const uint32_t len = len_base(sym - 257) + b.take(len_extra(sym - 257));
                    if (b.cnt < 32) b.refill();
                    uint32_t d = tab.rd(Y::DIST0 + b.peek(DB));
                    if (!(d & 15u)) d = decode_long<DB>(b.peek(15), tab, AD);
                    if (!(d & 15u) || (d >> 4) > 29) { err = 15; break; }
                    b.drop(d & 15u);
                    const uint32_t off = dist_base(d >> 4) + b.take(dist_extra(d >> 4));
                    if (off > (size_t)(out - dst) + npend || len > (size_t)(out_end - out) - npend) { err = 16; break; }
                    flush();
                    if (off <= nvalid && off < 8) {
                        // a run: the period is the top `off` bytes of last8; whole words, each starting at the head of a period (what a word
                        // writes beyond the run is this lane's own future output), the end of the block byte by byte
                        const uint64_t x = off == 1 ? (last8 >> 56) * 0x0101010101010101ull : periodic(last8 >> (8 * (8 - off)), off, 0);
                        if (!(DBG & 2)) {
                            const uint32_t step = off == 3 || off == 6 ? 6u : off == 5 ? 5u : off == 7 ? 7u : 8u;
                            uint32_t done = 0;
                            for (; done < len && (size_t)(out_end - (out + done)) >= 8; done += step) *reinterpret_cast<u64u*>(out + done) = x;
                            if (done < len) {
                                uint32_t k = done % off;
                                for (uint32_t i = done; i < len; ++i) { out[i] = (uint8_t)(x >> (8 * k)); if (++k == off) k = 0; }
                            }
                        }
                        if (len >= 8) { last8 = off == 1 ? x : periodic(x, off, (len - 8) % off); nvalid = 8; }
                        else { last8 = last8 >> (8 * len) | x << (8 * (8 - len)); nvalid = nvalid + len < 8 ? nvalid + len : 8; }
                    } else {
                        // a copy from memory: from the next (a) on
                        c_rem = len;
                        c_off = off;
                        c_dst = out;
                        nvalid = 0;
                    }
                    out += len;
                }
            }
            // (c) the one wait: the window slides (the words it takes in were asked for an iteration ago), the piece asked for in (a) is
            // stored, exactly its bytes, and the window's next words are asked for
            b.slide_take();
            if (f_n) {
                if (!(DBG & 2)) {
                    // exactly f_n <= 32 bytes in as few stores as their count has bits (16 + 8 + 4 + 2 + 1): every store instruction is a transaction
                    // per lane, and the decoder runs at about half of L2's transaction rate (profiles/r6_inflate_lds.txt item 10); byte by byte the
                    // last n < 8 bytes alone were up to seven
                    uint8_t* d = f_d;
                    const uint32_t n = f_n;
                    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3;
                    if (n >= 16u) { store16(d, w0, w1); d += 16; w0 = w2; w1 = w3; }
                    if (n == 32u) { store16(d, w0, w1); }
                    else {
                        if (n & 8u) { *reinterpret_cast<u64u*>(d) = w0; d += 8; w0 = w1; }
                        if (n & 4u) { *reinterpret_cast<u32u*>(d) = (uint32_t)w0; d += 4; w0 >>= 32; }
                        if (n & 2u) { *reinterpret_cast<u16u*>(d) = (uint16_t)w0; d += 2; w0 >>= 16; }
                        if (n & 1u) *d = (uint8_t)w0;
                    }
                }
                f_n = 0;
            }
            b.slide_ask();
            if (ended && !c_rem) break;
        }
        if (err) return err;
        if (b.taken > 8u * src_len) return 17;
        if (final_block) break;
    }
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
