// Raw DEFLATE (RFC 1951) decoder for BGZF blocks, written from the RFC for this reader: BGZF inflate is the largest
// single host cost of both polishing paths, and a decoder specialised for "whole block in, whole
// block out" needs no stream state: 64-bit bit buffer refilled eight bytes at a time, one table lookup per symbol
// (11-bit primary table for literals/lengths, 8-bit for distances, second-level tables for longer codes), the decoded
// length known in advance.  Returns false on anything it does not accept (malformed or truncated input, output size
// mismatch); the caller then falls back to zlib, so a false negative costs time, never correctness.
#include "np_inflate.h"

#include <cstdlib>
#include <cstring>

namespace np {
namespace {

constexpr int LIT_BITS = 11, DIST_BITS = 8;
constexpr int MAX_LIT_TABLE = (1 << LIT_BITS) + 1024, MAX_DIST_TABLE = (1 << DIST_BITS) + 512;   // primary + every possible second level

// table entry: bits 0..7 bits to consume (code length; for a length / distance symbol code length + extra bits, so that one shift takes
// both; first level of a long code: the primary bits), 8..11 extra bits, 12..15 kind, 16..31 value (literal byte(s), base length, base
// distance, or second-level offset with its index bits in 8..11).  Literal kinds carry bit 3 (= bit 15 of the entry: one test in the
// symbol loop), bit 0 of the kind then says "two literals in this slot".
enum : uint32_t { K_LENGTH = 1, K_END = 2, K_SUB = 3, K_INVALID = 4, K_LITERAL = 8, K_LITERAL2 = 9 };   // LITERAL2: two literals in one primary slot
constexpr uint32_t E_LITERAL = 0x8000u;
inline uint32_t mk(uint32_t value, uint32_t kind, uint32_t extra, uint32_t nbits) { return value << 16 | kind << 12 | extra << 8 | nbits; }
// base + extra bits of a length / distance entry: `saved` = the bit buffer before the entry's bits were dropped
inline uint32_t with_extra(uint32_t e, uint64_t saved) {
    const uint32_t xb = (e >> 8) & 15u, total = e & 0xffu;
    return (e >> 16) + ((uint32_t)(saved >> (total - xb)) & ((1u << xb) - 1u));
}

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct Rev8 {
    uint8_t t[256];
    Rev8() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t r = 0;
            for (int k = 0; k < 8; ++k) r |= ((i >> k) & 1u) << (7 - k);
            t[i] = (uint8_t)r;
        }
    }
};
const Rev8 kRev8;
inline uint32_t reverse_bits(uint32_t code, int len) {      // len <= 15
    return (((uint32_t)kRev8.t[code & 0xffu] << 8) | kRev8.t[(code >> 8) & 0xffu]) >> (16 - len);
}

// canonical Huffman decode table from code lengths.  is_dist selects the symbol meaning.  Returns false for an
// over-subscribed code; incomplete codes are accepted (unused slots decode as invalid), as zlib does for the cases
// real encoders emit (a single distance code).
// The primary level grows by doubling: codes are entered shortest first into a table of 2^len slots (one store per symbol, at the
// bit-reversed code), and the table is copied onto its upper half whenever the length goes up, so every slot congruent to a code
// modulo 2^len ends up holding it -- ~300 stores and 8 KB of copies instead of 2048 strided stores (every BGZF block brings its own
// tables: this was 8 % of the decoder's time).
inline uint32_t symbol_entry(int s, int len, bool is_dist) {
    if (is_dist) return s >= 30 ? mk(0, K_INVALID, 0, (uint32_t)len) : mk(kDistBase[s], K_LENGTH, kDistExtra[s], (uint32_t)len + kDistExtra[s]);
    if (s < 256) return mk((uint32_t)s, K_LITERAL, 0, (uint32_t)len);
    if (s == 256) return mk(0, K_END, 0, (uint32_t)len);
    if (s <= 285) return mk(kLenBase[s - 257], K_LENGTH, kLenExtra[s - 257], (uint32_t)len + kLenExtra[s - 257]);
    return mk(0, K_INVALID, 0, (uint32_t)len);
}

bool build_table(const uint8_t* lens, int n_sym, int table_bits, bool is_dist, uint32_t* table, int table_cap, bool pair_literals = false) {
    int count[16] = {0};
    for (int i = 0; i < n_sym; ++i) ++count[lens[i]];
    count[0] = 0;
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
        left = (left << 1) - count[len];
        if (left < 0) return false;
    }
    uint32_t next_code[16];
    uint16_t first[17];        // symbols in canonical order (by length, then by value): those of length l are order[first[l] .. first[l + 1])
    uint32_t code = 0;
    first[1] = 0;
    for (int len = 1; len <= 15; ++len) {
        code = (code + (uint32_t)count[len - 1]) << 1;
        next_code[len] = code;
        first[len + 1] = (uint16_t)(first[len] + count[len]);
    }
    uint16_t order[288], at[16];
    for (int len = 1; len <= 15; ++len) at[len] = first[len];
    for (int s = 0; s < n_sym; ++s)
        if (lens[s]) order[at[lens[s]]++] = (uint16_t)s;
    const uint32_t primary = 1u << table_bits;
    table[0] = table[1] = mk(0, K_INVALID, 0, 1);
    int cur = 1;
    for (int len = 1; len <= table_bits; ++len) {
        if (!count[len]) continue;
        for (; cur < len; ++cur) memcpy(table + (1u << cur), table, sizeof(uint32_t) << cur);
        uint32_t c = next_code[len];
        for (uint32_t k = first[len]; k < first[len + 1]; ++k) table[reverse_bits(c++, len)] = symbol_entry(order[k], len, is_dist);
    }
    for (; cur < table_bits; ++cur) memcpy(table + (1u << cur), table, sizeof(uint32_t) << cur);
    // codes longer than the primary index: a second level below every primary slot that heads one, as wide as its longest code
    if (first[16] > first[table_bits + 1]) {
        uint8_t sub_max[1 << LIT_BITS];
        memset(sub_max, 0, primary);
        for (int len = table_bits + 1; len <= 15; ++len) {
            uint32_t c = next_code[len];
            for (uint32_t k = first[len]; k < first[len + 1]; ++k) sub_max[reverse_bits(c++, len) & (primary - 1)] = (uint8_t)len;   // (ascending: the last is the longest)
        }
        uint32_t next_free = primary;
        for (int len = table_bits + 1; len <= 15; ++len) {
            uint32_t c = next_code[len];
            for (uint32_t k = first[len]; k < first[len + 1]; ++k) {
                const uint32_t rc = reverse_bits(c++, len), slot = rc & (primary - 1);
                if (((table[slot] >> 12) & 15u) != K_SUB) {     // (prefix-free: no short code lives here, the slot still says invalid)
                    const uint32_t sub_bits = (uint32_t)sub_max[slot] - (uint32_t)table_bits;
                    if (next_free + (1u << sub_bits) > (uint32_t)table_cap) return false;
                    table[slot] = mk(next_free, K_SUB, sub_bits, (uint32_t)table_bits);
                    for (uint32_t i = 0; i < (1u << sub_bits); ++i) table[next_free + i] = mk(0, K_INVALID, 0, 1);
                    next_free += 1u << sub_bits;
                }
                const uint32_t head = table[slot];
                const uint32_t base = head >> 16, sub_bits = (head >> 8) & 15u;
                const uint32_t e = symbol_entry(order[k], len, is_dist);
                for (uint32_t i = rc >> table_bits; i < (1u << sub_bits); i += 1u << (len - table_bits)) table[base + i] = e;
            }
        }
    }
    if (pair_literals) {   // a primary slot whose bits hold two whole literal codes decodes both at once.  Built pair by pair from the
        // literals of each length (they lead their length's group in canonical order): every slot is stored once, no test per slot
        uint16_t rc[256], sym[256], lit_first[17];
        uint32_t n_lit = 0;
        for (int len = 1; len <= table_bits; ++len) {
            lit_first[len] = (uint16_t)n_lit;
            uint32_t c = next_code[len];
            for (uint32_t k = first[len]; k < first[len + 1] && order[k] < 256; ++k, ++c) {
                rc[n_lit] = (uint16_t)reverse_bits(c, len);
                sym[n_lit++] = order[k];
            }
        }
        lit_first[table_bits + 1] = (uint16_t)n_lit;
        for (int l1 = 1; l1 < table_bits; ++l1)
            for (uint32_t a = lit_first[l1]; a < lit_first[l1 + 1]; ++a)
                for (int l2 = 1; l1 + l2 <= table_bits; ++l2) {
                    const uint32_t step = 1u << (l1 + l2);
                    for (uint32_t b2 = lit_first[l2]; b2 < lit_first[l2 + 1]; ++b2) {
                        const uint32_t e = mk((uint32_t)sym[a] | (uint32_t)sym[b2] << 8, K_LITERAL2, 0, (uint32_t)(l1 + l2));
                        for (uint32_t i = (uint32_t)rc[a] | (uint32_t)rc[b2] << l1; i < primary; i += step) table[i] = e;
                    }
                }
    }
    (void)left;
    return true;
}

struct Bits {
    const uint8_t* in;
    const uint8_t* end;
    uint64_t buf = 0;
    uint32_t cnt = 0;
    bool overrun = false;
    inline void refill() {
        if (end - in >= 8) {
            uint64_t w;
            memcpy(&w, in, 8);
            buf |= w << cnt;
            in += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && in < end) { buf |= (uint64_t)*in++ << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(uint32_t n) {
        if (n > cnt) { overrun = true; n = cnt; }
        buf >>= n;
        cnt -= n;
    }
    inline void drop_fast(uint32_t n) { buf >>= n; cnt -= n; }      // only right behind a refill, for codes known to fit what it left
    inline uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct FixedTables {
    uint32_t lit[MAX_LIT_TABLE], dist[MAX_DIST_TABLE];
    FixedTables() {
        uint8_t l[288], d[32];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(l, 288, LIT_BITS, false, lit, MAX_LIT_TABLE, true);
        build_table(d, 32, DIST_BITS, true, dist, MAX_DIST_TABLE);
    }
};

const FixedTables kFixed;

// the decoder proper; compiled twice below (plain x86-64, and with BMI2's three-operand shifts: every symbol is a shift by a
// table-given count on the loop-carried bit buffer)
__attribute__((always_inline)) inline bool inflate_body(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len) {
    const FixedTables& fixed = kFixed;
    uint32_t lit_dyn[MAX_LIT_TABLE], dist_dyn[MAX_DIST_TABLE];
    Bits b;
    b.in = src;
    b.end = src + src_len;
    uint8_t* out = dst;
    uint8_t* const out_end = dst + dst_len;
    for (;;) {
        b.refill();
        const uint32_t final_block = b.take(1), type = b.take(2);
        const uint32_t* lit;
        const uint32_t* dist;
        if (type == 0) {   // stored: skip to the byte boundary, LEN / NLEN, bytes
            b.drop(b.cnt & 7u);
            b.refill();
            if (b.cnt < 32) return false;
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ 0xffffu) != nlen) return false;
            // give back the whole bytes still sitting in the bit buffer
            const uint8_t* p = b.in - (b.cnt >> 3);
            if ((size_t)(b.end - p) < len || (size_t)(out_end - out) < len) return false;
            memcpy(out, p, len);
            out += len;
            b.in = p + len;
            b.buf = 0;
            b.cnt = 0;
            if (final_block) break;
            continue;
        } else if (type == 1) {
            lit = fixed.lit;
            dist = fixed.dist;
        } else if (type == 2) {
            const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
            if (hlit > 286 || hdist > 30) return false;
            uint8_t cl[19] = {0};
            for (uint32_t i = 0; i < hclen; ++i) {
                if (b.cnt < 3) b.refill();
                cl[kClOrder[i]] = (uint8_t)b.take(3);
            }
            uint32_t cl_table[(1 << 7) + 8];
            if (!build_table(cl, 19, 7, false, cl_table, (1 << 7) + 8)) return false;
            uint8_t lens[286 + 30 + 138];
            uint32_t n = 0;
            while (n < hlit + hdist) {
                b.refill();
                const uint32_t e = cl_table[b.peek(7)];
                const uint32_t kind = (e >> 12) & 15u;
                if (kind == K_INVALID) return false;
                b.drop(e & 0xffu);
                // the code-length alphabet reuses the literal/length constructor: symbols 0..18 arrive as literals
                // 0..18 (K_LITERAL) -- 16, 17, 18 are the repeat codes
                const uint32_t sym = e >> 16;
                if (kind != K_LITERAL || sym > 18) return false;
                if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16) { if (!n) return false; val = lens[n - 1]; rep = 3 + b.take(2); }
                else if (sym == 17) rep = 3 + b.take(3);
                else rep = 11 + b.take(7);
                if (n + rep > hlit + hdist) return false;
                memset(lens + n, (int)val, rep);
                n += rep;
            }
            if (b.overrun || lens[256] == 0) return false;
            if (!build_table(lens, (int)hlit, LIT_BITS, false, lit_dyn, MAX_LIT_TABLE, true)) return false;
            if (!build_table(lens + hlit, (int)hdist, DIST_BITS, true, dist_dyn, MAX_DIST_TABLE)) return false;
            lit = lit_dyn;
            dist = dist_dyn;
        } else {
            return false;
        }
        // ---- symbols of a Huffman block.  After a refill the buffer holds >= 56 bits (or all that is left).  BAM payloads are match
        // dominated (4-bit bases and binned qualities repeat: a match every 5-9 bytes, tests/tools measured 12 M matches in 123 MB of
        // long-read records), so the loop is laid out around the match: its test comes first, length + distance decode share one
        // refill whenever the length code is the first token behind it (<= 20 + 28 bits), and short copies are two unconditional
        // 16-byte moves.
        // ---- fast loop: while 32 bytes of input and 320 bytes of output lie ahead, no bounds test per symbol.  Bit budget of one round
        // (a refill leaves >= 56 bits): up to three primary literal slots (<= 11 bits each), or up to two and then a length symbol
        // (<= 15 + 5 bits); the distance symbol (<= 15 + 13) gets a refill of its own; the next round's entry is looked up before the
        // copy so that the load is under way while the bytes move.
        bool block_done = false;
        if (b.end - b.in >= 32 && out_end - out >= 320) {
            const uint8_t* in = b.in;
            const uint8_t* const in_stop = b.end - 32;
            uint8_t* const out_stop = out_end - 320;
            uint64_t buf = b.buf;
            uint32_t cnt = b.cnt;
#define NP_REFILL()                      \
    do {                                 \
        uint64_t w_;                     \
        memcpy(&w_, in, 8);              \
        buf |= w_ << cnt;                \
        in += (63 - cnt) >> 3;           \
        cnt |= 56;                       \
    } while (0)
#define NP_LITERALS(e_)                                  \
    do {                                                 \
        buf >>= (e_) & 0xffu;                            \
        cnt -= (e_) & 0xffu;                             \
        out[0] = (uint8_t)((e_) >> 16);                  \
        out[1] = (uint8_t)((e_) >> 24);                  \
        out += 1 + (((e_) >> 12) & 1u);                  \
    } while (0)
            constexpr uint32_t LMASK = (1u << LIT_BITS) - 1u, DMASK = (1u << DIST_BITS) - 1u;
            NP_REFILL();
            uint32_t e = lit[buf & LMASK];
            for (;;) {
                if (e & E_LITERAL) {
                    NP_LITERALS(e);
                    e = lit[buf & LMASK];
                    if (e & E_LITERAL) {
                        NP_LITERALS(e);
                        e = lit[buf & LMASK];
                        if (e & E_LITERAL) {
                            NP_LITERALS(e);
                            if (in > in_stop || out > out_stop) break;
                            NP_REFILL();
                            e = lit[buf & LMASK];
                            continue;
                        }
                    }
                }
                uint32_t kind = (e >> 12) & 15u;
                if (kind == K_SUB) {
                    e = lit[(e >> 16) + ((uint32_t)(buf >> LIT_BITS) & ((1u << ((e >> 8) & 15u)) - 1u))];
                    kind = (e >> 12) & 15u;
                    if (kind == K_LITERAL) {      // a literal with a long code (<= 15 bits; >= 34 were left)
                        NP_LITERALS(e);
                        if (in > in_stop || out > out_stop) break;
                        NP_REFILL();
                        e = lit[buf & LMASK];
                        continue;
                    }
                }
                if (kind != K_LENGTH) {
                    if (kind != K_END) return false;
                    buf >>= e & 0xffu;
                    cnt -= e & 0xffu;
                    block_done = true;
                    break;
                }
                uint64_t saved = buf;
                buf >>= e & 0xffu;
                cnt -= e & 0xffu;
                const uint32_t len = with_extra(e, saved);
                NP_REFILL();
                uint32_t d = dist[buf & DMASK];
                if (((d >> 12) & 15u) == K_SUB) d = dist[(d >> 16) + ((uint32_t)(buf >> DIST_BITS) & ((1u << ((d >> 8) & 15u)) - 1u))];
                if (((d >> 12) & 15u) != K_LENGTH) return false;
                saved = buf;
                buf >>= d & 0xffu;
                cnt -= d & 0xffu;
                const uint32_t off = with_extra(d, saved);
                if (off > (size_t)(out - dst)) return false;
                const bool more = in <= in_stop;      // (the refill below stays inside the input either way: 32 bytes of margin, <= 21 used per round)
                NP_REFILL();
                e = lit[buf & LMASK];
                const uint8_t* from = out - off;
                uint8_t* o = out;
                out += len;
                if (off >= 8) {           // 32 bytes in four 8-byte moves whatever the length (each move reads behind what the one before
                    uint64_t w;           // wrote: right for every distance >= 8); the slack bytes are overwritten by what follows.  No
                    memcpy(&w, from, 8);  // test of the length or the distance on the way: both are coin flips to the branch predictor
                    memcpy(o, &w, 8);
                    memcpy(&w, from + 8, 8);
                    memcpy(o + 8, &w, 8);
                    memcpy(&w, from + 16, 8);
                    memcpy(o + 16, &w, 8);
                    memcpy(&w, from + 24, 8);
                    memcpy(o + 24, &w, 8);
                    if (len > 32) {
                        o += 32;
                        from += 32;
                        if (off >= 16) {
                            do {
                                memcpy(o, from, 16);
                                from += 16;
                                o += 16;
                            } while (o < out);
                        } else {
                            do {
                                memcpy(&w, from, 8);
                                memcpy(o, &w, 8);
                                from += 8;
                                o += 8;
                            } while (o < out);
                        }
                    }
                } else if (off == 1) {
                    memset(o, *from, len);
                } else {
                    for (uint32_t i = 0; i < len; ++i) o[i] = from[i];   // overlapping: forward, byte by byte
                }
                if (!more || out > out_stop) break;
            }
#undef NP_REFILL
#undef NP_LITERALS
            b.in = in;
            b.buf = buf;
            b.cnt = cnt;
        }
        // ---- careful loop: the last bytes of the input / output (and blocks too short for the fast loop)
        if (!block_done)
        for (;;) {
            b.refill();
            uint32_t e = lit[b.peek(LIT_BITS)];
            uint32_t kind = (e >> 12) & 15u;
            const bool roomy = out_end - out >= 280;     // room for 2 x 2 literals + the longest match in whole 16-byte moves (258 -> 272).  (272 here let a
                                                         // 257/258-byte match behind literals write up to 4 bytes past the block: found by tests/model/inflate_fuzz.cpp)
            if (roomy && b.cnt >= 56 && (kind == K_LITERAL2 || kind == K_LITERAL)) {   // (cnt: a real refill, not the tail of the input)
                // up to two lookups of literals (one or two per slot, <= 11 bits each: primary entries) on this refill
                b.drop_fast(e & 0xffu);
                out[0] = (uint8_t)(e >> 16);
                out[1] = (uint8_t)(e >> 24);
                out += kind == K_LITERAL2 ? 2 : 1;
                e = lit[b.peek(LIT_BITS)];
                kind = (e >> 12) & 15u;
                if (kind == K_LITERAL2 || kind == K_LITERAL) {
                    b.drop_fast(e & 0xffu);
                    out[0] = (uint8_t)(e >> 16);
                    out[1] = (uint8_t)(e >> 24);
                    out += kind == K_LITERAL2 ? 2 : 1;
                    e = lit[b.peek(LIT_BITS)];
                    kind = (e >> 12) & 15u;
                    if (kind != K_LENGTH) continue;      // (whatever it is gets a fresh refill)
                }
            }
            if (kind != K_LENGTH) {
                if (kind == K_LITERAL2) {   // near the end of the output: literals one slot at a time
                    if (out_end - out < 2) return false;   // two more literals than the declared size has room for
                    b.drop(e & 0xffu);
                    *out++ = (uint8_t)(e >> 16);
                    *out++ = (uint8_t)(e >> 24);
                    continue;
                }
                if (kind == K_SUB) {
                    e = lit[(e >> 16) + ((uint32_t)(b.buf >> LIT_BITS) & ((1u << ((e >> 8) & 15u)) - 1u))];
                    kind = (e >> 12) & 15u;
                }
                if (kind == K_LITERAL) {
                    if (out >= out_end) return false;
                    b.drop(e & 0xffu);
                    *out++ = (uint8_t)(e >> 16);
                    continue;
                }
                if (kind == K_END) { b.drop(e & 0xffu); break; }
                if (kind != K_LENGTH) return false;
            }
            uint64_t saved = b.buf;
            b.drop(e & 0xffu);
            const uint32_t len = with_extra(e, saved);
            if (b.cnt < 32) b.refill();                  // distance code + extra bits: <= 28
            uint32_t d = dist[b.peek(DIST_BITS)];
            if (((d >> 12) & 15u) == K_SUB) d = dist[(d >> 16) + ((uint32_t)(b.buf >> DIST_BITS) & ((1u << ((d >> 8) & 15u)) - 1u))];
            if (((d >> 12) & 15u) != K_LENGTH) return false;
            saved = b.buf;
            b.drop(d & 0xffu);
            const uint32_t off = with_extra(d, saved);
            if (b.overrun || off > (size_t)(out - dst) || len > (size_t)(out_end - out)) return false;
            const uint8_t* from = out - off;
            if (off >= 16 && roomy) {   // whole 16-byte moves; the slack bytes are overwritten by what follows (roomy: 258 + 16 fit)
                uint8_t* o = out;
                memcpy(o, from, 16);
                if (len > 16) {
                    memcpy(o + 16, from + 16, 16);
                    if (len > 32) {
                        const uint8_t* const stop = out + len;
                        o += 32;
                        from += 32;
                        do {
                            memcpy(o, from, 16);
                            from += 16;
                            o += 16;
                        } while (o < stop);
                    }
                }
                out += len;
            } else if (off >= 8 && (size_t)(out_end - out) >= (size_t)len + 8) {
                uint8_t* o = out;
                const uint8_t* const stop = out + len;
                do {
                    uint64_t w;
                    memcpy(&w, from, 8);
                    memcpy(o, &w, 8);
                    from += 8;
                    o += 8;
                } while (o < stop);
                out += len;
            } else if (off == 1) {
                memset(out, *from, len);
                out += len;
            } else {
                for (uint32_t i = 0; i < len; ++i) out[i] = from[i];   // overlapping or near the end: forward, byte by byte
                out += len;
            }
        }
        if (b.overrun) return false;
        if (final_block) break;
    }
    return !b.overrun && out == out_end;
}

bool inflate_plain(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len) { return inflate_body(src, src_len, dst, dst_len); }
#if defined(__x86_64__)      // (the second build and its run-time dispatch exist on x86 only: ADVICE r4)
__attribute__((target("bmi,bmi2"))) bool inflate_bmi2(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len) { return inflate_body(src, src_len, dst, dst_len); }
#endif

}  // namespace

bool inflate_raw(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len) {
#if defined(__x86_64__)
    static const bool bmi2 = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("bmi") && !getenv("NP_INFLATE_PLAIN");   // (the variable: the tests run both builds)
    if (bmi2) return inflate_bmi2(src, src_len, dst, dst_len);
#endif
    return inflate_plain(src, src_len, dst, dst_len);
}

}  // namespace np
