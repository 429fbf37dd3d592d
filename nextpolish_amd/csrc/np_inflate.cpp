// Raw DEFLATE (RFC 1951) decoder for BGZF blocks, written from the RFC for this reader: BGZF inflate is the largest
// single host cost of both polishing paths, and a decoder specialised for "whole block in, whole
// block out" needs no stream state: 64-bit bit buffer refilled eight bytes at a time, one table lookup per symbol
// (11-bit primary table for literals/lengths, 8-bit for distances, second-level tables for longer codes), the decoded
// length known in advance.  Returns false on anything it does not accept (malformed or truncated input, output size
// mismatch); the caller then falls back to zlib, so a false negative costs time, never correctness.
#include "np_inflate.h"

#include <cstring>

namespace np {
namespace {

constexpr int LIT_BITS = 11, DIST_BITS = 8;
constexpr int MAX_LIT_TABLE = (1 << LIT_BITS) + 1024, MAX_DIST_TABLE = (1 << DIST_BITS) + 512;   // primary + every possible second level

// table entry: bits 0..7 code length to consume (first level of a long code: the primary bits), 8..11 extra bits,
// 12..15 kind, 16..31 value (literal byte, base length, base distance, or second-level offset with its index bits in 8..11)
enum : uint32_t { K_LITERAL = 0, K_LENGTH = 1, K_END = 2, K_SUB = 3, K_INVALID = 4, K_LITERAL2 = 5 };   // LITERAL2: two literals in one primary slot
inline uint32_t mk(uint32_t value, uint32_t kind, uint32_t extra, uint32_t nbits) { return value << 16 | kind << 12 | extra << 8 | nbits; }

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
    uint32_t code = 0;
    for (int len = 1; len <= 15; ++len) {
        code = (code + (uint32_t)count[len - 1]) << 1;
        next_code[len] = code;
    }
    const uint32_t primary = 1u << table_bits;
    if (left != 0)      // an incomplete code leaves slots no symbol fills (a complete one writes every slot below)
        for (uint32_t i = 0; i < primary; ++i) table[i] = mk(0, K_INVALID, 0, 1);
    // longest code below every primary slot that heads a second level
    uint8_t sub_max[1 << LIT_BITS];
    memset(sub_max, 0, primary);
    uint32_t codes[288];
    for (int s = 0; s < n_sym; ++s) {
        const int len = lens[s];
        if (!len) continue;
        codes[s] = reverse_bits(next_code[len]++, len);
        if (len > table_bits) {
            const uint32_t slot = codes[s] & (primary - 1);
            if (len > sub_max[slot]) sub_max[slot] = (uint8_t)len;
        }
    }
    uint32_t next_free = primary;
    for (uint32_t slot = 0; slot < primary; ++slot) {
        if (!sub_max[slot]) continue;
        const uint32_t sub_bits = (uint32_t)sub_max[slot] - (uint32_t)table_bits;
        if (next_free + (1u << sub_bits) > (uint32_t)table_cap) return false;
        table[slot] = mk(next_free, K_SUB, sub_bits, (uint32_t)table_bits);
        for (uint32_t i = 0; i < (1u << sub_bits); ++i) table[next_free + i] = mk(0, K_INVALID, 0, 1);
        next_free += 1u << sub_bits;
    }
    for (int s = 0; s < n_sym; ++s) {
        const int len = lens[s];
        if (!len) continue;
        uint32_t e;
        if (is_dist) {
            if (s >= 30) { e = mk(0, K_INVALID, 0, (uint32_t)len); }
            else e = mk(kDistBase[s], K_LENGTH, kDistExtra[s], (uint32_t)len);
        } else if (s < 256) {
            e = mk((uint32_t)s, K_LITERAL, 0, (uint32_t)len);
        } else if (s == 256) {
            e = mk(0, K_END, 0, (uint32_t)len);
        } else if (s <= 285) {
            e = mk(kLenBase[s - 257], K_LENGTH, kLenExtra[s - 257], (uint32_t)len);
        } else {
            e = mk(0, K_INVALID, 0, (uint32_t)len);
        }
        if (len <= table_bits) {
            for (uint32_t i = codes[s]; i < primary; i += 1u << len) table[i] = e;
        } else {
            const uint32_t head = table[codes[s] & (primary - 1)];
            const uint32_t base = head >> 16, sub_bits = (head >> 8) & 15u;
            for (uint32_t i = codes[s] >> table_bits; i < (1u << sub_bits); i += 1u << (len - table_bits)) table[base + i] = e;
        }
    }
    if (pair_literals) {   // a primary slot whose bits hold two whole literal codes decodes both at once
        uint32_t paired[1 << LIT_BITS];
        for (uint32_t i = 0; i < primary; ++i) {
            const uint32_t e1 = table[i];
            paired[i] = e1;
            const uint32_t l1 = e1 & 0xffu;
            if (((e1 >> 12) & 15u) != K_LITERAL || l1 >= (uint32_t)table_bits) continue;
            const uint32_t e2 = table[i >> l1];
            const uint32_t l2 = e2 & 0xffu;
            if (((e2 >> 12) & 15u) != K_LITERAL || l1 + l2 > (uint32_t)table_bits) continue;
            paired[i] = mk((e1 >> 16) | (e2 >> 16) << 8, K_LITERAL2, 0, l1 + l2);
        }
        memcpy(table, paired, sizeof(uint32_t) * primary);
    }
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

}  // namespace

bool inflate_raw(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len) {
    static const FixedTables fixed;
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
        for (;;) {
            b.refill();
            uint32_t e = lit[b.peek(LIT_BITS)];
            uint32_t kind = (e >> 12) & 15u;
            const bool roomy = out_end - out >= 272;     // room for 2 x 2 literals + the longest match + 16 bytes of copy slack
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
            b.drop(e & 0xffu);
            const uint32_t len = (e >> 16) + b.take((e >> 8) & 15u);
            if (b.cnt < 32) b.refill();                  // distance code + extra bits: <= 28
            uint32_t d = dist[b.peek(DIST_BITS)];
            if (((d >> 12) & 15u) == K_SUB) d = dist[(d >> 16) + ((uint32_t)(b.buf >> DIST_BITS) & ((1u << ((d >> 8) & 15u)) - 1u))];
            if (((d >> 12) & 15u) != K_LENGTH) return false;
            b.drop(d & 0xffu);
            const uint32_t off = (d >> 16) + b.take((d >> 8) & 15u);
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

}  // namespace np
