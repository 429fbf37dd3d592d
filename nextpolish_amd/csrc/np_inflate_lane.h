// Raw DEFLATE (RFC 1951) decoder for whole BGZF blocks in the form ONE LANE PER BLOCK.
//
// np_inflate_dev.h decodes one block per WAVE: the Huffman decode is sequential, so its 64 lanes run it redundantly on scalar state, and
// the kernel ends up bound by the scalar unit (one scalar instruction per cycle and CU, ~70 per token: 3.7 tokens per cycle on the whole
// chip, however many waves are resident).  A BAM file has no shortage of independent blocks (one per 64 KiB: 14 000 in a 16 Mb batch
// at 30x, 400 000 in a 3 Gb run), so here every LANE decodes a block of its own, start to end, as plain sequential code: a token costs a
// lane a dependent chain of a table lookup or two (L2 latency), but tens of thousands of lanes are in flight and the chip hides the
// latency with them -- throughput grows with the number of blocks in the batch instead of being capped by scalar issue.  The code is
// ordinary C++ (no cross-lane operation anywhere), compiled for the host too: the CPU tests run it against zlib over every block type,
// level and strategy and over damaged streams (tests/test_inflate.py), the device kernel calls the same function (np1_ingest.hip).
//
// Tables: canonical Huffman, one lookup per symbol -- 10-bit primary table for literals / lengths, 8-bit for distances, second-level
// tables for longer codes -- in a per-lane slice of HBM scratch (LANE_TABLE_WORDS words; the hot entries of a block live in L2).
// Returns 0 when exactly dst_len bytes came out of the stream, else an error code (the caller inflates refused blocks on the host:
// a refusal costs time, never correctness).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define NPL_HD __host__ __device__ __forceinline__
#define NPL_HD_CALL __host__ __device__ __noinline__     // a real call: one copy, its own register budget
#else
#define NPL_HD inline
#define NPL_HD_CALL inline
#endif

namespace nplane {

constexpr int LIT_BITS = 10, DIST_BITS = 8;       // primary index bits.  (8 / 6 with a 16-bit copy of the primaries in LDS was measured: slower, 70 vs 98 GB/s --
                                                  // three dependent loads for every code longer than the index cost more than the LDS lookups save)
constexpr uint32_t LIT_WORDS = (1u << LIT_BITS) + 1024u, DIST_WORDS = (1u << DIST_BITS) + 512u;   // primary + every possible second level
constexpr uint32_t SUBBASE_WORDS = ((1u << LIT_BITS) + (1u << DIST_BITS)) / 2;   // 16-bit offsets of the second-level tables, one per possible head
constexpr uint32_t SCRATCH_BYTES_EXTRA = 512 + (1u << LIT_BITS);     // code lengths + per-slot maxima while a table is built (bytes, behind the tables)
constexpr uint32_t LANE_TABLE_WORDS = LIT_WORDS + DIST_WORDS + SUBBASE_WORDS + SCRATCH_BYTES_EXTRA / 4;

// entry: bits 0..7 code length to consume (head of a second level: the primary bits), 8..11 second-level index bits, 12..15 kind,
// 16..31 value = the SYMBOL (literal byte; length symbol - 257; distance symbol; for a head: the number of its second-level table,
// whose offset is in the lane's subbase array) -- bases and extra bits of lengths / distances are computed, not stored
enum : uint32_t { K_LITERAL = 0, K_LENGTH = 1, K_END = 2, K_SUB = 3, K_INVALID = 4 };
NPL_HD uint32_t mk(uint32_t value, uint32_t kind, uint32_t extra, uint32_t nbits) { return value << 16 | kind << 12 | extra << 8 | nbits; }

NPL_HD uint32_t len_base(uint32_t i) {      // RFC 1951 3.2.5, computed (no constant tables: the same code runs on host and device)
    if (i < 8) return 3 + i;
    if (i == 28) return 258;
    const uint32_t xb = (i - 4) >> 2;
    return 3 + ((4 + ((i - 4) & 3)) << xb);
}
NPL_HD uint32_t len_extra(uint32_t i) { return i < 8 || i == 28 ? 0 : (i - 4) >> 2; }
NPL_HD uint32_t dist_base(uint32_t i) {
    if (i < 4) return 1 + i;
    const uint32_t xb = (i - 2) >> 1;
    return 1 + ((2 + (i & 1)) << xb);
}
NPL_HD uint32_t dist_extra(uint32_t i) { return i < 4 ? 0 : (i - 2) >> 1; }

NPL_HD uint32_t rev_bits(uint32_t code, uint32_t len) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < len; ++i) { r = r << 1 | (code & 1u); code >>= 1; }
    return r;
}

// kind of alphabet a table decodes
enum : int { A_LITLEN = 0, A_DIST = 1, A_CODELEN = 2 };

// Canonical decode table from code lengths lens[0..n_sym).  sub_max: scratch of 1 << table_bits bytes.  false: over-subscribed code,
// or second levels that do not fit.
NPL_HD_CALL bool build_table(const uint8_t* lens, uint32_t n_sym, uint32_t table_bits, int alphabet, uint32_t* table, uint32_t table_cap, uint8_t* sub_max,
                             uint16_t* subbase) {
    uint32_t count[16];
    for (int i = 0; i < 16; ++i) count[i] = 0;
    for (uint32_t i = 0; i < n_sym; ++i) ++count[lens[i] & 15u];
    count[0] = 0;
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
        left = (left << 1) - (int)count[len];
        if (left < 0) return false;
    }
    uint32_t next_code[16];
    uint32_t code = 0;
    next_code[0] = 0;
    for (int len = 1; len <= 15; ++len) {
        code = (code + count[len - 1]) << 1;
        next_code[len] = code;
    }
    const uint32_t primary = 1u << table_bits;
    for (uint32_t i = 0; i < primary; ++i) { table[i] = mk(0, K_INVALID, 0, 1); sub_max[i] = 0; }
    // pass 1: the longest code below every primary slot that heads a second level (codes are assigned in symbol order, twice)
    {
        uint32_t nc[16];
        for (int i = 0; i < 16; ++i) nc[i] = next_code[i];
        for (uint32_t s = 0; s < n_sym; ++s) {
            const uint32_t len = lens[s];
            if (!len) continue;
            const uint32_t c = rev_bits(nc[len]++, len);
            if (len > table_bits) {
                const uint32_t slot = c & (primary - 1);
                if (len > sub_max[slot]) sub_max[slot] = (uint8_t)len;
            }
        }
    }
    uint32_t next_free = primary, n_sub = 0;
    for (uint32_t slot = 0; slot < primary; ++slot) {
        if (!sub_max[slot]) continue;
        const uint32_t sub_bits = (uint32_t)sub_max[slot] - table_bits;
        if (next_free + (1u << sub_bits) > table_cap) return false;
        subbase[n_sub] = (uint16_t)next_free;
        table[slot] = mk(n_sub++, K_SUB, sub_bits, table_bits);
        for (uint32_t i = 0; i < (1u << sub_bits); ++i) table[next_free + i] = mk(0, K_INVALID, 0, 1);
        next_free += 1u << sub_bits;
    }
    // pass 2: the entries
    for (uint32_t s = 0; s < n_sym; ++s) {
        const uint32_t len = lens[s];
        if (!len) continue;
        const uint32_t c = rev_bits(next_code[len]++, len);
        uint32_t e;
        if (alphabet == A_DIST) e = s >= 30 ? mk(0, K_INVALID, 0, len) : mk(s, K_LENGTH, 0, len);
        else if (alphabet == A_CODELEN || s < 256) e = mk(s, K_LITERAL, 0, len);
        else if (s == 256) e = mk(0, K_END, 0, len);
        else if (s <= 285) e = mk(s - 257, K_LENGTH, 0, len);
        else e = mk(0, K_INVALID, 0, len);
        if (len <= table_bits) {
            for (uint32_t i = c; i < primary; i += 1u << len) table[i] = e;
        } else {
            const uint32_t head = table[c & (primary - 1)];
            const uint32_t base = subbase[head >> 16], sub_bits = (head >> 8) & 15u;
            for (uint32_t i = c >> table_bits; i < (1u << sub_bits); i += 1u << (len - table_bits)) table[base + i] = e;
        }
    }
    return true;
}

typedef uint64_t __attribute__((aligned(1))) u64u;

struct Bits {
    const uint8_t* in;
    const uint8_t* end;
    uint64_t buf;
    uint32_t cnt;
    bool overrun;
    NPL_HD void refill() {
        if (end - in >= 8) {
            buf |= *reinterpret_cast<const u64u*>(in) << cnt;
            in += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && in < end) { buf |= (uint64_t)*in++ << cnt; cnt += 8; }
        }
    }
    NPL_HD uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    NPL_HD void drop(uint32_t n) {
        if (n > cnt) { overrun = true; n = cnt; }
        buf >>= n;
        cnt -= n;
    }
    NPL_HD uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

// src[0 .. src_len): raw DEFLATE stream (readable up to 8 bytes behind its end is NOT required); dst[0 .. dst_len): its output;
// tab: LANE_TABLE_WORDS words of scratch owned by the caller.
NPL_HD int inflate_block(const uint8_t* src, uint32_t src_len, uint8_t* dst, uint32_t dst_len, uint32_t* tab) {
    uint32_t* const lit = tab;
    uint32_t* const dist = tab + LIT_WORDS;
    uint16_t* const lit_sub = reinterpret_cast<uint16_t*>(tab + LIT_WORDS + DIST_WORDS);
    uint16_t* const dist_sub = lit_sub + (1u << LIT_BITS);
    uint8_t* const bytes = reinterpret_cast<uint8_t*>(tab + LIT_WORDS + DIST_WORDS + SUBBASE_WORDS);    // SCRATCH_BYTES_EXTRA: lens[0..454), then sub_max[1 << LIT_BITS]
    uint8_t* const lens = bytes;
    uint8_t* const sub_max = bytes + 512;
    Bits b;
    b.in = src; b.end = src + src_len; b.buf = 0; b.cnt = 0; b.overrun = false;
    uint8_t* out = dst;
    uint8_t* const out_end = dst + dst_len;
    const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (;;) {
        b.refill();
        const uint32_t final_block = b.take(1), type = b.take(2);
        if (type == 0) {   // stored: skip to the byte boundary, LEN / NLEN, bytes
            b.drop(b.cnt & 7u);
            b.refill();
            if (b.cnt < 32) return 2;
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ 0xffffu) != nlen) return 2;
            const uint8_t* p = b.in - (b.cnt >> 3);      // give back the whole bytes still sitting in the bit buffer
            if ((size_t)(b.end - p) < len || (size_t)(out_end - out) < len) return 3;
            for (uint32_t i = 0; i < len; ++i) out[i] = p[i];
            out += len;
            b.in = p + len;
            b.buf = 0;
            b.cnt = 0;
            if (final_block) break;
            continue;
        } else if (type == 1) {
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            if (!build_table(lens, 288, LIT_BITS, A_LITLEN, lit, LIT_WORDS, sub_max, lit_sub)) return 11;
            for (int i = 0; i < 32; ++i) lens[i] = 5;
            if (!build_table(lens, 32, DIST_BITS, A_DIST, dist, DIST_WORDS, sub_max, dist_sub)) return 12;
        } else if (type == 2) {
            const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
            if (hlit > 286 || hdist > 30) return 5;
            uint8_t cl[19];
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            for (uint32_t i = 0; i < hclen; ++i) {
                if (b.cnt < 3) b.refill();
                cl[kClOrder[i]] = (uint8_t)b.take(3);
            }
            // the code-length alphabet borrows the head of the literal table (built afterwards)
            uint32_t* const cl_table = lit;
            if (!build_table(cl, 19, 7, A_CODELEN, cl_table, 128 + 8, sub_max, lit_sub)) return 6;
            uint32_t n = 0;
            while (n < hlit + hdist) {
                b.refill();
                const uint32_t e = cl_table[b.peek(7)];
                if (((e >> 12) & 15u) != K_LITERAL) return 7;
                b.drop(e & 0xffu);
                const uint32_t sym = e >> 16;
                if (sym > 18) return 7;
                if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16) { if (!n) return 8; val = lens[n - 1]; rep = 3 + b.take(2); }
                else if (sym == 17) rep = 3 + b.take(3);
                else rep = 11 + b.take(7);
                if (n + rep > hlit + hdist) return 9;
                for (uint32_t i = 0; i < rep; ++i) lens[n + i] = (uint8_t)val;
                n += rep;
            }
            if (b.overrun || lens[256] == 0) return 10;
            if (!build_table(lens + hlit, hdist, DIST_BITS, A_DIST, dist, DIST_WORDS, sub_max, dist_sub)) return 12;
            if (!build_table(lens, hlit, LIT_BITS, A_LITLEN, lit, LIT_WORDS, sub_max, lit_sub)) return 11;
        } else {
            return 4;
        }
        // ---- the symbols of the block.  After a refill the buffer holds >= 56 bits (or all that is left): a literal / length code
        // with its extra bits needs <= 20, the distance (<= 28) gets its own refill.
        for (;;) {
            b.refill();
            uint32_t e = lit[b.peek(LIT_BITS)];
            if (((e >> 12) & 15u) == K_SUB)         // a code longer than the primary index: second level
                e = lit[(uint32_t)lit_sub[e >> 16] + ((uint32_t)(b.buf >> LIT_BITS) & ((1u << ((e >> 8) & 15u)) - 1u))];
            const uint32_t kind = (e >> 12) & 15u;
            if (kind == K_LITERAL) {
                if (out >= out_end) return 13;
                b.drop(e & 0xffu);
                *out++ = (uint8_t)(e >> 16);
                continue;
            }
            if (kind == K_END) { b.drop(e & 0xffu); break; }
            if (kind != K_LENGTH) return 14;
            b.drop(e & 0xffu);
            const uint32_t len = len_base(e >> 16) + b.take(len_extra(e >> 16));
            b.refill();
            uint32_t d = dist[b.peek(DIST_BITS)];
            if (((d >> 12) & 15u) == K_SUB)
                d = dist[(uint32_t)dist_sub[d >> 16] + ((uint32_t)(b.buf >> DIST_BITS) & ((1u << ((d >> 8) & 15u)) - 1u))];
            if (((d >> 12) & 15u) != K_LENGTH) return 15;
            b.drop(d & 0xffu);
            const uint32_t off = dist_base(d >> 16) + b.take(dist_extra(d >> 16));
            if (b.overrun || off > (size_t)(out - dst) || len > (size_t)(out_end - out)) return 16;
            const uint8_t* from = out - off;
            // The copy.  On the device every load that may alias an earlier store of the lane waits for the memory round trip, so the
            // loops below keep the number of load -> store turns small: sources far enough away are taken four words at a time (one
            // turn per 32 bytes), a short period (offset < 8: runs) is loaded ONCE and written out as pure stores.
            const size_t room = (size_t)(out_end - out);
            if (off >= 32 && room >= (size_t)len + 32) {
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
            } else if (off < 8 && (size_t)(out - dst) >= 8) {   // a period of 1 .. 7 bytes: the last eight bytes hold at least one whole period
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
        if (b.overrun) return 17;
        if (final_block) break;
    }
    return (!b.overrun && out == out_end) ? 0 : 19;
}

}  // namespace nplane
