// Device-side raw-DEFLATE (RFC 1951) decoder for BGZF blocks: ONE WAVE PER BLOCK, written for gfx950 from the RFC.
//
// The reference inflates every BGZF block on the host, twice per contig (htslib bgzf.c behind source/lib/contig.c:172-174,
// 692-694).  Here the compressed file bytes go to HBM as they are and the 256 CUs inflate the independent blocks:
//
//  * the Huffman decode is sequential by nature, so all 64 lanes of the wave run it REDUNDANTLY on wave-uniform state (bit
//    buffer, positions): no divergence, table lookups are LDS broadcasts, and uniform values can live in scalar registers;
//  * the compressed bytes are fetched 256 B at a time as one coalesced load (one dword per lane) and handed to the bit
//    buffer with v_readlane; the next 256 B are already in flight while the current ones are consumed;
//  * decoded symbols are collected as up to 64 TOKENS (literal or length/distance), one per lane, in a register; a full
//    group is resolved by the whole wave: an exclusive scan of the token lengths places every token, literal lanes store
//    their byte, match lanes copy their bytes -- a match whose source lies inside the group waits for the lanes before it
//    (rounds over a frontier, one workgroup-scope fence per round; cf. Sitaridi et al., "Massively-Parallel Lossless Data
//    Decompression", ICPP 2016);
//  * code tables are built by the wave: code lengths sit in registers (symbol s -> lane s % 64 of register s / 64),
//    canonical codes come from ballots + popcounts, every lane fills the primary-table slots of its own symbols.  Codes
//    longer than the primary index (10 bits literal/length, 8 bits distance) are decoded arithmetically from the
//    canonical first-code table (they are < 1 % of the symbols), so there are no second-level tables to build.
//
// Anything the decoder does not accept (malformed stream, size mismatch) sets the block's status word; the host inflates
// those blocks itself (np_bgzf.cpp) -- a false negative costs time, never correctness.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace npdev {

constexpr int LIT_BITS = 10, DIST_BITS = 8;
constexpr uint32_t GROUP_BYTES = 2048;   // a token group is closed when 64 tokens are collected or its output could exceed this
constexpr uint32_t K_LITERAL = 0, K_LENGTH = 1, K_END = 2, K_LONG = 3;   // K_LONG: code longer than the primary index (or no code)

struct SymIo { uint32_t rc, op, ntok, gbytes, widx, nb, bb_lo, bb_hi; };   // scalar state handed to / back from decode_symbols

// LDS of one wave
struct InflateLds {
    uint32_t lit[1 << LIT_BITS];     // bits 0..3 code length, 4..7 extra bits, 8..9 kind, 16..31 value (literal, base length)
    uint32_t dist[1 << DIST_BITS];   // same layout, value = base distance
    uint16_t lit_sorted[288];        // symbols ordered by (code length, symbol): canonical decode of long codes
    uint16_t dist_sorted[32];
    uint16_t lit_first[16], lit_count[16], lit_offs[16];    // per code length: first canonical code, number of codes, offset into *_sorted
    uint16_t dist_first[16], dist_count[16], dist_offs[16];
    uint8_t cl_len[320];             // code lengths while a dynamic header is read
    uint32_t cl_tab[128];            // code-length alphabet, 7-bit direct table
    uint32_t gbuf[(GROUP_BYTES + 8) / 4];   // output of the token group being resolved (same dword phase as its place in HBM)
    SymIo io;
};

struct BlockDesc {          // one BGZF block (host-built table)
    uint64_t in_off;        // offset of the raw-deflate payload in the compressed buffer
    uint64_t out_off;       // offset of its inflated bytes in the output buffer
    uint32_t in_len;        // payload bytes
    uint32_t out_len;       // ISIZE
};

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// ---- wave-uniform bit reader over a coalesced register window of the input
struct BitReader {
    const uint32_t* words;   // 4-byte aligned base of the payload (may start up to 3 bytes early)
    uint32_t cur, nxt;       // lane l holds word (chunk * 64 + l) of the current / the next 256-byte chunk
    uint32_t widx;           // next word to feed into the bit buffer (global word index)
    uint64_t bb;
    uint32_t nb;
    uint64_t consumed;       // bits handed out, for the overrun check
    __device__ __forceinline__ void init(const uint8_t* p) {
        const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
        words = reinterpret_cast<const uint32_t*>(p - mis);
        cur = words[lane_id()];
        nxt = words[64 + lane_id()];
        widx = 0;
        bb = 0;
        nb = 0;
        consumed = 0;
        refill();
        bb >>= 8 * mis;
        nb -= 8 * mis;
    }
    __device__ __forceinline__ uint32_t next_word() {
        const uint32_t w = rdlane(cur, widx & 63u);
        ++widx;
        if ((widx & 63u) == 0) {   // crossed into the next chunk: rotate the windows, start the load after next
            cur = nxt;
            nxt = words[(size_t)widx + 64 + lane_id()];
        }
        return w;
    }
    __device__ __forceinline__ void refill() {   // keeps >= 32 valid bits
        if (nb <= 32) {
            bb |= (uint64_t)next_word() << nb;
            nb += 32;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) { bb >>= n; nb -= n; consumed += n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    // byte position (relative to words) of the next unread bit, which must be byte aligned
    __device__ __forceinline__ uint64_t byte_pos() const { return (uint64_t)widx * 4u - (nb >> 3); }
};

// ---- canonical Huffman tables from code lengths held in registers.
// lens[k] on lane l = code length of symbol 64 k + l (0 beyond n_sym).  Fills the primary table and the canonical arrays.
// Returns false (uniformly) for an over-subscribed code.
template <int NREG> struct Lens { uint32_t v[NREG]; };
template <int NREG, int TBITS, bool IS_DIST>
__device__ __forceinline__ bool build_tables(const Lens<NREG> lens_in, uint32_t* table, uint16_t* sorted, uint16_t* first, uint16_t* count, uint16_t* offs) {
    const uint32_t lane = lane_id();
    uint32_t lens[NREG];
#pragma unroll
    for (int k = 0; k < NREG; ++k) lens[k] = lens_in.v[k];
    uint32_t cnt[16];
#pragma unroll
    for (int L = 0; L < 16; ++L) cnt[L] = 0;
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
#pragma unroll
        for (int L = 1; L < 16; ++L) cnt[L] += (uint32_t)__popcll(__ballot(lens[k] == (uint32_t)L));
    }
    // Kraft check + canonical first codes (wave-uniform scalar work)
    int left = 1;
    uint32_t code = 0, off = 0;
    uint32_t firstc[16], offc[16];
    firstc[0] = 0; offc[0] = 0;
    bool over = false;
#pragma unroll
    for (int L = 1; L < 16; ++L) {
        left = (left << 1) - (int)cnt[L];
        if (left < 0) over = true;
        code = (code + cnt[L - 1]) << 1;
        firstc[L] = code;
        offc[L] = off;
        off += cnt[L];
    }
    if (over) return false;
    if (lane < 16) {
        uint32_t f = 0, c = 0, o = 0;
#pragma unroll
        for (int L = 1; L < 16; ++L)
            if (lane == (uint32_t)L) { f = firstc[L]; c = cnt[L]; o = offc[L]; }
        first[lane] = (uint16_t)f;
        count[lane] = (uint16_t)c;
        offs[lane] = (uint16_t)o;
    }
    // every slot starts as "long / no code"
    for (uint32_t i = lane; i < (1u << TBITS); i += 64) table[i] = (K_LONG << 8) | 1u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rank of every symbol among the symbols of its length (in symbol order) -> canonical code
    uint32_t seen[16];
#pragma unroll
    for (int L = 0; L < 16; ++L) seen[L] = 0;
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
        const uint32_t len = lens[k];
        uint32_t rank = 0, fc = 0, oc = 0;
#pragma unroll
        for (int L = 1; L < 16; ++L) {
            const uint64_t m = __ballot(len == (uint32_t)L);
            if (len == (uint32_t)L) {
                rank = seen[L] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                fc = firstc[L];
                oc = offc[L];
            }
            seen[L] += (uint32_t)__popcll(m);
        }
        if (len) {
            const uint32_t sym = 64u * (uint32_t)k + lane;
            sorted[oc + rank] = (uint16_t)sym;
            if (len <= (uint32_t)TBITS) {
                const uint32_t c = fc + rank;                                   // canonical code, MSB first
                const uint32_t rev = __builtin_bitreverse32(c) >> (32u - len);   // as it appears in the LSB-first stream
                uint32_t e;
                if (IS_DIST) {
                    e = sym < 30 ? ((uint32_t)kDistBase[sym] << 16 | K_LENGTH << 8 | (uint32_t)kDistExtra[sym] << 4 | len) : ((K_LONG << 8) | 1u);
                } else if (sym < 256) e = sym << 16 | K_LITERAL << 8 | len;
                else if (sym == 256) e = K_END << 8 | len;
                else if (sym <= 285) e = (uint32_t)kLenBase[sym - 257] << 16 | K_LENGTH << 8 | (uint32_t)kLenExtra[sym - 257] << 4 | len;
                else e = (K_LONG << 8) | 1u;
                for (uint32_t i = rev; i < (1u << TBITS); i += 1u << len) table[i] = e;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

// Canonical decode of a code longer than TBITS from the next 15 stream bits; returns the symbol and its length, or len = 0.
template <int TBITS, class P16>
__device__ __forceinline__ uint32_t decode_long(uint32_t bits15, P16 sorted, P16 first, P16 count, P16 offs, uint32_t* len_out) {
    const uint32_t msb = __builtin_bitreverse32(bits15) >> 17;   // 15 bits, first stream bit on top
    for (uint32_t L = TBITS + 1; L <= 15; ++L) {
        const uint32_t c = msb >> (15u - L);
        const uint32_t f = uni((uint32_t)first[L]), n = uni((uint32_t)count[L]);
        const uint32_t d = c - f;
        if (c >= f && d < n) {
            *len_out = L;
            return uni((uint32_t)sorted[uni((uint32_t)offs[L]) + d]);
        }
    }
    *len_out = 0;
    return 0;
}

// Optional phase clocks (diagnostics: np1_debug_inflate_device with prof != NULL): cycles and counts per block, wave-uniform.
struct Prof { unsigned long long t_tables = 0, t_decode = 0, t_flush = 0, tokens = 0, groups = 0, rounds = 0, matches = 0, match_bytes = 0; };
__device__ __forceinline__ unsigned long long clk() { return __builtin_readcyclecounter(); }

// Token group: lane j of `tok` holds token j.  literal: byte in 0..7; match: bit 31, length in 16..24, distance - 1 in 0..14.
// Resolves ntok tokens (at most GROUP_BYTES of output) at output position op; returns the new position or ~0u on a bad
// distance / overflow.  The group's bytes are assembled in LDS: literals and the parts of matches that come from before the
// group (HBM, written by earlier groups) go in at once; matches that copy from inside the group wait for the lanes before
// them (rounds over a frontier, LDS latency, no fence); then the wave writes the group out with coalesced dword stores.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return (uint64_t)uni((uint32_t)v) | (uint64_t)uni((uint32_t)(v >> 32)) << 32; }

__device__ __noinline__ uint32_t flush_tokens(uint8_t* out_, uint32_t op_, uint32_t out_len_, uint32_t tok, uint32_t ntok_, lds_u32* gbuf32_, Prof* pf = nullptr) {
    // a real call: the arguments arrive in vector registers; everything but `tok` is wave-uniform -> back to scalars
    uint8_t* out = reinterpret_cast<uint8_t*>(uni64((uint64_t)(uintptr_t)out_));
    const uint32_t op = uni(op_), out_len = uni(out_len_), ntok = uni(ntok_);
    lds_u32* gbuf32 = (lds_u32*)(uintptr_t)uni((uint32_t)(uintptr_t)gbuf32_);
    const uint32_t lane = lane_id();
    const unsigned long long t_in = pf ? clk() : 0;
    lds_u8* gbuf = (lds_u8*)gbuf32;
    const bool act = lane < ntok;
    const bool is_match = act && (tok >> 31);
    const uint32_t len = is_match ? ((tok >> 16) & 0x1ffu) : (act ? 1u : 0u);
    const uint32_t dist = (tok & 0x7fffu) + 1u;
    // inclusive scan of the lengths
    uint32_t inc = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += v;
    }
    const uint32_t total = rdlane(inc, 63);
    const uint32_t start = op + inc - len;           // position in the block's output
    if (op + total > out_len || total > GROUP_BYTES) return ~0u;
    const uint64_t mm = __ballot(is_match);
    if (__ballot(is_match && dist > start)) return ~0u;
    const uint32_t gbase = op & 3u;                  // the group sits in gbuf at the dword phase it has in HBM
    const uint32_t rel = gbase + (start - op);
    if (act && !is_match) gbuf[rel] = (uint8_t)tok;
    if (mm) {
        const uint32_t src = start - dist;           // first source byte (output position)
        // bytes that come from before the group: straight from HBM (the stores of earlier groups must have landed)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        uint32_t n_glob = 0;
        if (is_match && src < op) {
            const uint32_t span = len < dist ? len : dist;          // distinct source bytes
            n_glob = op - src < span ? op - src : span;
#pragma clang loop unroll_count(4)
            for (uint32_t i = 0; i < n_glob; ++i) gbuf[rel + i] = out[src + i];
        }
        // the rest copies from inside the group (or repeats the pattern just placed): wait for the lanes before
        uint64_t undone = __ballot(is_match && n_glob < len);
        const uint32_t need = (src + len < start ? src + len : start);   // everything below `need` has to be final
        while (undone) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t k = (uint32_t)__ffsll((long long)undone) - 1u;
            const uint32_t W = rdlane(start, k);                         // all bytes below W are final
            const bool ready = ((undone >> lane) & 1ull) && need <= W;
            if (ready) {
                // byte i comes from source byte i % dist; source bytes below op are already in place (copied above)
                if (dist >= len) {
                    const uint32_t s0 = gbase + (src + n_glob - op);      // src + n_glob >= op here
                    uint32_t i = n_glob;
                    // four bytes in flight per step (source and destination cannot overlap: dist >= len)
#pragma clang loop unroll(disable)
                    for (; i + 4 <= len; i += 4) {
                        const uint32_t o = s0 + (i - n_glob);
                        const uint8_t b0 = gbuf[o], b1 = gbuf[o + 1], b2 = gbuf[o + 2], b3 = gbuf[o + 3];
                        gbuf[rel + i] = b0; gbuf[rel + i + 1] = b1; gbuf[rel + i + 2] = b2; gbuf[rel + i + 3] = b3;
                    }
#pragma clang loop unroll(disable)
                    for (; i < len; ++i) gbuf[rel + i] = gbuf[s0 + (i - n_glob)];
                } else {
                    // self-overlapping: the first `dist` bytes are the pattern (from HBM up to n_glob, else from the group), then it repeats
#pragma clang loop unroll(disable)
                    for (uint32_t i = n_glob; i < dist; ++i) gbuf[rel + i] = gbuf[gbase + (src + i - op)];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma clang loop unroll(disable)
                    for (uint32_t i = dist; i < len; ++i) gbuf[rel + i] = gbuf[rel + i - dist];
                }
            }
            undone &= ~__ballot(ready);
            if (pf) ++pf->rounds;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // write-out: dword w of gbuf <-> dword at out + (op - gbase) + 4 w
    {
        uint8_t* obase = out + (op - gbase);          // 4-byte aligned relative to `out`'s own alignment phase only if out is aligned; handled below
        const uint32_t end = gbase + total;           // bytes [gbase, end) of gbuf are the group
        const bool out_aligned = ((uintptr_t)out & 3u) == 0;
#pragma clang loop unroll(disable)
        for (uint32_t w = lane; 4u * w < end; w += 64) {
            const uint32_t b0 = 4u * w;
            if (out_aligned && b0 >= gbase && b0 + 4u <= end) {
                *reinterpret_cast<uint32_t*>(obase + b0) = gbuf32[w];
            } else {
                for (uint32_t j = 0; j < 4; ++j)
                    if (b0 + j >= gbase && b0 + j < end) obase[b0 + j] = gbuf[b0 + j];
            }
        }
    }
    if (pf) {
        pf->t_flush += clk() - t_in;
        pf->tokens += ntok;
        ++pf->groups;
        pf->matches += (unsigned long long)__popcll(mm);
        pf->match_bytes += total - (ntok - (uint32_t)__popcll(mm));
    }
    return op + total;
}

// Inflates one block; all 64 lanes call it with the same arguments.  Returns 0 on success.

// ---- the symbols of one deflate block, as a function of its own ------------------------------------------------------------
// The block decoder below carries the bit reader, the table-building state and the header fields; inlined into it, the symbol loop
// shared the 100 scalar registers with all of that and spilled its loop counters into vector lanes on every token (the loop is
// bound by scalar issue: one SALU instruction per cycle and CU, whatever the number of waves).  As a real call the loop gets a
// register file of its own: the arguments arrive in vector registers and go back to scalars once, the result goes back through a
// few words of the wave's LDS.
typedef __attribute__((address_space(3))) InflateLds lds_inflate;
typedef __attribute__((address_space(3))) const uint16_t* lds_cu16;
typedef __attribute__((address_space(1))) const uint32_t* glb_cu32;
struct SymState {
    glb_cu32 words;
    lds_inflate* L;
    uint32_t cur, nxt;        // lane l: word l of the current / the next 256-byte chunk
    uint32_t chunk, wl;       // word index = chunk * 64 + wl
    uint64_t bb;
    uint32_t nb;
    uint32_t tok, ntok, gbytes;
    uint32_t lane;
};
// one token.  kFast: the current chunk still holds every word this token can need (at most three refills), so the reader only
// indexes `cur`; otherwise a refill may step into the next chunk (rotate the windows, start the load after next).
// returns 0 = token taken, 1 = end of block, else the error code
template <bool kFast>
__device__ __forceinline__ uint32_t sym_token(SymState& S) {
    auto refill = [&]() {          // keeps >= 32 valid bits
        if (S.nb <= 32) {
            S.bb |= (uint64_t)rdlane(S.cur, S.wl) << S.nb;
            S.nb += 32;
            ++S.wl;
            if (!kFast && S.wl == 64) {
                S.wl = 0;
                ++S.chunk;
                S.cur = S.nxt;
                S.nxt = S.words[(size_t)(S.chunk + 1) * 64 + S.lane];
            }
        }
    };
    refill();
    uint32_t e = uni(S.L->lit[(uint32_t)S.bb & ((1u << LIT_BITS) - 1u)]);
    uint32_t kind = (e >> 8) & 3u;
    if (kind == K_LONG) {
        uint32_t len;
        const uint32_t sym = decode_long<LIT_BITS>((uint32_t)S.bb & 0x7fffu, (lds_cu16)S.L->lit_sorted, (lds_cu16)S.L->lit_first, (lds_cu16)S.L->lit_count, (lds_cu16)S.L->lit_offs, &len);
        if (!len || sym > 285) return 13;
        if (sym < 256) { e = sym << 16 | K_LITERAL << 8 | len; kind = K_LITERAL; }
        else if (sym == 256) { e = K_END << 8 | len; kind = K_END; }
        else { e = (uint32_t)kLenBase[sym - 257] << 16 | K_LENGTH << 8 | (uint32_t)kLenExtra[sym - 257] << 4 | len; kind = K_LENGTH; }
    }
    { const uint32_t n = e & 15u; S.bb >>= n; S.nb -= n; }
    if (kind == K_LITERAL) {
        if (S.lane == S.ntok) S.tok = e >> 16;
        ++S.ntok;
        ++S.gbytes;
        return 0;
    }
    if (kind == K_END) return 1;
    const uint32_t xl = (e >> 4) & 15u;
    const uint32_t mlen = (e >> 16) + ((uint32_t)S.bb & ((1u << xl) - 1u));
    S.bb >>= xl; S.nb -= xl;
    refill();
    uint32_t d = uni(S.L->dist[(uint32_t)S.bb & ((1u << DIST_BITS) - 1u)]);
    if (((d >> 8) & 3u) == K_LONG) {
        uint32_t len;
        const uint32_t sym = decode_long<DIST_BITS>((uint32_t)S.bb & 0x7fffu, (lds_cu16)S.L->dist_sorted, (lds_cu16)S.L->dist_first, (lds_cu16)S.L->dist_count, (lds_cu16)S.L->dist_offs, &len);
        if (!len || sym >= 30) return 14;
        d = (uint32_t)kDistBase[sym] << 16 | K_LENGTH << 8 | (uint32_t)kDistExtra[sym] << 4 | len;
    }
    { const uint32_t n = d & 15u; S.bb >>= n; S.nb -= n; }
    const uint32_t xb = (d >> 4) & 15u;
    refill();
    const uint32_t off = (d >> 16) + ((uint32_t)S.bb & ((1u << xb) - 1u));
    S.bb >>= xb; S.nb -= xb;
    if (S.lane == S.ntok) S.tok = 0x80000000u | mlen << 16 | (off - 1u);
    ++S.ntok;
    S.gbytes += mlen;
    return 0;
}

__device__ __noinline__ uint32_t decode_symbols(const uint32_t* words_, uint32_t cur, uint32_t nxt, uint32_t tok, uint8_t* out_, uint32_t out_len_, uint32_t max_words_,
                                                lds_inflate* L_, uint32_t* cur_out, uint32_t* nxt_out, Prof* pf = nullptr) {
    lds_inflate* L = (lds_inflate*)(uintptr_t)uni((uint32_t)(uintptr_t)L_);
    uint8_t* out = reinterpret_cast<uint8_t*>(uni64((uint64_t)(uintptr_t)out_));
    const uint32_t out_len = uni(out_len_), max_words = uni(max_words_);
    SymState S;
    S.words = (glb_cu32)(uintptr_t)uni64((uint64_t)(uintptr_t)words_);
    S.L = L;
    S.cur = cur; S.nxt = nxt; S.tok = tok;
    S.lane = lane_id();
    const uint32_t widx0 = uni(L->io.widx);
    S.chunk = widx0 >> 6; S.wl = widx0 & 63u;
    S.nb = uni(L->io.nb);
    S.bb = (uint64_t)uni(L->io.bb_lo) | (uint64_t)uni(L->io.bb_hi) << 32;
    S.ntok = uni(L->io.ntok); S.gbytes = uni(L->io.gbytes);
    uint32_t op = uni(L->io.op);
    uint32_t rc = 0;
    for (;;) {
        uint32_t r;
        if (S.wl <= 60) r = sym_token<true>(S);
        else {
            r = sym_token<false>(S);
            if (S.chunk * 64u > max_words) r = r ? r : 16;     // ran off the payload (the input is padded: checked per chunk)
        }
        if (r) { rc = r == 1 ? 0 : r; break; }
        if (S.ntok == 64 || S.gbytes + 258u > GROUP_BYTES) {
            op = uni(flush_tokens(out, op, out_len, S.tok, S.ntok, (lds_u32*)L->gbuf, pf));
            S.ntok = 0;
            S.gbytes = 0;
            if (op == ~0u) { rc = 15; break; }
        }
    }
    if (S.lane == 0) {
        L->io.rc = rc; L->io.op = op; L->io.ntok = S.ntok; L->io.gbytes = S.gbytes; L->io.widx = S.chunk * 64u + S.wl; L->io.nb = S.nb;
        L->io.bb_lo = (uint32_t)S.bb; L->io.bb_hi = (uint32_t)(S.bb >> 32);
    }
    *cur_out = S.cur;
    *nxt_out = S.nxt;
    return S.tok;
}

__device__ __forceinline__ int inflate_block_wave(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, InflateLds& L, Prof* pf = nullptr) {
    const uint32_t lane = lane_id();
    BitReader br;
    br.init(in);
    uint32_t op = 0;
    uint32_t tok = 0, ntok = 0, gbytes = 0;
    const uint32_t mis = (uint32_t)((uintptr_t)in & 3u);
    for (;;) {
        br.refill();
        const unsigned long long t_hdr = pf ? clk() : 0;
        const uint32_t final_block = br.take(1), type = br.take(2);
        if (type == 0) {
            // stored: pending tokens first, then a wave copy
            if (ntok) { op = uni(flush_tokens(out, op, out_len, tok, ntok, (lds_u32*)L.gbuf, pf)); ntok = 0; gbytes = 0; if (op == ~0u) return 1; }
            br.drop(br.nb & 7u);
            br.refill();
            const uint32_t len = br.take(16);
            br.refill();
            const uint32_t nlen = br.take(16);
            if ((len ^ 0xffffu) != nlen) return 2;
            const uint64_t bp = br.byte_pos();                 // relative to br.words
            const uint8_t* src = reinterpret_cast<const uint8_t*>(br.words) + bp;
            if (bp - mis + len > in_len || op + len > out_len) return 3;
            for (uint32_t i = lane; i < len; i += 64) out[op + i] = src[i];
            op += len;
            // restart the bit reader behind the stored bytes
            const uint64_t used_bits = br.consumed + 0;   // consumed counts header bits only; recompute from the byte position
            (void)used_bits;
            const uint8_t* np = src + len;
            const uint64_t consumed_bytes = (uint64_t)(np - in);
            br.init(np);
            br.consumed = consumed_bytes * 8u;
            if (final_block) break;
            continue;
        }
        if (type == 3) return 4;
        Lens<5> llv; Lens<1> dlv; uint32_t (&ll)[5] = llv.v; uint32_t (&dl)[1] = dlv.v;
        uint32_t hlit = 288, hdist = 30;
        if (type == 1) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint32_t s = 64u * k + lane;
                ll[k] = s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : s < 288 ? 8u : 0u;
            }
            dl[0] = lane < 30 ? 5u : 0u;   // fixed distance codes 30, 31 never occur in valid data
        } else {
            br.refill();
            hlit = br.take(5) + 257;
            hdist = br.take(5) + 1;
            const uint32_t hclen = br.take(4) + 4;
            if (hlit > 286 || hdist > 30) return 5;
            // code-length alphabet: 19 lengths of 3 bits, lane s keeps the length of symbol s
            uint32_t cl = 0;
            for (uint32_t i = 0; i < hclen; ++i) {
                br.refill();
                const uint32_t v = br.take(3);
                if (lane == (uint32_t)kClOrder[i]) cl = v;
            }
            {   // 7-bit direct table of the code-length code (Kraft-checked, incomplete codes leave empty slots)
                uint32_t cnt[8], firstc[8];
                int left = 1;
                uint32_t code = 0;
                cnt[0] = 0; firstc[0] = 0;
                bool over = false;
#pragma unroll
                for (int b = 1; b < 8; ++b) cnt[b] = (uint32_t)__popcll(__ballot(cl == (uint32_t)b));
#pragma unroll
                for (int b = 1; b < 8; ++b) {
                    left = (left << 1) - (int)cnt[b];
                    if (left < 0) over = true;
                    code = (code + cnt[b - 1]) << 1;
                    firstc[b] = code;
                }
                if (over) return 6;
                L.cl_tab[lane] = 0xffu;
                L.cl_tab[64 + lane] = 0xffu;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                uint32_t rank = 0, fc = 0;
#pragma unroll
                for (int b = 1; b < 8; ++b) {
                    const uint64_t m = __ballot(cl == (uint32_t)b);
                    if (cl == (uint32_t)b) { rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); fc = firstc[b]; }
                }
                if (cl) {
                    const uint32_t rev = __builtin_bitreverse32(fc + rank) >> (32u - cl);
                    for (uint32_t i = rev; i < 128; i += 1u << cl) L.cl_tab[i] = lane << 4 | cl;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // the hlit + hdist code lengths (run-length coded), decoded uniformly; lane 0 writes them to LDS
            uint32_t n = 0, prev = 0;
            const uint32_t want = hlit + hdist;
            while (n < want) {
                br.refill();
                const uint32_t e = uni(L.cl_tab[br.peek(7)]);
                if (e == 0xffu) return 7;
                br.drop(e & 15u);
                const uint32_t sym = e >> 4;
                uint32_t rep = 1, val = sym;
                if (sym == 16) { if (!n) return 8; val = prev; rep = 3 + br.take(2); }
                else if (sym == 17) { val = 0; rep = 3 + br.take(3); }
                else if (sym == 18) { val = 0; rep = 11 + br.take(7); }
                if (n + rep > want) return 9;
                if (lane < rep) L.cl_len[n + lane] = (uint8_t)val;
                if (rep > 64 && lane + 64 < rep) L.cl_len[n + 64 + lane] = (uint8_t)val;
                if (rep > 128 && lane + 128 < rep) L.cl_len[n + 128 + lane] = (uint8_t)val;
                n += rep;
                prev = val;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint32_t s = 64u * k + lane;
                ll[k] = s < hlit ? (uint32_t)L.cl_len[s] : 0u;
            }
            dl[0] = lane < hdist ? (uint32_t)L.cl_len[hlit + lane] : 0u;
            if (rdlane(ll[4], 0) == 0) return 10;   // no end-of-block code
        }
        if (!build_tables<5, LIT_BITS, false>(llv, L.lit, L.lit_sorted, L.lit_first, L.lit_count, L.lit_offs)) return 11;
        if (!build_tables<1, DIST_BITS, true>(dlv, L.dist, L.dist_sorted, L.dist_first, L.dist_count, L.dist_offs)) return 12;
        const unsigned long long t_sym = pf ? clk() : 0;
        if (pf) pf->t_tables += t_sym - t_hdr;
        const unsigned long long fl0 = pf ? pf->t_flush : 0;
        // ---- symbols (decode_symbols: a call, see there)
        {
            if (lane == 0) {
                L.io.op = op; L.io.ntok = ntok; L.io.gbytes = gbytes; L.io.widx = br.widx; L.io.nb = br.nb;
                L.io.bb_lo = (uint32_t)br.bb; L.io.bb_hi = (uint32_t)(br.bb >> 32);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t max_words = (in_len + mis + 3u) / 4u + 16u + (uint32_t)(reinterpret_cast<const uint32_t*>(in - mis) - br.words);
            uint32_t ncur, nnxt;
            tok = decode_symbols(br.words, br.cur, br.nxt, tok, out, out_len, max_words, (lds_inflate*)&L, &ncur, &nnxt, pf);
            br.cur = ncur; br.nxt = nnxt;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t rc = uni(L.io.rc);
            op = uni(L.io.op); ntok = uni(L.io.ntok); gbytes = uni(L.io.gbytes);
            const uint32_t w1 = uni(L.io.widx), n1 = uni(L.io.nb);
            br.consumed += (uint64_t)(w1 - br.widx) * 32u + br.nb - n1;
            br.widx = w1; br.nb = n1;
            br.bb = (uint64_t)uni(L.io.bb_lo) | (uint64_t)uni(L.io.bb_hi) << 32;
            if (rc) return (int)rc;
            if (br.consumed > (uint64_t)in_len * 8u + 64u) return 16;   // ran off the payload
        }
        if (pf) pf->t_decode += (clk() - t_sym) - (pf->t_flush - fl0);
        if (final_block) break;
    }
    if (ntok) { op = uni(flush_tokens(out, op, out_len, tok, ntok, (lds_u32*)L.gbuf, pf)); if (op == ~0u) return 17; }
    if (br.consumed > (uint64_t)in_len * 8u) return 18;
    return op == out_len ? 0 : 19;
}

}  // namespace npdev
