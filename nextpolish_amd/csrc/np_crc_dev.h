// CRC-32 of inflated BGZF blocks on the device.  The reference's htslib rejects a block whose gzip trailer CRC does not match
// (htslib 1.9 bgzf.c: inflate_block / check_header; the host reader np_bgzf.cpp does the same), so the device-side inflate paths check
// it too: one wave per block, the lanes take the block in 1 KiB pieces aligned to its END (a block has at most 64 KiB), every lane
// runs the byte-wise table CRC over its piece, and the pieces are joined with the identity zlib's crc32_combine rests on,
//     crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B)
// (bit-reflected polynomial arithmetic).  Aligned to the end, lane i's piece is followed by exactly i KiB, so the factors are the 64
// constants x^(8192 i) mod P (crc_shift_table, built on the host).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace npdev {

constexpr uint32_t CRC_POLY = 0xedb88320u;

// a(x) * b(x) mod P, reflected representation (the polynomial 1 is 0x80000000)
__host__ __device__ inline uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}

// out[i] = x^(8 * 1024 * i) mod P, i < 64
inline void crc_shift_table(uint32_t* out) {
    uint32_t x8k = 1u << 30;                         // x^1
    for (int k = 0; k < 13; ++k) x8k = crc_mulmod(x8k, x8k);   // x^(2^13) = x^8192
    uint32_t v = 1u << 31;                           // x^0
    for (int i = 0; i < 64; ++i) { out[i] = v; v = crc_mulmod(v, x8k); }
}

// Four 256-entry tables into LDS by the first 256 threads of a workgroup (slicing by four: tab[k * 256 + b] = the CRC state a byte b
// leaves k bytes further on).  Contains the barriers it needs; every thread of the workgroup has to call it.
// Round 4: the byte-wise loop was one chain of dependent LDS lookups per byte (1 024 of them per lane and block, ~100 cycles each: the
// kernel was bound by that latency, 7 % of the device time of the from-files pipeline); with four tables a word is four INDEPENDENT
// lookups, the chain is one lookup per word.
__device__ __forceinline__ void crc_table_build(uint32_t* tab) {
    uint32_t c = 0;
    if (threadIdx.x < 256) {
        c = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        tab[threadIdx.x] = c;
    }
    __syncthreads();
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        if (threadIdx.x < 256) {
            c = (c >> 8) ^ tab[c & 0xffu];
            tab[k * 256 + threadIdx.x] = c;
        }
    }
    __syncthreads();
}

// CRC-32 (zlib's crc32) of p[0 .. len), len <= 65536, computed by one wave; the result is the same in every lane
__device__ __forceinline__ uint32_t crc_block_wave(const uint8_t* __restrict__ p, uint32_t len, const uint32_t* tab, const uint32_t* __restrict__ shift) {
    const uint32_t lane = __lane_id();
    const uint32_t hi = len > lane * 1024u ? len - lane * 1024u : 0u;      // my piece = [lo, hi)
    const uint32_t lo = hi > 1024u ? hi - 1024u : 0u;
    uint32_t c = 0xffffffffu;
    uint32_t i = lo;
    auto word = [&](uint32_t w) {
        c ^= w;
        c = tab[768 + (c & 0xffu)] ^ tab[512 + ((c >> 8) & 0xffu)] ^ tab[256 + ((c >> 16) & 0xffu)] ^ tab[c >> 24];
    };
    for (; i < hi && ((uintptr_t)(p + i) & 3u); ++i) c = tab[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    for (; i + 4 <= hi && ((uintptr_t)(p + i) & 15u); i += 4) word(*reinterpret_cast<const uint32_t*>(p + i));
    // 16 bytes per load (round 6): the lanes' pieces lie 1 KiB apart, so a load instruction touches 64 cache lines whatever its width --
    // with one-word loads every line came in from L2 sixteen times (the 32 waves of a CU walk 2 MiB, far beyond its L1)
    for (; i + 16 <= hi; i += 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + i);
        word(v.x); word(v.y); word(v.z); word(v.w);
    }
    for (; i + 4 <= hi; i += 4) word(*reinterpret_cast<const uint32_t*>(p + i));
    for (; i < hi; ++i) c = tab[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    c = hi > lo ? ~c : 0u;                            // an empty piece contributes nothing
    uint32_t t = hi > lo ? crc_mulmod(shift[lane], c) : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t ^= (uint32_t)__shfl_xor((int)t, d, 64);
    return t;
}

// Body of the check kernels (np1_ingest.hip, np_bgzf_dev.hip): 4 waves per workgroup, one wave per block.  A block the decoder accepted
// (status 0) whose CRC differs from the gzip trailer (the 4 bytes behind the deflate payload) gets status CRC_MISMATCH: the host then
// inflates it again, checks again and fails the run if the mismatch stands.
constexpr uint32_t CRC_MISMATCH = 0x43524321u;
template <class Desc>
__device__ __forceinline__ void crc_check_body(const uint8_t* __restrict__ comp, const Desc* __restrict__ blocks, uint32_t n_blocks, const uint8_t* __restrict__ out,
                                               uint32_t* __restrict__ status, const uint32_t* __restrict__ shift) {
    __shared__ uint32_t tab[1024];
    crc_table_build(tab);
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const Desc d = blocks[b];
    if (d.out_len == 0 || status[b] != 0) return;
    const uint32_t got = crc_block_wave(out + d.out_off, d.out_len, tab, shift);
    const uint8_t* t = comp + d.in_off + d.in_len;
    const uint32_t want = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
    if (got != want && (threadIdx.x & 63u) == 0) status[b] = CRC_MISMATCH;
}

}  // namespace npdev
