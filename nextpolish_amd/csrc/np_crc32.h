// CRC-32 (gzip polynomial, reflected) of BGZF blocks: the carry-less-multiply folding of Gopal et al., "Fast CRC Computation for
// Generic Polynomials Using PCLMULQDQ Instruction" (Intel, 2009) -- four 128-bit lanes folded per 64 input bytes, then 128 -> 64 ->
// 32 bits with a Barrett reduction -- behind zlib's crc32() interface; zlib's table version (1 GB/s) when the CPU has no PCLMULQDQ
// and for the bytes that do not fill a 16-byte lane.  The reference's htslib verifies the CRC of every block (bgzf.c); so does this
// reader, and at 1 GB/s that check was a quarter of the host time of inflating a block.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace np {

#if defined(__x86_64__)
// len >= 64 and a multiple of 16; crc = the running value as the hardware-style algorithms keep it (bitwise NOT of zlib's)
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold(const uint8_t* buf, size_t len, uint32_t crc) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i*)k1k2);
    buf += 64;
    len -= 64;
    while (len >= 64) {          // four lanes in parallel
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    x0 = _mm_load_si128((const __m128i*)k3k4);          // the four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {          // single lanes
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);            // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i*)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i*)poly);          // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// same value as zlib's crc32(crc32(0, NULL, 0), buf, len)
inline uint32_t crc32_block(const uint8_t* buf, size_t len) {
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
#if defined(__x86_64__)
    static const bool fast = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (fast && len >= 64) {
        const size_t n = len & ~(size_t)15;
        crc = ~crc32_fold(buf, n, ~crc);
        buf += n;
        len -= n;
    }
#endif
    return len ? (uint32_t)crc32(crc, buf, (uInt)len) : crc;
}

}  // namespace np
