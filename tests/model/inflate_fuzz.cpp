// Memory-safety fuzz of the two block decoders that run on the host (test infrastructure only): the host decoder of the BGZF reader
// (csrc/np_inflate.cpp) and the host builds of the lane-per-block decoders of the device ingest (csrc/np_inflate_lane.h, csrc/np_inflate_lds.h).  Built with
// -fsanitize=address,undefined (tests/model/Makefile: inflate_fuzz) and run by tests/test_inflate.py.
//   * streams of seven kinds of data (incompressible, 2-bit literals, short near matches, long far matches, binned qualities, runs, a
//     period of 300 bytes with rare substitutions = maximum-length matches with literals between them up to the last byte) x zlib
//     levels 0-9 x five strategies are decoded from and into heap buffers of EXACTLY the stream's and the data's size, so a read or a
//     write one byte outside either is an AddressSanitizer report (blocks of a BGZF window lie back to back and are decoded by different
//     threads: a write past a block's end lands in its neighbour);
//   * the same streams damaged (bit flips, truncation) must be refused or decoded, never touch a byte outside.
// usage: inflate_fuzz <host|lane|lds|lds96|lds85|lds75|lds64> <iterations> <seed>      exit code 0 = every intact stream decoded to its data
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <zlib.h>

#include "np_inflate.h"
#include "np_inflate_lane.h"
#include "np_inflate_lds.h"

static bool decode_lane(const uint8_t* s, size_t sl, uint8_t* d, size_t dl) {
    static std::vector<uint32_t> tab(nplane::LANE_TABLE_WORDS);
    return nplane::inflate_block(s, (uint32_t)sl, d, (uint32_t)dl, tab.data()) == 0;
}

template <int LB, int DB> static bool decode_lds(const uint8_t* s, size_t sl, uint8_t* d, size_t dl) {
    static std::vector<uint16_t> slots(nplds::Layout<LB, DB>::SLOTS);
    nplds::ArrayTab tab{slots.data()};
    static nplds::Scratch sc;
    return nplds::inflate_block<LB, DB>(s, (uint32_t)sl, d, (uint32_t)dl, tab, &sc) == 0;
}

static std::vector<uint8_t> raw_deflate(const std::vector<uint8_t>& d, int level, int strategy) {
    z_stream z;
    memset(&z, 0, sizeof z);
    deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> o(2 * d.size() + 1024);   // (deflateBound is too small for Z_FIXED on incompressible bytes)
    z.next_in = (Bytef*)d.data();
    z.avail_in = (uInt)d.size();
    z.next_out = o.data();
    z.avail_out = (uInt)o.size();
    const int rc = deflate(&z, Z_FINISH);
    o.resize(rc == Z_STREAM_END ? z.total_out : 0);
    deflateEnd(&z);
    return o;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: inflate_fuzz <host|lane> <iterations> <seed>\n"); return 2; }
    bool (*decode)(const uint8_t*, size_t, uint8_t*, size_t) = strcmp(argv[1], "lane") == 0 ? decode_lane : strcmp(argv[1], "lds") == 0 ? decode_lds<10, 8>
                                                               : strcmp(argv[1], "lds96") == 0 ? decode_lds<9, 6> : strcmp(argv[1], "lds75") == 0 ? decode_lds<7, 5> : strcmp(argv[1], "lds85") == 0 ? decode_lds<8, 5> : strcmp(argv[1], "lds64") == 0 ? decode_lds<6, 4> : np::inflate_raw;
    const int iters = atoi(argv[2]);
    std::mt19937_64 rng((uint64_t)atoll(argv[3]));
    static const int strategies[5] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
    size_t good = 0, bad = 0, refused = 0;
    for (int it = 0; it < iters; ++it) {
        const size_t n = rng() % 4 == 0 ? rng() % 400 : rng() % 66000;      // (a BGZF block inflates to at most 65 536 bytes)
        const int kind = (int)(rng() % 8);      // (the periodic kind twice as often: it is the one that found a real overrun)
        std::vector<uint8_t> d(n);
        for (size_t i = 0; i < n; ++i) {
            switch (kind) {
                case 0: d[i] = (uint8_t)rng(); break;
                case 1: d[i] = (uint8_t)"ACGT"[rng() & 3]; break;
                case 2: d[i] = i >= 40 && rng() % 10 ? d[i - 1 - rng() % 39] : (uint8_t)(rng() % 50); break;
                case 3: d[i] = i >= 3000 && rng() % 50 ? d[i - 3000] : (uint8_t)rng(); break;      // long matches at distance 3000: the 16-byte copy path up to the last byte
                case 4: d[i] = (uint8_t)(33 + (rng() % 8) * 5); break;
                case 5: d[i] = i && rng() % 30 ? d[i - 1] : (uint8_t)rng(); break;
                default: d[i] = i >= 300 && rng() % 400 ? d[i - 300] : (uint8_t)rng(); break;      // 258-byte matches at distance 300, a literal now and then
            }
        }
        if (kind >= 6 && n > 1000) {      // 1-4 literals, then a 258-byte match that ends 10-13 bytes before the end of the data: the decoder's 16-byte
            const size_t j = 1 + rng() % 4, t = rng() % j;      // moves reach up to the last byte of the block, or (the overrun found) 1-4 bytes past it
            for (size_t i = 0; i < j; ++i) d[n - 272 - t + i] ^= 0x55;
        }
        const std::vector<uint8_t> c = raw_deflate(d, (int)(rng() % 10), strategies[rng() % 5]);
        if (c.empty()) continue;
        uint8_t* in = (uint8_t*)malloc(c.size());
        memcpy(in, c.data(), c.size());
        uint8_t* out = (uint8_t*)malloc(n ? n : 1);
        if (decode(in, c.size(), out, n) && (n == 0 || memcmp(out, d.data(), n) == 0)) ++good;
        else { ++bad; fprintf(stderr, "intact stream not decoded: iteration %d, %zu bytes, kind %d\n", it, n, kind); }
        if (n > 1 && decode(in, c.size(), out, n - 1)) { ++bad; fprintf(stderr, "declared size one short was accepted: iteration %d\n", it); }
        free(out);
        for (int k = 0; k < 6 && c.size() > 4; ++k) {
            size_t m = c.size();
            uint8_t* in2 = (uint8_t*)malloc(m);
            memcpy(in2, c.data(), m);
            if (k < 4) {
                const int flips = 1 + (int)(rng() % 4);
                for (int f = 0; f < flips; ++f) in2[rng() % m] ^= (uint8_t)(1u << (rng() % 8));
            } else {
                m = rng() % m;
            }
            uint8_t* out2 = (uint8_t*)malloc(n ? n : 1);
            if (!decode(in2, m, out2, n)) ++refused;
            free(out2);
            free(in2);
        }
        free(in);
    }
    printf("%s decoder: %zu intact streams decoded, %zu failures, %zu damaged streams refused\n", argv[1], good, bad, refused);
    return bad != 0;
}
