// The product's tiling driver (np1_tile.cpp) over a FAKE device, on real files through the real host readers: a host-only stress of its
// read-ahead thread (and, in round 5, of a batch-reuse branch that round 6 deleted), written after a one-process GPU suite run stopped inside the first
// tiling test.  Build and run (files: any FASTA + sorted, indexed BAM made by nat.Stream.write_files with contigs ctg0001, ctg0002):
//   C=nextpolish_amd/csrc; g++ -O1 -g -std=c++17 -fsanitize=thread -I$C -o /tmp/tile_stress tests/tools/np1_tile_host_stress.cpp $C/np1_tile.cpp \
//       $C/np_stream.cpp $C/np_bam.cpp $C/np_bgzf.cpp $C/np_inflate.cpp -lz -lpthread
//   /tmp/tile_stress 6      (expects /tmp/tsan/g.fa, /tmp/tsan/r.bam)
// Round 5: ThreadSanitizer reports nothing over 6 rounds (2 contigs x tiles of 700 / 5 000 / 20 000 bases, halo 1: hundreds of retries racing the
// read-ahead); 300 rounds of the plain build: no stop.  The host side of the driver is not where that run stopped.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include "../../include/nextpolish1.h"
#include "np1_priv.h"

static thread_local std::string g_err;
void np1_set_error(const std::string& e) { g_err = e; }
extern "C" const char* np1_last_error(void) { return g_err.c_str(); }
struct np1_ctx { int x; };
struct np1_batch { uint64_t G = 0; int64_t n = 0; bool keep = false; bool ran = false; uint64_t sum = 0; };
static std::mt19937 rng(12345);
void np1_stream_unpin(np1_stream*) {}
extern "C" {
np1_batch* np1_batch_create(np1_ctx*) { return new np1_batch(); }
static void take(np1_batch* b, const np1_stream* st) {
    const np::ReadStream& s = st->s;
    b->G = s.draft.size(); b->n = (int64_t)s.n_reads(); b->ran = false;
    uint64_t x = 0;      // touch every array like an upload would
    for (char c : s.draft) x += (unsigned char)c;
    for (size_t i = 0; i < s.pos.size(); ++i) x += (uint64_t)s.pos[i] + s.n_cigar[i] + s.l_qseq[i];
    for (uint32_t c : s.cigar) x += c;
    for (unsigned char c : s.seq) x += c;
    b->sum = x;
}
np1_batch* np1_batch_upload(np1_ctx*, const np1_stream* st) { np1_batch* b = new np1_batch(); take(b, st); return b; }
int np1_batch_reload(np1_batch* b, const np1_stream* st) { take(b, st); return 0; }
void np1_batch_free(np1_batch* b) { delete b; }
int np1_batch_keep_single(np1_batch* b, int on) { b->keep = on != 0; return 0; }
int np1_batch_score_chain(np1_batch* b, const Configure*, float*) { b->ran = true; return 0; }
int np1_batch_tile_join(np1_batch* b, uint32_t i_elo, uint32_t i_a, uint32_t i_b, uint32_t i_ehi, uint32_t, uint32_t out[4]) {
    if (!b->ran || !b->keep || !(i_elo <= i_a && i_a <= i_b && i_b <= i_ehi && i_ehi <= b->G)) { np1_set_error("bad join"); return -1; }
    out[0] = rng() % 3 == 0; out[1] = rng() % 3 == 0; out[2] = i_a; out[3] = i_b;
    return 0;
}
int np1_batch_result_range(np1_batch*, uint32_t o0, uint32_t o1, char* dst) { memset(dst, 'A', o1 - o0); return 0; }
np1_ctx* np1_ctx_create(int) { return new np1_ctx(); }
void np1_ctx_destroy(np1_ctx* c) { delete c; }
int np1_pipe_run_files(np1_pipe*, const char*, const char*, const char* const*, int, int64_t, const Configure*, int, np1_sink_fn, void*) { return 0; }
}
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    Configure cfg{};
    np1_ctx ctx{0};
    for (int r = 0; r < rounds; ++r) {
        for (const char* name : {"ctg0001", "ctg0002"}) {
            for (int64_t tile : {700, 5000, 20000}) {
                char* out = nullptr; int64_t len = 0; uint64_t st[4];
                if (np1_score_chain_tiled(&ctx, "/tmp/tsan/g.fa", "/tmp/tsan/r.bam", name, &cfg, tile, 1, 0, 1, &out, &len, st) != 0) { fprintf(stderr, "failed: %s\n", np1_last_error()); return 1; }
                np1_free_string(out);
            }
        }
        fprintf(stderr, "round %d ok\n", r);
    }
    return 0;
}
