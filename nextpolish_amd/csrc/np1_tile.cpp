// Intra-contig tiling of score_chain in the product (the scheme and its proof by fuzz: tests/model/np1_model.cpp,
// np1m_score_chain_tiled).  A contig of any length -- the reference takes contigs up to 2^31 (source/nextPolish:101-102), one HBM
// batch holds about 250 Mb at 30x -- is polished as independent tiles [a, b) with a halo on each side:
//   * votes are local: a slot's pileup depends on the records that cover it (source/lib/contig.c:247-331), and
//   * the chain restarts behind every slot that left the vote with a single state (exact integer scores: np1_core.h, dp_run),
// so a tile computed from the records touching [a - halo, b + halo] gives the untiled result on its own bases as soon as each halo
// holds a single-state slot among the slots whose votes are complete; if one does not, the tile is recomputed with the halo doubled.
// Nothing travels between tiles: each reads its own region of the BAM through the index (np_stream.cpp: load_stream_region), which is
// also what lets the tiles of one dominant contig be dealt over the GPUs of a node.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np1_priv.h"
#include "np_bam.h"
#include "np_stream.h"

// One contig opened for tiling: the FASTA index entry, the contig's draft and the BAM index are read ONCE and serve every tile of every
// call (round 4 read all three again for every tile a rank asked for: O(tiles x contig length) of host I/O).
struct np1_tiler {
    std::string fasta, bam, name;
    std::string draft;
    np::BaiIndex bai;
};

namespace {

struct TileJob { int64_t a = 0, b = 0; };
struct Loaded { bool ok = false; np1_stream* st = nullptr; int32_t lo = 0, hi = 0, e_lo = 0, e_hi = 0; std::string err; };

Loaded load_tile(const np1_tiler* t, int64_t a, int64_t b, int64_t halo) {
    Loaded r;
    const int64_t L = (int64_t)t->draft.size();
    r.e_lo = (int32_t)(a - halo > 0 ? a - halo : 0);
    r.e_hi = (int32_t)(b + halo < L ? b + halo : L);
    r.st = new np1_stream();
    r.ok = np::load_stream_region(t->bam, t->bai, t->name, t->draft, r.e_lo, r.e_hi, &r.st->s, &r.lo, &r.hi, &r.err);
    if (!r.ok) { delete r.st; r.st = nullptr; }
    return r;
}

}  // namespace

extern "C" {

np1_tiler* np1_tiler_open(const char* fasta, const char* bam, const char* name) {
    if (!fasta || !bam || !name) { np1_set_error("np1_tiler_open: null argument"); return nullptr; }
    np::Fai fai;
    if (!fai.load(fasta)) { np1_set_error(std::string("cannot load FASTA/index: ") + fasta); return nullptr; }
    const int id = fai.find(name);
    if (id < 0) { np1_set_error(std::string("contig not in FASTA index: ") + name); return nullptr; }
    np1_tiler* t = new np1_tiler();
    t->fasta = fasta; t->bam = bam; t->name = name;
    if (!fai.fetch(id, &t->draft)) { np1_set_error(std::string("cannot fetch contig: ") + name); delete t; return nullptr; }
    if (!t->bai.load(std::string(bam) + ".bai")) { np1_set_error(std::string("cannot load BAM index: ") + bam + ".bai"); delete t; return nullptr; }
    return t;
}
void np1_tiler_close(np1_tiler* t) { delete t; }
int64_t np1_tiler_length(const np1_tiler* t) { return t ? (int64_t)t->draft.size() : -1; }

// Tiles first_tile, first_tile + tile_stride, ... of the opened contig on ctx's device; *out = their pieces joined in tile order,
// piece_len[i] (optional, room for every tile of this call) = the length of the i-th of them, so that a caller that deals tiles over ranks
// can cut the string back into its pieces.  While the device runs tile t a helper thread reads the records of tile t + 1 (a tile whose
// halo turns out too narrow is read again, synchronously, with the halo doubled).
int np1_tiler_run(np1_tiler* t, np1_ctx* ctx, const Configure* cfg, int64_t tile_bp, int64_t halo_bp, int64_t first_tile, int64_t tile_stride,
                  char** out, int64_t* out_len, int64_t* piece_len, uint64_t* stats) {
    if (!t || !ctx || !cfg || !out || !out_len || tile_bp <= 0 || halo_bp <= 0 || tile_stride <= 0 || first_tile < 0) {
        np1_set_error("np1_tiler_run: bad argument");
        return -1;
    }
    const int64_t L = (int64_t)t->draft.size();
    std::vector<TileJob> jobs;
    {
        int64_t tile_no = 0;
        for (int64_t a = 0; a < L; a += tile_bp, ++tile_no) {
            if (tile_no < first_tile || (tile_no - first_tile) % tile_stride != 0) continue;     // (this rank's tiles: first_tile, first_tile + stride, ...)
            jobs.push_back(TileJob{a, a + tile_bp < L ? a + tile_bp : L});
        }
    }
    std::string joined;
    uint64_t n_redo = 0, n_rec = 0, max_tile_records = 0;
    // one tile ahead: the loader's result is handed over through a future-like slot
    std::mutex mu;
    std::condition_variable cv;
    Loaded next;
    bool next_ready = false;
    std::thread loader;
    // While the device runs tile t a helper thread reads tile t + 1 (NP1_TILE_PREFETCH=0 switches that off: read, upload, run, release, tile
    // after tile on one thread).  History: round 5 had this and a reused batch object as defaults for a day, then switched both off on suspicion
    // after a one-process run of the GPU suite stopped inside a tiling test.  Round 6 found the stop: hipFree itself, in any long-lived process
    //  -- the read-ahead thread makes no HIP call at all.  Since the allocator cache of np_devalloc.h a fresh batch per
    // tile costs no runtime call in steady state, so the reused batch object is gone and the read-ahead is the default again.
    static const bool prefetch = !(getenv("NP1_TILE_PREFETCH") && strcmp(getenv("NP1_TILE_PREFETCH"), "0") == 0);
    auto start_load = [&](size_t k) {
        next_ready = false;
        if (!prefetch) return;
        try {
            loader = std::thread([&, k] {
                Loaded r = load_tile(t, jobs[k].a, jobs[k].b, halo_bp);
                std::lock_guard<std::mutex> g(mu);
                next = std::move(r);
                next_ready = true;
                cv.notify_all();
            });
        } catch (const std::system_error&) {      // no thread to be had: the tile is read when its turn comes
            next = Loaded();
            next.err = "";
            next_ready = false;
        }
    };
    auto take_load = [&](size_t k) -> Loaded {
        if (loader.joinable()) {
            { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return next_ready; }); }
            loader.join();
            Loaded r = std::move(next);
            next = Loaded();
            return r;
        }
        return load_tile(t, jobs[k].a, jobs[k].b, halo_bp);
    };
    int rc_all = 0;
    np1_batch* bt = nullptr;
    if (!jobs.empty()) start_load(0);
    for (size_t k = 0; k < jobs.size() && rc_all == 0; ++k) {
        const int64_t a = jobs[k].a, b = jobs[k].b;
        Loaded cur = take_load(k);
        if (k + 1 < jobs.size()) start_load(k + 1);
        size_t piece = 0;
        for (int64_t halo = halo_bp;; halo *= 2) {
            if (!cur.ok) { np1_set_error(cur.err.empty() ? "np1_tiler_run: cannot read a tile" : cur.err); rc_all = -1; break; }
            const int32_t e_lo = cur.e_lo, e_hi = cur.e_hi, lo = cur.lo;
            n_rec += cur.st->s.n_reads();
            if (cur.st->s.n_reads() > max_tile_records) max_tile_records = cur.st->s.n_reads();
            int rc = 0;
            bt = np1_batch_upload(ctx, cur.st);      // (complete on return: the stream may go)
            rc = bt ? 0 : -1;
            delete cur.st;
            cur.st = nullptr;
            if (rc == 0) rc = np1_batch_keep_single(bt, 1);
            if (rc == 0) rc = np1_batch_score_chain(bt, cfg, nullptr);
            uint32_t j[4] = {0, 0, 0, 0};
            if (rc == 0) rc = np1_batch_tile_join(bt, (uint32_t)(e_lo - lo), (uint32_t)(a - lo), (uint32_t)(b - lo), (uint32_t)(e_hi - lo), e_lo > 0 ? 2u : 0u, j);
            if (rc != 0) { if (bt) { np1_batch_free(bt); bt = nullptr; } rc_all = -1; break; }
            // a single-state slot inside each halo, among the slots whose votes are complete (e_lo .. e_hi), clear of the two slots behind
            // an artificial start whose draft context is cut short (contig.c:373-383); a halo that reaches the contig's end needs none
            const bool left_ok = a == 0 || e_lo == 0 || j[0] != 0, right_ok = b == L || e_hi == L || j[1] != 0;
            if (left_ok && right_ok) {
                const size_t at = joined.size();
                piece = (size_t)(j[3] - j[2]);
                joined.resize(at + piece);
                rc = np1_batch_result_range(bt, j[2], j[3], &joined[at]);
                np1_batch_free(bt);
                bt = nullptr;
                if (rc != 0) rc_all = -1;
                break;
            }
            np1_batch_free(bt);
            bt = nullptr;
            ++n_redo;
            if (halo > ((int64_t)1 << 30)) { np1_set_error("np1_tiler_run: no single-state slot found in a halo of 2^30 bases"); rc_all = -1; break; }
            cur = load_tile(t, a, b, halo * 2);
        }
        if (cur.st) { delete cur.st; cur.st = nullptr; }
        if (piece_len) piece_len[k] = (int64_t)piece;
    }
    if (loader.joinable()) {      // (a failure above: the tile being read ahead is dropped)
        { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return next_ready; }); }
        loader.join();
        if (next.st) delete next.st;
    }
    if (bt) np1_batch_free(bt);
    if (rc_all != 0) return -1;
    char* buf = (char*)malloc(joined.size() + 1);
    if (!buf) { np1_set_error("np1_tiler_run: out of memory"); return -1; }
    memcpy(buf, joined.data(), joined.size());
    buf[joined.size()] = '\0';
    *out = buf;
    *out_len = (int64_t)joined.size();
    if (stats) { stats[0] = jobs.size(); stats[1] = n_redo; stats[2] = n_rec; stats[3] = max_tile_records; }
    return 0;
}

int np1_score_chain_tiled(np1_ctx* ctx, const char* fasta, const char* bam, const char* name, const Configure* cfg, int64_t tile_bp, int64_t halo_bp,
                          int64_t first_tile, int64_t tile_stride, char** out, int64_t* out_len, uint64_t* stats) {
    if (!ctx || !fasta || !bam || !name || !cfg || !out || !out_len || tile_bp <= 0 || halo_bp <= 0 || tile_stride <= 0 || first_tile < 0) {
        np1_set_error("np1_score_chain_tiled: bad argument");
        return -1;
    }
    np1_tiler* t = np1_tiler_open(fasta, bam, name);
    if (!t) return -1;
    const int rc = np1_tiler_run(t, ctx, cfg, tile_bp, halo_bp, first_tile, tile_stride, out, out_len, nullptr, stats);
    np1_tiler_close(t);
    return rc;
}

void np1_free_string(char* s) { free(s); }

// The CLI's / caller's pass over a whole FASTA index with tiling switched on: contigs of more than tile_bp bases are polished tile by
// tile (above), the runs of shorter contigs between them flow through the pipe in batches as before; every contig reaches `sink` in
// index order.
int np1_run_files_tiled(np1_pipe* pipe, int device, const char* fasta, const char* bam, int64_t batch_bp, int64_t tile_bp, int64_t halo_bp,
                        const Configure* cfg, np1_sink_fn sink, void* user) {
    if (!pipe || !fasta || !bam || !cfg || !sink || tile_bp <= 0) { np1_set_error("np1_run_files_tiled: bad argument"); return -1; }
    np::Fai fai;
    if (!fai.load(fasta)) { np1_set_error(std::string("cannot load FASTA/index: ") + fasta); return -1; }
    np1_ctx* ctx = nullptr;
    std::vector<const char*> run;
    auto flush = [&]() -> int {
        if (run.empty()) return 0;
        const int rc = np1_pipe_run_files(pipe, fasta, bam, run.data(), (int)run.size(), batch_bp, cfg, 1, sink, user);
        run.clear();
        return rc;
    };
    int rc = 0;
    for (int i = 0; i < fai.nseq() && rc == 0; ++i) {
        const np::FaiEntry& e = fai.entry(i);
        if ((int64_t)e.len <= tile_bp) { run.push_back(e.name.c_str()); continue; }
        rc = flush();
        if (rc != 0) break;
        if (!ctx) ctx = np1_ctx_create(device);
        if (!ctx) { rc = -1; break; }
        char* seq = nullptr;
        int64_t len = 0;
        uint64_t st[4] = {0, 0, 0, 0};
        rc = np1_score_chain_tiled(ctx, fasta, bam, e.name.c_str(), cfg, tile_bp, halo_bp > 0 ? halo_bp : 1000, 0, 1, &seq, &len, st);
        if (rc == 0) {
            if (getenv("NP1_TIMING")) fprintf(stderr, "[np1 tiles] %s: %lld bases in %llu tiles (%llu recomputed with a wider halo), %llu records read\n", e.name.c_str(),
                                              (long long)e.len, (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2]);
            sink(user, e.name.c_str(), seq, len);
            np1_free_string(seq);
        }
    }
    if (rc == 0) rc = flush();
    if (ctx) np1_ctx_destroy(ctx);
    return rc;
}

}  // extern "C"
