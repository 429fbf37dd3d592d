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
#include <string>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np1_priv.h"
#include "np_bam.h"
#include "np_stream.h"

extern "C" {

int np1_score_chain_tiled(np1_ctx* ctx, const char* fasta, const char* bam, const char* name, const Configure* cfg, int64_t tile_bp, int64_t halo_bp,
                          int64_t first_tile, int64_t tile_stride, char** out, int64_t* out_len, uint64_t* stats) {
    if (!ctx || !fasta || !bam || !name || !cfg || !out || !out_len || tile_bp <= 0 || halo_bp <= 0 || tile_stride <= 0 || first_tile < 0) {
        np1_set_error("np1_score_chain_tiled: bad argument");
        return -1;
    }
    np::Fai fai;
    if (!fai.load(fasta)) { np1_set_error(std::string("cannot load FASTA/index: ") + fasta); return -1; }
    const int id = fai.find(name);
    if (id < 0) { np1_set_error(std::string("contig not in FASTA index: ") + name); return -1; }
    std::string draft;
    if (!fai.fetch(id, &draft)) { np1_set_error(std::string("cannot fetch contig: ") + name); return -1; }
    np::BaiIndex bai;
    if (!bai.load(std::string(bam) + ".bai")) { np1_set_error(std::string("cannot load BAM index: ") + bam + ".bai"); return -1; }
    const int64_t L = (int64_t)draft.size();
    std::string joined;
    uint64_t n_tiles = 0, n_redo = 0, n_rec = 0, max_tile_records = 0;
    int64_t tile_no = 0;
    for (int64_t a = 0; a < L; a += tile_bp, ++tile_no) {
        if (tile_no < first_tile || (tile_no - first_tile) % tile_stride != 0) continue;     // (this rank's tiles: first_tile, first_tile + stride, ...)
        const int64_t b = a + tile_bp < L ? a + tile_bp : L;
        ++n_tiles;
        for (int64_t halo = halo_bp;; halo *= 2) {
            const int32_t e_lo = (int32_t)(a - halo > 0 ? a - halo : 0), e_hi = (int32_t)(b + halo < L ? b + halo : L);
            np1_stream st;
            std::string err;
            int32_t lo = 0, hi = 0;
            if (!np::load_stream_region(bam, bai, name, draft, e_lo, e_hi, &st.s, &lo, &hi, &err)) { np1_set_error(err); return -1; }
            n_rec += st.s.n_reads();
            if (st.s.n_reads() > max_tile_records) max_tile_records = st.s.n_reads();
            np1_batch* bt = np1_batch_upload(ctx, &st);
            if (!bt) return -1;
            int rc = np1_batch_keep_single(bt, 1);
            if (rc == 0) rc = np1_batch_score_chain(bt, cfg, nullptr);
            uint32_t j[4] = {0, 0, 0, 0};
            if (rc == 0) rc = np1_batch_tile_join(bt, (uint32_t)(e_lo - lo), (uint32_t)(a - lo), (uint32_t)(b - lo), (uint32_t)(e_hi - lo), e_lo > 0 ? 2u : 0u, j);
            if (rc != 0) { np1_batch_free(bt); return -1; }
            // a single-state slot inside each halo, among the slots whose votes are complete (e_lo .. e_hi), clear of the two slots behind
            // an artificial start whose draft context is cut short (contig.c:373-383); a halo that reaches the contig's end needs none
            const bool left_ok = a == 0 || e_lo == 0 || j[0] != 0, right_ok = b == L || e_hi == L || j[1] != 0;
            if (left_ok && right_ok) {
                const size_t at = joined.size();
                joined.resize(at + (size_t)(j[3] - j[2]));
                rc = np1_batch_result_range(bt, j[2], j[3], &joined[at]);
                np1_batch_free(bt);
                if (rc != 0) return -1;
                break;
            }
            np1_batch_free(bt);
            ++n_redo;
            if (halo > ((int64_t)1 << 30)) { np1_set_error("np1_score_chain_tiled: no single-state slot found in a halo of 2^30 bases"); return -1; }
        }
    }
    char* buf = (char*)malloc(joined.size() + 1);
    if (!buf) { np1_set_error("np1_score_chain_tiled: out of memory"); return -1; }
    memcpy(buf, joined.data(), joined.size());
    buf[joined.size()] = '\0';
    *out = buf;
    *out_len = (int64_t)joined.size();
    if (stats) { stats[0] = n_tiles; stats[1] = n_redo; stats[2] = n_rec; stats[3] = max_tile_records; }
    return 0;
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
