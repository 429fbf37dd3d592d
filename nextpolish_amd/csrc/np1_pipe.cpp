// Streamed polishing over several device lanes (include/nextpolish1.h: np1_pipe_*).
//
// The reference polishes one contig per call, each call opening the BAM and walking it twice (source/lib/contig.c:172-174,
// 692-694; callers source/lib/nextpolish1.py:181-189, source/lib/contig.c:1084-1100).  Here whole batches of contigs flow
//   loader threads (BGZF inflate + record split)  ->  pinned host arrays  ->  H2D  ->  kernels  ->  D2H  ->  sink
// with `lanes` batches in flight on the device (one HIP stream, one reusable HBM batch and one host thread per lane), so the
// copies and the host-side syncs of one batch hide behind the kernels of another.  Host code only: everything that touches
// the device goes through the np1_batch_* entry points.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <mutex>
#include <system_error>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np1_priv.h"
#include "np_threads.h"
#include "np_bam.h"
#include "np1_ingest.h"

struct np1_pipe {
    int device = 0;
    struct Lane { np1_ctx* ctx = nullptr; np1_batch* batch = nullptr; };
    std::vector<Lane> lanes;
    // results of the last np1_pipe_run
    struct Blob { char* p = nullptr; size_t cap = 0, len = 0; };   // page-locked: the D2H copy of a batch lands here directly
    std::vector<Blob> out;                         // per batch: concatenated strings
    std::vector<std::vector<uint32_t>> bounds;     // per batch: nc + 1 offsets
    std::vector<np1_batch*> resident;              // np1_pipe_upload: batch k lives on lane k % lanes
    uint64_t host_inflated_blocks = 0;             // last np1_pipe_run_files: BGZF blocks the device decoder handed back to the host
    // from-files mode: pinned staging buffers and per-lane HBM scratch live as long as the pipe (allocated on first use)
    std::vector<np1ingest::Staging*> staging;
    std::vector<np1ingest::Staging*> staging_lr;   // np1_pipe_run_phase_files: the long-read file's compressed blocks
    std::vector<np1ingest::Scratch*> scratch;
    std::vector<np1_batch*> phase_lr;              // np1_pipe_run_phase_files: the long-read batch of every lane
};

namespace {

// task numbers of the reference's caller (nextpolish1.py:220): 1 score_chain, 2 kmer_count, 4 snp_valid
int run_task(np1_batch* b, const Configure* cfg, int task) {
    return task == 2 ? np1_batch_kmer_count(b, cfg, nullptr) : task == 4 ? np1_batch_snp_valid(b, cfg, nullptr) : np1_batch_score_chain(b, cfg, nullptr);
}

int polish_on_lane(np1_pipe::Lane& ln, np1_stream* st, const Configure* cfg, int task, np1_pipe::Blob* out) {
    if (np1_batch_reload(ln.batch, st) != 0) return -1;
    const int rc = run_task(ln.batch, cfg, task);
    if (rc != 0) return -1;
    const size_t total = np1_batch_results_total(ln.batch);
    if (total + 1 > out->cap) {
        np1_host_free_pinned(out->p);
        out->cap = total + total / 8 + 4096;
        out->p = (char*)np1_host_alloc_pinned(out->cap);
        if (!out->p) { out->cap = 0; np1_set_error("hipHostMalloc failed"); return -1; }
    }
    out->len = total;
    return np1_batch_results_fetch_to(ln.batch, out->p, out->cap);
}

double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
bool timing_on() { static const bool t = getenv("NP1_TIMING") != nullptr; return t; }

unsigned loader_threads() {
    const char* e = getenv("NP1_LOADERS");
    int n = e ? atoi(e) : 3;
    return (unsigned)(n < 1 ? 1 : n > 16 ? 16 : n);
}

}  // namespace

extern "C" {

np1_pipe* np1_pipe_open(int device, int lanes) {
    if (lanes < 1) lanes = 1;
    if (lanes > 8) lanes = 8;
    np1_pipe* p = new np1_pipe();
    p->device = device;
    for (int i = 0; i < lanes; ++i) {
        np1_pipe::Lane ln;
        ln.ctx = np1_ctx_create(device);
        if (ln.ctx) ln.batch = np1_batch_create(ln.ctx);
        if (!ln.ctx || !ln.batch) {
            if (ln.ctx) np1_ctx_destroy(ln.ctx);
            np1_pipe_close(p);
            return nullptr;
        }
        p->lanes.push_back(ln);
    }
    return p;
}

static void drop_resident(np1_pipe* p) {
    for (np1_batch* b : p->resident) np1_batch_free(b);
    p->resident.clear();
}

int np1_pipe_upload(np1_pipe* p, np1_stream* const* streams, int n) {
    if (!p || (n > 0 && !streams)) { np1_set_error("np1_pipe_upload: null argument"); return -1; }
    drop_resident(p);
    for (int k = 0; k < n; ++k) {
        np1_batch* b = np1_batch_upload(p->lanes[(size_t)k % p->lanes.size()].ctx, streams[k]);
        if (!b) { drop_resident(p); return -1; }
        p->resident.push_back(b);
    }
    return 0;
}

np1_batch* np1_pipe_resident_batch(np1_pipe* p, int k) {
    return (p && k >= 0 && (size_t)k < p->resident.size()) ? p->resident[(size_t)k] : nullptr;
}

// One instrumented pass of resident batch k on lane 0's work buffers: per-stage HIP-event times into stage_ms (np1_stage_count()
// entries); the lengths of its results stay readable through np1_batch_result_len.
int np1_pipe_run_resident_timed(np1_pipe* p, int k, const Configure* cfg, float* stage_ms) {
    if (!p || !cfg || k < 0 || (size_t)k >= p->resident.size()) { np1_set_error("np1_pipe_run_resident_timed: bad argument"); return -1; }
    np1_batch* b = p->resident[(size_t)k];
    np1_batch* w = p->lanes[0].batch;
    np1_batch_swap_work(b, w);
    const int rc = np1_batch_score_chain(b, cfg, stage_ms);
    np1_batch_swap_work(b, w);
    return rc;
}

int np1_pipe_run_resident(np1_pipe* p, const Configure* cfg, int task, int passes) {
    if (!p || !cfg) { np1_set_error("np1_pipe_run_resident: null argument"); return -1; }
    std::atomic<bool> failed(false);
    std::string err;
    std::mutex err_mu;
    const size_t nl = p->lanes.size();
    // (pass, batch) pairs are handed out through one counter: a lane whose thread the system refuses is simply not there, and the
    // lanes that did start take its share (round 3 strode the batches by lane index, which left every nl-th batch undone in that case)
    const size_t nb = p->resident.size();
    const size_t n_items = nb * (size_t)(passes > 0 ? passes : 0);
    std::atomic<size_t> next(0);
    std::vector<std::mutex> busy(nb);                            // (a fast lane can reach batch k of the next pass while a slow one still holds it)
    auto work = [&](size_t lane) {
        for (;;) {
            const size_t it = next.fetch_add(1);
            if (it >= n_items || failed) return;
            np1_batch* b = p->resident[it % nb];
            std::lock_guard<std::mutex> hold(busy[it % nb]);
            np1_batch* w = p->lanes[lane].batch;        // the lane's work buffers (slot arrays, descriptors, DP records ...)
            np1_batch_swap_work(b, w);
            const int rc = run_task(b, cfg, task);
            np1_batch_swap_work(b, w);                           // the batch keeps only its inputs (and its host-side result bounds)
            if (rc != 0) {
                std::lock_guard<std::mutex> g(err_mu);
                if (!failed) err = np1_last_error();
                failed = true;
                return;
            }
        }
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < nl; ++i) {
        try { th.emplace_back(work, i); } catch (const std::system_error&) { break; }
    }
    work(0);
    for (std::thread& t : th) t.join();
    if (failed) { np1_set_error(err); return -1; }
    return 0;
}

// BGZF blocks the device-side ingest handed back to the host since the pipe was opened (a block its decoder does not accept, or one whose
// CRC it found wrong: the host inflates and checks it again); 0 on well-formed files
uint64_t np1_pipe_host_inflated_blocks(np1_pipe* p) { return p ? p->host_inflated_blocks : 0; }
// {block decoder ms, CRC ms, compressed bytes in, inflated bytes out, launches} of the device-side ingest, summed over the lanes, since the
// pipe was opened or the last reset (HIP events on the lanes' streams)
void np1_pipe_ingest_stats(np1_pipe* p, double out[5], int reset) {
    for (int i = 0; i < 5; ++i) out[i] = 0;
    if (!p) return;
    for (np1ingest::Scratch* s : p->scratch) { np1ingest::scratch_stats(s, out); if (reset) np1ingest::scratch_stats_reset(s); }
}

void np1_pipe_close(np1_pipe* p) {
    if (!p) return;
    for (np1_batch* b : p->phase_lr) if (b) np1_batch_free(b);
    for (np1_pipe::Blob& o : p->out) np1_host_free_pinned(o.p);
    drop_resident(p);
    for (np1ingest::Scratch* s : p->scratch) np1ingest::scratch_destroy(s);
    for (np1ingest::Staging* s : p->staging) delete s;
    for (np1ingest::Staging* s : p->staging_lr) delete s;
    for (np1_pipe::Lane& ln : p->lanes) {
        if (ln.batch) np1_batch_free(ln.batch);
        if (ln.ctx) np1_ctx_destroy(ln.ctx);
    }
    delete p;
}

int np1_pipe_run(np1_pipe* p, np1_stream* const* streams, int n, const Configure* cfg, int task) {
    if (!p || !cfg || (n > 0 && !streams)) { np1_set_error("np1_pipe_run: null argument"); return -1; }
    if (p->out.size() > (size_t)n)
        for (size_t k = (size_t)n; k < p->out.size(); ++k) np1_host_free_pinned(p->out[k].p);
    p->out.resize((size_t)n);                      // the buffers of earlier runs are reused
    p->bounds.assign((size_t)n, {});
    std::atomic<int> next(0);
    std::atomic<bool> failed(false);
    std::string err;
    std::mutex err_mu;
    auto work = [&](np1_pipe::Lane& ln) {
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= n || failed) break;
            if (polish_on_lane(ln, streams[k], cfg, task, &p->out[(size_t)k]) != 0) {
                std::lock_guard<std::mutex> g(err_mu);
                if (!failed) err = np1_last_error();
                failed = true;
                break;
            }
            np1_stream_view v;
            np1_stream_get_view(streams[k], &v);
            const uint32_t* b = np1_batch_results_bounds(ln.batch);
            p->bounds[(size_t)k].assign(b, b + v.n_contigs + 1);
        }
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < p->lanes.size() && (int)i < n; ++i) {
        try { th.emplace_back(work, std::ref(p->lanes[i])); } catch (const std::system_error&) { break; }
    }
    work(p->lanes[0]);
    for (std::thread& t : th) t.join();
    if (failed) { np1_set_error(err); return -1; }
    return 0;
}

const char* np1_pipe_result(np1_pipe* p, int batch, int64_t contig, int64_t* len) {
    if (!p || batch < 0 || (size_t)batch >= p->out.size()) return nullptr;
    const std::vector<uint32_t>& b = p->bounds[(size_t)batch];
    if (contig < 0 || (size_t)contig + 1 >= b.size()) return nullptr;
    if (len) *len = (int64_t)b[(size_t)contig + 1] - (int64_t)b[(size_t)contig];
    return p->out[(size_t)batch].p + b[(size_t)contig];
}

int np1_pipe_run_files(np1_pipe* p, const char* fasta, const char* bam, const char* const* names, int n_names, int64_t batch_bp,
                       const Configure* cfg, int task, np1_sink_fn sink, void* user) {
    if (!p || !fasta || !bam || !cfg) { np1_set_error("np1_pipe_run_files: null argument"); return -1; }
    np1ingest::BamSource src;
    {
        std::string e;
        if (!src.open(fasta, bam, &e)) { np1_set_error(e); return -1; }
    }
    const np::Fai& fai = src.fai;
    std::vector<std::string> want;
    if (names && n_names > 0) for (int i = 0; i < n_names; ++i) want.push_back(names[i]);
    else for (int i = 0; i < fai.nseq(); ++i) want.push_back(fai.entry(i).name);
    // greedy in-order packing (a contig longer than batch_bp is a batch of its own)
    std::vector<std::vector<std::string>> plan;
    int64_t cur_bp = 0;
    for (const std::string& nm : want) {
        const int id = fai.find(nm);
        if (id < 0) { np1_set_error("contig not in FASTA index: " + nm); return -1; }
        const int64_t L = fai.entry(id).len;
        if (plan.empty() || (cur_bp > 0 && cur_bp + L > batch_bp)) { plan.emplace_back(); cur_bp = 0; }
        plan.back().push_back(nm);
        cur_bp += L;
    }
    const int n = (int)plan.size();
    const bool with_qual = task == 2 || task == 4;
    // kmer_count / snp_valid follow the reference's region iterator (contig.c:982-1043) replayed on the BAM index and the records' virtual
    // offsets (np1_replay.h), whichever way a batch is decoded; NP1_ITER_REPLAY=0: records in file order (no index: nothing to replay)
    const char* rpl = getenv("NP1_ITER_REPLAY");
    const bool replay = with_qual && src.have_bai && !(rpl && rpl[0] == '0');
    // NP1_INGEST=host: inflate and split the records on host threads (np_stream.cpp) instead of on the device (np1_ingest.hip)
    const char* ing = getenv("NP1_INGEST");
    const bool device_ingest = src.have_bai && !(ing && strcmp(ing, "host") == 0);
    // One work item per batch: either a staging buffer (compressed blocks for the device) or a host-decoded stream.
    struct Item { np1ingest::Staging* staging = nullptr; np1_stream* stream = nullptr; };
    const size_t depth = p->lanes.size() + loader_threads();    // loaded-but-unpolished batches allowed in memory
    std::vector<np1ingest::Staging*> free_staging;
    if (device_ingest) {
        while (p->staging.size() < depth) p->staging.push_back(new np1ingest::Staging());
        free_staging = p->staging;
    }
    p->scratch.resize(p->lanes.size(), nullptr);
    std::vector<np1ingest::Scratch*>& scratch = p->scratch;
    std::mutex mu;
    std::condition_variable cv;
    std::map<int, Item> ready;             // loaded batches waiting for a lane
    struct Done { std::vector<std::string> names; std::vector<char> out; std::vector<uint32_t> bounds; };
    std::map<int, Done> done;              // polished batches waiting for their turn at the sink
    int next_load = 0, next_polish = 0, next_emit = 0, in_memory = 0;
    bool failed = false;
    std::string err;
    auto fail = [&](const std::string& e) {
        std::lock_guard<std::mutex> g(mu);
        if (!failed) err = e;
        failed = true;
        cv.notify_all();
    };
    auto load_host = [&](int k) -> np1_stream* {
        std::vector<const char*> nm;
        for (const std::string& s : plan[(size_t)k]) nm.push_back(s.c_str());
        np1_stream* st = np1_stream_load(fasta, bam, nm.data(), (int)nm.size(), with_qual ? 1 : 0);
        if (st) (void)np1_stream_pin(st);   // best effort: an unpinned stream still uploads, just synchronously
        return st;
    };
    auto loader = [&]() {
        for (;;) {
            int k;
            np1ingest::Staging* sg = nullptr;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return failed || next_load >= n || ((size_t)in_memory < depth && (!device_ingest || !free_staging.empty())); });
                if (failed || next_load >= n) return;
                k = next_load++;
                ++in_memory;
                if (device_ingest) { sg = free_staging.back(); free_staging.pop_back(); }
            }
            Item it;
            const double t_l0 = now_ms();
            if (sg) {
                std::string e;
                const int rc = np1ingest::prepare(src, plan[(size_t)k], sg, &e);
                if (rc < 0) { fail(e); return; }
                if (rc == 0) it.staging = sg;
                else { std::lock_guard<std::mutex> g(mu); free_staging.push_back(sg); }
            }
            if (!it.staging) {
                it.stream = load_host(k);
                if (!it.stream) { fail(np1_last_error()); return; }
            }
            if (timing_on()) fprintf(stderr, "[np1 pipe] batch %d staged on the host in %.1f ms (%s)\n", k, now_ms() - t_l0, it.staging ? "compressed blocks" : "host loader");
            std::lock_guard<std::mutex> g(mu);
            ready[k] = it;
            cv.notify_all();
        }
    };
    auto emit_ready = [&](std::unique_lock<std::mutex>& g) {   // called with mu held; one thread at a time
        while (done.count(next_emit)) {
            Done d = std::move(done[next_emit]);
            done.erase(next_emit);
            ++next_emit;
            g.unlock();
            if (sink)
                for (size_t c = 0; c < d.names.size(); ++c)
                    sink(user, d.names[c].c_str(), d.out.data() + d.bounds[c], (int64_t)d.bounds[c + 1] - (int64_t)d.bounds[c]);
            g.lock();
            --in_memory;
            cv.notify_all();
        }
    };
    auto lane_work = [&](size_t li) {
        np1_pipe::Lane& ln = p->lanes[li];
        for (;;) {
            int k;
            Item it;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return failed || next_polish >= n || ready.count(next_polish); });
                if (failed || next_polish >= n) return;
                k = next_polish++;
                it = ready[k];
                ready.erase(k);
            }
            Done d;
            int rc = 0;
            const double t_p0 = now_ms();
            double t_p1 = t_p0, t_p2 = t_p0;
            if (it.staging) {
                if (!scratch[li]) scratch[li] = np1ingest::scratch_create();
                d.names = it.staging->names();
                rc = np1ingest::ingest(ln.batch, it.staging, with_qual, scratch[li], replay ? &src.bai : nullptr);
                {
                    std::lock_guard<std::mutex> g(mu);
                    free_staging.push_back(it.staging);
                    cv.notify_all();
                }
                if (rc == 1) {      // records the device path does not take (CIGARs in CG tags): the host loader decodes this batch
                    it.stream = load_host(k);
                    if (!it.stream) rc = -1;
                }
            }
            if (rc >= 0 && it.stream) {
                np1_stream_view v;
                np1_stream_get_view(it.stream, &v);
                d.names.clear();
                for (int64_t c = 0; c < v.n_contigs; ++c) d.names.push_back(np1_stream_contig_name(it.stream, c));
                rc = np1_batch_reload(ln.batch, it.stream);
                // the host loader keeps the records' virtual offsets, the device-side ingest brings them down: both replay the iterator
                if (rc == 0 && replay && np1_batch_enable_replay(ln.batch, it.stream, bam) != 0) rc = -1;
            }
            t_p1 = now_ms();
            if (rc == 0) rc = run_task(ln.batch, cfg, task);
            t_p2 = now_ms();
            if (rc == 0) rc = np1_batch_results_fetch(ln.batch);
            if (timing_on()) fprintf(stderr, "[np1 pipe] batch %d lane %zu: ingest %.1f ms, kernels %.1f ms, fetch %.1f ms\n", k, li, t_p1 - t_p0, t_p2 - t_p1, now_ms() - t_p2);
            if (it.stream) np1_stream_free(it.stream);     // after the pass: its arrays were the source of asynchronous copies
            if (rc != 0) { fail(np1_last_error()); return; }
            const uint32_t* b = np1_batch_results_bounds(ln.batch);
            const char* s = np1_batch_results_ptr(ln.batch);
            d.bounds.assign(b, b + d.names.size() + 1);
            d.out.assign(s, s + b[d.names.size()]);     // strings are not NUL-terminated inside the blob: the sink gets (pointer, length)
            std::unique_lock<std::mutex> g(mu);
            done[k] = std::move(d);
            cv.notify_all();          // (the emitter thread below hands the batches to the sink, in order)
        }
    };
    // Round 6: the sink runs on a thread of its own.  It used to run on whichever lane finished a batch while nobody else was emitting -- and
    // the CLI's sink writes the FASTA text into a pipe: 3 GB at the pace of the reader held one of three lanes half of the time.
    auto emitter = [&]() {
        std::unique_lock<std::mutex> g(mu);
        for (;;) {
            cv.wait(g, [&] { return failed || next_emit >= n || done.count(next_emit); });
            if (failed || next_emit >= n) return;
            emit_ready(g);
        }
    };
    std::vector<std::thread> th;
    const unsigned nl = std::min<unsigned>(loader_threads(), (unsigned)std::max(1, n));
    if (np::spawn_helpers(th, nl, loader) == 0) { np1_set_error("cannot start a loader thread (out of threads or mappings)"); return -1; }
    for (size_t i = 1; i < p->lanes.size(); ++i) {      // lanes take the staged batches as they come: one that cannot start is not missed
        try { th.emplace_back(lane_work, i); } catch (const std::system_error&) { break; }
    }
    bool have_emitter = true;
    try { th.emplace_back(emitter); } catch (const std::system_error&) { have_emitter = false; }
    lane_work(0);
    if (!have_emitter) {      // (no thread to be had: everything at the end, in order)
        std::unique_lock<std::mutex> g(mu);
        if (!failed) emit_ready(g);
        next_emit = n;
    }
    for (std::thread& t : th) t.join();
    {   // (nothing is left here unless the run failed)
        std::unique_lock<std::mutex> g(mu);
        if (!failed) emit_ready(g);
        for (auto& kv : ready) if (kv.second.stream) np1_stream_free(kv.second.stream);
    }
    uint64_t host_blocks = 0;
    for (np1ingest::Scratch* s : scratch) host_blocks += np1ingest::scratch_host_blocks(s);
    p->host_inflated_blocks = host_blocks;     // cumulative over the life of the pipe
    if (failed) { np1_set_error(err); return -1; }
    return 0;
}

// Task 3 from files (reference: one snp_phase(tigname, cfg) call per contig and worker, nextpolish1.py:95-96,181-189): contigs in
// batches of batch_bp draft bases; per batch the short-read records come in as compressed BGZF blocks and are inflated and split
// on the device (np1_ingest.hip; host loader where the index or the records do not allow it), and so do the long-read records
// (the host loader spent most of a batch's wall time on them: decode, then page-locking the decoded arrays; NP1_INGEST_LR=host
// keeps it); both land in two batches of lane 0 and one np1_batch_snp_phase pass runs over them.  A loader thread stages batch
// k + 1 while the device works on batch k.
int np1_pipe_run_phase_files(np1_pipe* p, const char* fasta, const char* bam_sr, const char* bam_lr, const char* const* names, int n_names, int64_t batch_bp,
                             const Configure* cfg, np1_sink_fn sink, void* user) {
    if (!p || !fasta || !bam_sr || !bam_lr || !cfg) { np1_set_error("np1_pipe_run_phase_files: null argument"); return -1; }
    np1ingest::BamSource src;
    {
        std::string e;
        if (!src.open(fasta, bam_sr, &e)) { np1_set_error(e); return -1; }
    }
    np1ingest::BamSource src_lr;
    {
        std::string e;
        if (!src_lr.open(fasta, bam_lr, &e)) { np1_set_error(e); return -1; }
    }
    const np::Fai& fai = src.fai;
    std::vector<std::string> want;
    if (names && n_names > 0) for (int i = 0; i < n_names; ++i) want.push_back(names[i]);
    else for (int i = 0; i < fai.nseq(); ++i) want.push_back(fai.entry(i).name);
    std::vector<std::vector<std::string>> plan;
    int64_t cur_bp = 0;
    for (const std::string& nm : want) {
        const int id = fai.find(nm);
        if (id < 0) { np1_set_error("contig not in FASTA index: " + nm); return -1; }
        const int64_t L = fai.entry(id).len;
        if (plan.empty() || (cur_bp > 0 && cur_bp + L > batch_bp)) { plan.emplace_back(); cur_bp = 0; }
        plan.back().push_back(nm);
        cur_bp += L;
    }
    const int n = (int)plan.size();
    const bool all_in_one = n == 1 && !(names && n_names > 0);   // every contig: sequential passes over the files, no index seeks
    const char* ing = getenv("NP1_INGEST");
    const bool device_ingest = src.have_bai && !(ing && strcmp(ing, "host") == 0);
    const char* ing_lr = getenv("NP1_INGEST_LR");
    const bool device_ingest_lr = src_lr.have_bai && !(ing && strcmp(ing, "host") == 0) && !(ing_lr && strcmp(ing_lr, "host") == 0);
    // lanes: batch k runs on lane k % L (own stream, own pair of batch objects, own staging buffers and scratch) -- the staging of one
    // batch and the copies of its blocks overlap the kernels of the other lane's batch; results leave in batch order
    const size_t L = std::min<size_t>(std::max<size_t>(1, p->lanes.size()), 4);
    p->phase_lr.resize(p->lanes.size(), nullptr);
    for (size_t li = 0; li < L; ++li) {
        if (!p->phase_lr[li]) p->phase_lr[li] = np1_batch_create(p->lanes[li].ctx);
        if (!p->phase_lr[li]) return -1;
    }
    while (device_ingest && p->staging.size() < L) p->staging.push_back(new np1ingest::Staging());
    while (device_ingest_lr && p->staging_lr.size() < L) p->staging_lr.push_back(new np1ingest::Staging());
    p->scratch.resize(p->lanes.size(), nullptr);
    struct Item { np1ingest::Staging* staging = nullptr; np1ingest::Staging* staging_lr = nullptr; np1_stream* sr = nullptr; np1_stream* lr = nullptr; std::string err; };
    auto load_host = [&](int k, const char* bam) -> np1_stream* {
        std::vector<const char*> nm;
        for (const std::string& s : plan[(size_t)k]) nm.push_back(s.c_str());
        np1_stream* st = all_in_one ? np1_stream_load(fasta, bam, nullptr, 0, 1) : np1_stream_load(fasta, bam, nm.data(), (int)nm.size(), 1);
        if (st) (void)np1_stream_pin(st);
        return st;
    };
    auto stage = [&](int k, size_t li) -> Item {   // host half of batch k, into lane li's staging buffers
        Item it;
        const double t0 = now_ms();
        std::string lr_err;                // the long-read thread's own error string (merged after the join)
        auto stage_lr = [&] {
            if (device_ingest_lr) {
                np1ingest::Staging* sg = p->staging_lr[li];
                std::string e;
                const int rc = np1ingest::prepare(src_lr, plan[(size_t)k], sg, &e);
                if (rc < 0) { lr_err = e; return; }
                if (rc == 0) { it.staging_lr = sg; return; }
            }
            it.lr = load_host(k, bam_lr);
            if (!it.lr) lr_err = np1_last_error();
        };
        std::thread lr_thread;
        try { lr_thread = std::thread(stage_lr); } catch (const std::system_error&) {}
        if (device_ingest) {
            np1ingest::Staging* sg = p->staging[li];
            std::string e;
            const int rc = np1ingest::prepare(src, plan[(size_t)k], sg, &e);
            if (rc < 0) it.err = e;
            else if (rc == 0) it.staging = sg;
        }
        if (!it.staging && it.err.empty()) {
            it.sr = load_host(k, bam_sr);
            if (!it.sr) it.err = np1_last_error();
        }
        if (lr_thread.joinable()) lr_thread.join();
        else stage_lr();          // no thread to be had: one file after the other
        if (it.err.empty()) it.err = lr_err;
        if (timing_on())
            fprintf(stderr, "[np1 phase] batch %d staged on the host in %.1f ms (short reads: %s, long reads: %s)\n", k, now_ms() - t0, it.staging ? "compressed blocks" : "host loader",
                    it.staging_lr ? "compressed blocks" : "host loader");
        return it;
    };
    auto drop = [](Item& it) { if (it.sr) np1_stream_free(it.sr); if (it.lr) np1_stream_free(it.lr); it.sr = it.lr = nullptr; };
    std::mutex mu;
    std::condition_variable cv;
    int next_emit = 0, rc_all = 0, next_batch = 0;
    std::string err;
    auto lane_work = [&](size_t li) {
        np1_pipe::Lane& ln = p->lanes[li];
        np1_batch* lrb = p->phase_lr[li];
        for (;;) {      // batches in turn to whichever lane is free (results still leave in batch order)
            int k;
            { std::lock_guard<std::mutex> g(mu); if (rc_all != 0 || next_batch >= n) return; k = next_batch++; }
            Item cur = stage(k, li);
            int rc = 0;
            std::string e;
            std::vector<std::string> nm = plan[(size_t)k];
            const double t0 = now_ms();
            if (!cur.err.empty()) { e = cur.err; rc = -1; }
            if (rc == 0 && (cur.staging || cur.staging_lr) && !p->scratch[li]) p->scratch[li] = np1ingest::scratch_create();
            if (rc == 0 && cur.staging) {
                rc = np1ingest::ingest(ln.batch, cur.staging, true, p->scratch[li]);
                if (rc == 1) {   // records the device path does not take: the host loader decodes this batch
                    cur.sr = load_host(k, bam_sr);
                    rc = cur.sr ? 0 : -1;
                } else if (rc == 0) {
                    nm = cur.staging->names();
                }
            }
            if (rc == 0 && cur.sr) rc = np1_batch_reload(ln.batch, cur.sr);
            if (rc == 0 && cur.staging_lr) {
                rc = np1ingest::ingest(lrb, cur.staging_lr, true, p->scratch[li]);
                if (rc == 1) {   // a CIGAR in a CG tag (ultra-long reads): the host loader swaps it in
                    cur.lr = load_host(k, bam_lr);
                    rc = cur.lr ? 0 : -1;
                }
            }
            if (rc == 0 && cur.lr) rc = np1_batch_reload(lrb, cur.lr);
            const double t1 = now_ms();
            if (rc == 0) rc = np1_batch_snp_phase(ln.batch, lrb, cfg);
            const double t2 = now_ms();
            if (rc == 0) rc = np1_batch_results_fetch(ln.batch);
            if (rc != 0 && e.empty()) e = np1_last_error();
            if (timing_on()) fprintf(stderr, "[np1 phase] batch %d (lane %zu): ingest %.1f ms, snp_phase %.1f ms, fetch %.1f ms\n", k, li, t1 - t0, t2 - t1, now_ms() - t2);
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return next_emit == k || rc_all != 0; });
                if (rc != 0 && rc_all == 0) { rc_all = rc; err = e; }
                if (rc_all == 0 && sink) {
                    const uint32_t* b = np1_batch_results_bounds(ln.batch);
                    const char* sq = np1_batch_results_ptr(ln.batch);
                    for (size_t c = 0; c < nm.size(); ++c) sink(user, nm[c].c_str(), sq + b[c], (int64_t)b[c + 1] - (int64_t)b[c]);
                }
                next_emit = k + 1;
                cv.notify_all();
            }
            drop(cur);
            if (rc != 0) return;
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t li = 1; li < L; ++li) {
            try { th.emplace_back(lane_work, li); } catch (const std::system_error&) { break; }
        }
        lane_work(0);
        for (std::thread& t : th) t.join();
    }
    const int rc = rc_all;
    if (rc != 0) { np1_set_error(err); return -1; }
    return 0;
}

}  // extern "C"
