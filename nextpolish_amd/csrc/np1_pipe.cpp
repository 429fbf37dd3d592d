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
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np1_priv.h"
#include "np_bam.h"

struct np1_pipe {
    int device = 0;
    struct Lane { np1_ctx* ctx = nullptr; np1_batch* batch = nullptr; };
    std::vector<Lane> lanes;
    // results of the last np1_pipe_run
    std::vector<std::vector<char>> out;            // per batch: concatenated strings
    std::vector<std::vector<uint32_t>> bounds;     // per batch: nc + 1 offsets
    std::vector<np1_batch*> resident;              // np1_pipe_upload: batch k lives on lane k % lanes
};

namespace {

int polish_on_lane(np1_pipe::Lane& ln, np1_stream* st, const Configure* cfg, int task) {
    if (np1_batch_reload(ln.batch, st) != 0) return -1;
    const int rc = task == 2 ? np1_batch_kmer_count(ln.batch, cfg, nullptr) : np1_batch_score_chain(ln.batch, cfg, nullptr);
    if (rc != 0) return -1;
    return np1_batch_results_fetch(ln.batch);
}

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

int np1_pipe_run_resident(np1_pipe* p, const Configure* cfg, int task, int passes) {
    if (!p || !cfg) { np1_set_error("np1_pipe_run_resident: null argument"); return -1; }
    std::atomic<bool> failed(false);
    std::string err;
    std::mutex err_mu;
    const size_t nl = p->lanes.size();
    auto work = [&](size_t lane) {
        for (int pass = 0; pass < passes && !failed; ++pass)
            for (size_t k = lane; k < p->resident.size(); k += nl) {
                np1_batch* b = p->resident[k];
                const int rc = task == 2 ? np1_batch_kmer_count(b, cfg, nullptr) : np1_batch_score_chain(b, cfg, nullptr);
                if (rc != 0) {
                    std::lock_guard<std::mutex> g(err_mu);
                    if (!failed) err = np1_last_error();
                    failed = true;
                    return;
                }
            }
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < nl; ++i) th.emplace_back(work, i);
    work(0);
    for (std::thread& t : th) t.join();
    if (failed) { np1_set_error(err); return -1; }
    return 0;
}

void np1_pipe_close(np1_pipe* p) {
    if (!p) return;
    drop_resident(p);
    for (np1_pipe::Lane& ln : p->lanes) {
        if (ln.batch) np1_batch_free(ln.batch);
        if (ln.ctx) np1_ctx_destroy(ln.ctx);
    }
    delete p;
}

int np1_pipe_run(np1_pipe* p, np1_stream* const* streams, int n, const Configure* cfg, int task) {
    if (!p || !cfg || (n > 0 && !streams)) { np1_set_error("np1_pipe_run: null argument"); return -1; }
    p->out.assign((size_t)n, {});
    p->bounds.assign((size_t)n, {});
    std::atomic<int> next(0);
    std::atomic<bool> failed(false);
    std::string err;
    std::mutex err_mu;
    auto work = [&](np1_pipe::Lane& ln) {
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= n || failed) break;
            if (polish_on_lane(ln, streams[k], cfg, task) != 0) {
                std::lock_guard<std::mutex> g(err_mu);
                if (!failed) err = np1_last_error();
                failed = true;
                break;
            }
            np1_stream_view v;
            np1_stream_get_view(streams[k], &v);
            const uint32_t* b = np1_batch_results_bounds(ln.batch);
            p->bounds[(size_t)k].assign(b, b + v.n_contigs + 1);
            const char* s = np1_batch_results_ptr(ln.batch);
            p->out[(size_t)k].assign(s, s + b[v.n_contigs]);
        }
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < p->lanes.size() && (int)i < n; ++i) th.emplace_back(work, std::ref(p->lanes[i]));
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
    return p->out[(size_t)batch].data() + b[(size_t)contig];
}

int np1_pipe_run_files(np1_pipe* p, const char* fasta, const char* bam, const char* const* names, int n_names, int64_t batch_bp,
                       const Configure* cfg, int task, np1_sink_fn sink, void* user) {
    if (!p || !fasta || !bam || !cfg) { np1_set_error("np1_pipe_run_files: null argument"); return -1; }
    np::Fai fai;
    if (!fai.load(fasta)) { np1_set_error(std::string("cannot load FASTA/index: ") + fasta); return -1; }
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
    const size_t depth = p->lanes.size() + loader_threads();    // loaded-but-unpolished batches allowed in memory
    std::mutex mu;
    std::condition_variable cv;
    std::map<int, np1_stream*> ready;      // loaded batches waiting for a lane
    std::map<int, np1_stream*> done;       // polished batches waiting for their turn at the sink (stream kept for the names)
    std::map<int, std::pair<std::vector<char>, std::vector<uint32_t>>> done_out;
    int next_load = 0, next_polish = 0, next_emit = 0, in_memory = 0;
    bool failed = false;
    std::string err;
    const bool with_qual = task == 2;
    auto fail = [&](const std::string& e) {
        std::lock_guard<std::mutex> g(mu);
        if (!failed) err = e;
        failed = true;
        cv.notify_all();
    };
    auto loader = [&]() {
        for (;;) {
            int k;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return failed || next_load >= n || (size_t)in_memory < depth; });
                if (failed || next_load >= n) return;
                k = next_load++;
                ++in_memory;
            }
            std::vector<const char*> nm;
            for (const std::string& s : plan[(size_t)k]) nm.push_back(s.c_str());
            np1_stream* st = np1_stream_load(fasta, bam, nm.data(), (int)nm.size(), with_qual ? 1 : 0);
            if (!st) { fail(np1_last_error()); return; }
            (void)np1_stream_pin(st);   // best effort: an unpinned stream still uploads, just synchronously
            std::lock_guard<std::mutex> g(mu);
            ready[k] = st;
            cv.notify_all();
        }
    };
    auto emit_ready = [&](std::unique_lock<std::mutex>& g) {   // called with mu held
        while (done.count(next_emit)) {
            np1_stream* st = done[next_emit];
            auto res = std::move(done_out[next_emit]);
            done.erase(next_emit);
            done_out.erase(next_emit);
            const int k = next_emit++;
            (void)k;
            g.unlock();
            if (sink) {
                np1_stream_view v;
                np1_stream_get_view(st, &v);
                for (int64_t c = 0; c < v.n_contigs; ++c)
                    sink(user, np1_stream_contig_name(st, c), res.first.data() + res.second[(size_t)c],
                         (int64_t)res.second[(size_t)c + 1] - (int64_t)res.second[(size_t)c]);
            }
            np1_stream_free(st);
            g.lock();
            --in_memory;
            cv.notify_all();
        }
    };
    bool emitting = false;
    auto lane_work = [&](np1_pipe::Lane& ln) {
        for (;;) {
            int k;
            np1_stream* st;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return failed || next_polish >= n || ready.count(next_polish); });
                if (failed || next_polish >= n) return;
                k = next_polish++;
                st = ready[k];
                ready.erase(k);
            }
            if (polish_on_lane(ln, st, cfg, task) != 0) { fail(np1_last_error()); return; }
            np1_stream_view v;
            np1_stream_get_view(st, &v);
            const uint32_t* b = np1_batch_results_bounds(ln.batch);
            const char* s = np1_batch_results_ptr(ln.batch);
            std::pair<std::vector<char>, std::vector<uint32_t>> res;
            res.second.assign(b, b + v.n_contigs + 1);
            res.first.assign(s, s + b[v.n_contigs]);
            for (uint32_t& x : res.second) (void)x;
            // strings are not NUL-terminated inside the blob: the sink gets (pointer, length)
            std::unique_lock<std::mutex> g(mu);
            done[k] = st;
            done_out[k] = std::move(res);
            if (!emitting) {          // one thread at a time drains the in-order queue
                emitting = true;
                emit_ready(g);
                emitting = false;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    const unsigned nl = std::min<unsigned>(loader_threads(), (unsigned)std::max(1, n));
    for (unsigned i = 0; i < nl; ++i) th.emplace_back(loader);
    for (size_t i = 1; i < p->lanes.size(); ++i) th.emplace_back(lane_work, std::ref(p->lanes[i]));
    lane_work(p->lanes[0]);
    for (std::thread& t : th) t.join();
    {   // whatever finished out of turn while another thread was emitting
        std::unique_lock<std::mutex> g(mu);
        if (!failed) emit_ready(g);
        for (auto& kv : ready) np1_stream_free(kv.second);
        for (auto& kv : done) np1_stream_free(kv.second);
    }
    if (failed) { np1_set_error(err); return -1; }
    return 0;
}

}  // extern "C"
