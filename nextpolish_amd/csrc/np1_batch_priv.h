// HBM-side objects behind include/nextpolish1.h, Part 2 (np1_ctx, np1_batch), shared by the launch sequences
// (np1_device.hip) and the device-side ingest (np1_ingest.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "np1_kernels.h"
#include "np1_priv.h"
#include "np_bam.h"
#include "np_devalloc.h"
#include "np_hostcopy.h"

namespace np1dev {

inline bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    np1_set_error(std::string(what) + ": " + hipGetErrorString(e));
    // the caller returns now, and most callers have asynchronous copies into or out of their own locals in flight: nothing of this
    // process may still be moving when those locals go away
    (void)hipDeviceSynchronize();
    return false;
}
#define HIPCHK(x) do { if (!hip_ok((x), #x)) return -1; } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes, double slack = 1.0) {
        if (bytes <= cap && p) return 0;
        if (p) { (void)npalloc::dev_free(p); p = nullptr; cap = 0; }
        // No kernel reads past the bytes it was given (round 4: the wide staging loads and the chase window clamp at the logical end;
        // NP_EFENCE=1 places every buffer against an unmapped page to prove it, np_devalloc.h).  The 64 KiB behind each buffer are
        // defence in depth only; under NP_EFENCE there is no slack at all.
        size_t want = npalloc::efence() ? bytes : (size_t)((double)bytes * slack) + 65536;
        if (!hip_ok(npalloc::dev_malloc(&p, want), "hipMalloc")) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)npalloc::dev_free(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// page-locked host memory that only grows (targets of asynchronous D2H copies)
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap && p) return true;
        if (p) (void)npalloc::host_free(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (npalloc::host_malloc(&p, want, hipHostMallocPortable) != hipSuccess) { p = nullptr; return false; }
        if (npalloc::poison() >= 0) memset(p, npalloc::poison(), want);      // (debugging: a D2H target read before it was written shows)
        cap = want;
        return true;
    }
    void release() { if (p) (void)npalloc::host_free(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

constexpr int kStages = 8;

}  // namespace np1dev


// A context is shared by its owner (np1_ctx_create ... np1_ctx_destroy) and by every batch made on it: the stream and the events live
// until the last of them lets go, so a batch that is freed after its context was destroyed (a Python Batch waiting for the garbage
// collector behind an explicit Context.close()) still finds the stream it has to wait on.
struct np1_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0[np1dev::kStages], ev1[np1dev::kStages];
    std::atomic<int> refs{1};
};
void np1_ctx_retain(np1_ctx* c);      // np1_device.hip
void np1_ctx_release(np1_ctx* c);

struct np1_batch {
    np1_ctx* ctx = nullptr;
    uint32_t nc = 0;
    uint64_t G = 0;
    int64_t n_reads = 0;
    // inputs
    np1dev::DevBuf draft, ctg_off, pos, ctg, flag, ncig, ncig16, lq, cigoff, seqoff, cigar, seq, seq2, esc_at, esc_val, up_plain, up_xlq, up_xncig, up_xcigar, up_dpos, up_xpos, up_work, draft4, desc_at, desc_val;
    // work
    np1dev::DevBuf desc, ovf_desc, slot_g, dbg, single_map, join_out;
    bool keep_single = false;   // intra-contig tiling: keep which slots left the vote with one state (np1_batch_tile_join)
    // kmer_count inputs (uploaded only when the stream carries qualities) and work buffers
    np1dev::DevBuf mapq, isize, qualoff, qual, read_begin;
    np1dev::DevBuf kc_level, kc_endpos, kc_code, kc_flag, kc_fpos, kc_flagged, kc_work, kc_nd_ctg, kc_nd_se, kc_kr_ctg, kc_kr_se, kc_cnt,
        kc_sbase, kc_sflag, kc_srefk, kc_scount, kc_lhead, kc_lpool, kc_stsc, kc_stkm, kc_strk, kc_hpool, kc_workoff, kc_nparts,
        kc_partoff, kc_pt_ctg, kc_pt_se, kc_pt_len, kc_woff, kc_wpool, kc_haswin;
    // snp_valid: second-round work (regions nothing spanned, their split values and parts)
    np1dev::DevBuf sv_failse, sv_failcnt, sv_vsz, sv_voff, sv_val, sv_p2ctg, sv_p2se, sv_p2len, sv_woff2, sv_haswin2, sv_range;
    std::vector<np1dev::DevBuf> spw;   // snp_phase work buffers (np1_phase_device.hip), owned by the short-read batch
    // kmer_count / snp_valid with the reference's region iterator replayed (np1_replay.h): the BAM index, the BAM reference id of every
    // contig and the records' virtual offsets -- a view of the host stream the batch was filled from (np1_batch_enable_replay; the stream
    // must outlive the pass) or arrays the device-side ingest brought down (np1_ingest.hip); positions and end positions are fetched from
    // the device once per pass
    struct Replay {
        bool on = false;
        const np::BaiIndex* bai = nullptr;     // the pipe's index, or own_bai
        np::BaiIndex own_bai;
        std::string own_bai_path;
        std::vector<int32_t> tid;
        np1dev::PinBuf pos, endpos;            // int32 per record
        bool have_pos = false;
        const uint64_t *voff = nullptr, *voff_end = nullptr;
        np1dev::PinBuf own_voff, own_voff_end; // uint64 per record
        np1dev::DevBuf first, list, stale, n2, brk, snap;
        uint32_t revotes = 0;                  // diagnostics: vote launches the max_count_kmer break made necessary in the last pass
    } replay;
    bool has_qual = false;
    std::vector<uint64_t> h_read_begin;
    np1dev::DevBuf qs, qe, span, ins, soff, slot_info, rbase, capb, rowoff, rows, meta, chunk_first, chunk_last, slot_res,
        slot_rec, pool, heads, redo, redo2, redo3, ctx_lists, counters, opos, out, bounds, scan_tmp, totals;
    size_t input_bytes = 0;
    uint32_t max_lq = 0;   // longest record of the batch (bases)
    uint32_t last_counters[np1k::CNT_WORDS] = {0};
    bool force_staged = false;   // a record exceeded the descriptor capacity once: this batch uses the staged sequence
    // results of the last run
    uint32_t S = 0;
    uint64_t votes = 0;
    bool ran = false, out_cached = false, out_pinned = false;
    uint8_t* h_pin = nullptr;   // pinned copy of `out` (np1_batch_results_fetch)
    size_t h_pin_cap = 0;
    std::vector<uint32_t> h_bounds;
    std::vector<uint8_t> h_out;
    std::vector<uint32_t> h_ctg_off;

    // Everything a pass allocates besides the uploaded inputs (record arrays, draft, qualities) changes places with the same
    // buffers of another batch: resident batches keep only their inputs in HBM and borrow the work set of the lane they run on
    // (np1_pipe_run_resident), so a draft of any size stays resident with two work sets instead of one per batch.
    // the buffers a run computes into (everything but the uploaded records and their forms): what two lanes of the pipe swap, and what the
    // debugging poison fills before every run
    template <class F> static void for_each_work(F f) {
        static np1dev::DevBuf np1_batch::* const list[] = {&np1_batch::desc, &np1_batch::ovf_desc, &np1_batch::slot_g, &np1_batch::dbg,
            &np1_batch::single_map, &np1_batch::join_out, &np1_batch::kc_level, &np1_batch::kc_endpos, &np1_batch::kc_code, &np1_batch::kc_flag,
            &np1_batch::kc_fpos, &np1_batch::kc_flagged, &np1_batch::kc_work, &np1_batch::kc_nd_ctg, &np1_batch::kc_nd_se, &np1_batch::kc_kr_ctg,
            &np1_batch::kc_kr_se, &np1_batch::kc_cnt, &np1_batch::kc_sbase, &np1_batch::kc_sflag, &np1_batch::kc_srefk, &np1_batch::kc_scount,
            &np1_batch::kc_lhead, &np1_batch::kc_lpool, &np1_batch::kc_stsc, &np1_batch::kc_stkm, &np1_batch::kc_strk, &np1_batch::kc_hpool,
            &np1_batch::kc_workoff, &np1_batch::kc_nparts, &np1_batch::kc_partoff, &np1_batch::kc_pt_ctg, &np1_batch::kc_pt_se,
            &np1_batch::kc_pt_len, &np1_batch::kc_woff, &np1_batch::kc_wpool, &np1_batch::kc_haswin, &np1_batch::sv_failse, &np1_batch::sv_failcnt,
            &np1_batch::sv_vsz, &np1_batch::sv_voff, &np1_batch::sv_val, &np1_batch::sv_p2ctg, &np1_batch::sv_p2se, &np1_batch::sv_p2len,
            &np1_batch::sv_woff2, &np1_batch::sv_haswin2, &np1_batch::sv_range, &np1_batch::qs, &np1_batch::qe, &np1_batch::span, &np1_batch::ins,
            &np1_batch::soff, &np1_batch::slot_info, &np1_batch::rbase, &np1_batch::capb, &np1_batch::rowoff, &np1_batch::rows, &np1_batch::meta,
            &np1_batch::chunk_first, &np1_batch::chunk_last, &np1_batch::slot_res, &np1_batch::slot_rec, &np1_batch::pool, &np1_batch::heads,
            &np1_batch::redo, &np1_batch::redo2, &np1_batch::redo3, &np1_batch::ctx_lists, &np1_batch::counters, &np1_batch::opos, &np1_batch::out,
            &np1_batch::bounds, &np1_batch::scan_tmp, &np1_batch::totals};
        for (np1dev::DevBuf np1_batch::* m : list) f(m);
    }
    void swap_work(np1_batch& o) {
        for_each_work([&](np1dev::DevBuf np1_batch::* m) {
            np1dev::DevBuf t = this->*m;
            this->*m = o.*m;
            o.*m = t;
        });
        out_cached = false;
        out_pinned = false;
    }
    // NP_DEVPOISON (np_devalloc.h) set to 2..: besides new allocations, every work buffer is filled with a5 at the start of every run, so a
    // kernel that reads what an EARLIER run left in a buffer -- the other thing a long-lived process has and a fresh one has not -- shows
    void poison_work(hipStream_t q) {
        for_each_work([&](np1dev::DevBuf np1_batch::* m) {
            np1dev::DevBuf& d = this->*m;
            if (d.p && d.cap) (void)hipMemsetAsync(d.p, 0xa5, d.cap, q);
        });
    }
    size_t device_bytes() const {
        const np1dev::DevBuf* all[] = {&draft, &ctg_off, &pos, &ctg, &flag, &ncig, &ncig16, &lq, &cigoff, &seqoff, &cigar, &seq, &seq2, &esc_at, &esc_val, &up_plain, &up_xlq, &up_xncig, &up_xcigar, &up_dpos, &up_xpos, &up_work, &draft4, &desc_at, &desc_val, &qs, &qe,
                               &span, &ins, &soff, &slot_info, &rbase, &capb, &rowoff, &rows, &meta, &chunk_first,
                               &chunk_last, &slot_res, &slot_rec, &pool, &heads, &redo, &redo2, &redo3, &ctx_lists, &counters, &opos, &out,
                               &bounds, &scan_tmp, &totals, &desc, &ovf_desc, &slot_g, &dbg, &single_map, &join_out, &mapq, &isize, &qualoff, &qual, &read_begin,
                               &kc_level, &kc_endpos, &kc_code, &kc_flag, &kc_fpos, &kc_flagged, &kc_work, &kc_nd_ctg, &kc_nd_se,
                               &kc_kr_ctg, &kc_kr_se, &kc_cnt, &kc_sbase, &kc_sflag, &kc_srefk, &kc_scount, &kc_lhead, &kc_lpool,
                               &kc_stsc, &kc_stkm, &kc_strk, &kc_hpool, &kc_workoff, &kc_nparts, &kc_partoff, &kc_pt_ctg,
                               &kc_pt_se, &kc_pt_len, &kc_woff, &kc_wpool, &kc_haswin, &sv_failse, &sv_failcnt, &sv_vsz, &sv_voff, &sv_val,
                               &sv_p2ctg, &sv_p2se, &sv_p2len, &sv_woff2, &sv_haswin2, &sv_range};
        size_t t = 0;
        for (const np1dev::DevBuf* b : all) t += b->cap;
        for (const np1dev::DevBuf& b : spw) t += b.cap;
        return t;
    }
    void release_all() {
        np1dev::DevBuf* all[] = {&draft, &ctg_off, &pos, &ctg, &flag, &ncig, &ncig16, &lq, &cigoff, &seqoff, &cigar, &seq, &seq2, &esc_at, &esc_val, &up_plain, &up_xlq, &up_xncig, &up_xcigar, &up_dpos, &up_xpos, &up_work, &draft4, &desc_at, &desc_val, &qs, &qe,
                         &span, &ins, &soff, &slot_info, &rbase, &capb, &rowoff, &rows, &meta, &chunk_first,
                         &chunk_last, &slot_res, &slot_rec, &pool, &heads, &redo, &redo2, &redo3, &ctx_lists, &counters, &opos, &out,
                         &bounds, &scan_tmp, &totals, &desc, &ovf_desc, &slot_g, &dbg, &single_map, &join_out, &mapq, &isize, &qualoff, &qual, &read_begin,
                         &kc_level, &kc_endpos, &kc_code, &kc_flag, &kc_fpos, &kc_flagged, &kc_work, &kc_nd_ctg, &kc_nd_se,
                         &kc_kr_ctg, &kc_kr_se, &kc_cnt, &kc_sbase, &kc_sflag, &kc_srefk, &kc_scount, &kc_lhead, &kc_lpool,
                         &kc_stsc, &kc_stkm, &kc_strk, &kc_hpool, &kc_workoff, &kc_nparts, &kc_partoff, &kc_pt_ctg, &kc_pt_se,
                         &kc_pt_len, &kc_woff, &kc_wpool, &kc_haswin, &sv_failse, &sv_failcnt, &sv_vsz, &sv_voff, &sv_val, &sv_p2ctg,
                         &sv_p2se, &sv_p2len, &sv_woff2, &sv_haswin2, &sv_range};
        for (np1dev::DevBuf* b : all) b->release();
        for (np1dev::DevBuf& b : spw) b.release();
        replay.first.release(); replay.list.release(); replay.stale.release(); replay.n2.release(); replay.brk.release(); replay.snap.release();
        replay.pos.release(); replay.endpos.release(); replay.own_voff.release(); replay.own_voff_end.release();
    }
};


// replay of the reference's region iterator for a batch the device-side ingest filled (np1_device.hip)
int np1_batch_enable_replay_ingested(np1_batch* b, const np::BaiIndex* bai, const std::vector<int32_t>& tid);
