// HBM-side objects behind include/nextpolish1.h, Part 2 (np1_ctx, np1_batch), shared by the launch sequences
// (np1_device.hip) and the device-side ingest (np1_ingest.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "np1_kernels.h"
#include "np1_priv.h"
#include "np_bam.h"
#include "np_devalloc.h"

namespace np1dev {

inline bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    np1_set_error(std::string(what) + ": " + hipGetErrorString(e));
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
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) { p = nullptr; return false; }
        cap = want;
        return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

constexpr int kStages = 8;

}  // namespace np1dev


struct np1_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0[np1dev::kStages], ev1[np1dev::kStages];
};

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
    void swap_work(np1_batch& o) {
        np1dev::DevBuf* mine[] = {&desc, &ovf_desc, &slot_g, &dbg, &single_map, &join_out, &kc_level, &kc_endpos, &kc_code, &kc_flag, &kc_fpos, &kc_flagged, &kc_work, &kc_nd_ctg,
                                  &kc_nd_se, &kc_kr_ctg, &kc_kr_se, &kc_cnt, &kc_sbase, &kc_sflag, &kc_srefk, &kc_scount, &kc_lhead, &kc_lpool, &kc_stsc,
                                  &kc_stkm, &kc_strk, &kc_hpool, &kc_workoff, &kc_nparts, &kc_partoff, &kc_pt_ctg, &kc_pt_se, &kc_pt_len, &kc_woff,
                                  &kc_wpool, &kc_haswin, &sv_failse, &sv_failcnt, &sv_vsz, &sv_voff, &sv_val, &sv_p2ctg, &sv_p2se, &sv_p2len, &sv_woff2,
                                  &sv_haswin2, &sv_range, &qs, &qe, &span, &ins, &soff, &slot_info, &rbase, &capb, &rowoff, &rows, &meta, &chunk_first,
                                  &chunk_last, &slot_res, &slot_rec, &pool, &heads, &redo, &redo2, &redo3, &ctx_lists, &counters, &opos, &out, &bounds, &scan_tmp, &totals};
        np1dev::DevBuf* theirs[] = {&o.desc, &o.ovf_desc, &o.slot_g, &o.dbg, &o.single_map, &o.join_out, &o.kc_level, &o.kc_endpos, &o.kc_code, &o.kc_flag, &o.kc_fpos, &o.kc_flagged,
                                    &o.kc_work, &o.kc_nd_ctg, &o.kc_nd_se, &o.kc_kr_ctg, &o.kc_kr_se, &o.kc_cnt, &o.kc_sbase, &o.kc_sflag, &o.kc_srefk,
                                    &o.kc_scount, &o.kc_lhead, &o.kc_lpool, &o.kc_stsc, &o.kc_stkm, &o.kc_strk, &o.kc_hpool, &o.kc_workoff, &o.kc_nparts,
                                    &o.kc_partoff, &o.kc_pt_ctg, &o.kc_pt_se, &o.kc_pt_len, &o.kc_woff, &o.kc_wpool, &o.kc_haswin, &o.sv_failse,
                                    &o.sv_failcnt, &o.sv_vsz, &o.sv_voff, &o.sv_val, &o.sv_p2ctg, &o.sv_p2se, &o.sv_p2len, &o.sv_woff2, &o.sv_haswin2,
                                    &o.sv_range, &o.qs, &o.qe, &o.span, &o.ins, &o.soff, &o.slot_info, &o.rbase, &o.capb, &o.rowoff, &o.rows, &o.meta,
                                    &o.chunk_first, &o.chunk_last, &o.slot_res, &o.slot_rec, &o.pool, &o.heads, &o.redo, &o.redo2, &o.redo3, &o.ctx_lists, &o.counters, &o.opos,
                                    &o.out, &o.bounds, &o.scan_tmp, &o.totals};
        static_assert(sizeof(mine) == sizeof(theirs), "work buffer lists differ");
        for (size_t i = 0; i < sizeof(mine) / sizeof(mine[0]); ++i) {
            np1dev::DevBuf t = *mine[i];
            *mine[i] = *theirs[i];
            *theirs[i] = t;
        }
        out_cached = false;
        out_pinned = false;
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
