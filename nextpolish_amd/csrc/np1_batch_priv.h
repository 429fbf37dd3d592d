// HBM-side objects behind include/nextpolish1.h, Part 2 (np1_ctx, np1_batch), shared by the launch sequences
// (np1_device.hip) and the device-side ingest (np1_ingest.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "np1_kernels.h"
#include "np1_priv.h"

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
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = (size_t)((double)bytes * slack) + 256;
        if (!hip_ok(hipMalloc(&p, want), "hipMalloc")) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
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
    np1dev::DevBuf draft, ctg_off, pos, ctg, flag, ncig, lq, cigoff, seqoff, cigar, seq;
    // work
    np1dev::DevBuf desc, ovf_desc, slot_g, dbg;
    // kmer_count inputs (uploaded only when the stream carries qualities) and work buffers
    np1dev::DevBuf mapq, isize, qualoff, qual, read_begin;
    np1dev::DevBuf kc_level, kc_endpos, kc_code, kc_flag, kc_fpos, kc_flagged, kc_work, kc_nd_ctg, kc_nd_se, kc_kr_ctg, kc_kr_se, kc_cnt,
        kc_sbase, kc_sflag, kc_srefk, kc_scount, kc_lhead, kc_lpool, kc_stsc, kc_stkm, kc_strk, kc_hpool, kc_workoff, kc_nparts,
        kc_partoff, kc_pt_ctg, kc_pt_se, kc_pt_len, kc_woff, kc_wpool, kc_haswin;
    // snp_valid: second-round work (regions nothing spanned, their split values and parts)
    np1dev::DevBuf sv_failse, sv_failcnt, sv_vsz, sv_voff, sv_val, sv_p2ctg, sv_p2se, sv_p2len, sv_woff2, sv_haswin2, sv_range;
    bool has_qual = false;
    std::vector<uint64_t> h_read_begin;
    np1dev::DevBuf qs, qe, span, ins, soff, slot_info, rbase, capb, rowoff, rows, meta, chunk_first, chunk_last, slot_res,
        slot_rec, pool, heads, redo, redo2, counters, opos, out, bounds, scan_tmp, totals;
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

    size_t device_bytes() const {
        const np1dev::DevBuf* all[] = {&draft, &ctg_off, &pos, &ctg, &flag, &ncig, &lq, &cigoff, &seqoff, &cigar, &seq, &qs, &qe,
                               &span, &ins, &soff, &slot_info, &rbase, &capb, &rowoff, &rows, &meta, &chunk_first,
                               &chunk_last, &slot_res, &slot_rec, &pool, &heads, &redo, &redo2, &counters, &opos, &out,
                               &bounds, &scan_tmp, &totals, &desc, &ovf_desc, &slot_g, &mapq, &isize, &qualoff, &qual, &read_begin,
                               &kc_level, &kc_endpos, &kc_code, &kc_flag, &kc_fpos, &kc_flagged, &kc_work, &kc_nd_ctg, &kc_nd_se,
                               &kc_kr_ctg, &kc_kr_se, &kc_cnt, &kc_sbase, &kc_sflag, &kc_srefk, &kc_scount, &kc_lhead, &kc_lpool,
                               &kc_stsc, &kc_stkm, &kc_strk, &kc_hpool, &kc_workoff, &kc_nparts, &kc_partoff, &kc_pt_ctg,
                               &kc_pt_se, &kc_pt_len, &kc_woff, &kc_wpool, &kc_haswin, &sv_failse, &sv_failcnt, &sv_vsz, &sv_voff, &sv_val,
                               &sv_p2ctg, &sv_p2se, &sv_p2len, &sv_woff2, &sv_haswin2, &sv_range};
        size_t t = 0;
        for (const np1dev::DevBuf* b : all) t += b->cap;
        return t;
    }
    void release_all() {
        np1dev::DevBuf* all[] = {&draft, &ctg_off, &pos, &ctg, &flag, &ncig, &lq, &cigoff, &seqoff, &cigar, &seq, &qs, &qe,
                         &span, &ins, &soff, &slot_info, &rbase, &capb, &rowoff, &rows, &meta, &chunk_first,
                         &chunk_last, &slot_res, &slot_rec, &pool, &heads, &redo, &redo2, &counters, &opos, &out,
                         &bounds, &scan_tmp, &totals, &desc, &ovf_desc, &slot_g, &mapq, &isize, &qualoff, &qual, &read_begin,
                         &kc_level, &kc_endpos, &kc_code, &kc_flag, &kc_fpos, &kc_flagged, &kc_work, &kc_nd_ctg, &kc_nd_se,
                         &kc_kr_ctg, &kc_kr_se, &kc_cnt, &kc_sbase, &kc_sflag, &kc_srefk, &kc_scount, &kc_lhead, &kc_lpool,
                         &kc_stsc, &kc_stkm, &kc_strk, &kc_hpool, &kc_workoff, &kc_nparts, &kc_partoff, &kc_pt_ctg, &kc_pt_se,
                         &kc_pt_len, &kc_woff, &kc_wpool, &kc_haswin, &sv_failse, &sv_failcnt, &sv_vsz, &sv_voff, &sv_val, &sv_p2ctg,
                         &sv_p2se, &sv_p2len, &sv_woff2, &sv_haswin2, &sv_range};
        for (np1dev::DevBuf* b : all) b->release();
    }
};

