// Launch wrappers of the kmer_count kernels (np1_kmer_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "np1_kmer.h"

namespace np1k {

// kmer_count device counters (uint32 words)
enum { KCC_NODEPTH = 0, KCC_KREG = 1, KCC_ND_LEN = 2 /* u64: words 2,3 */, KCC_MAXSPAN = 4, KCC_LCOUNT = 5, KCC_STCOUNT = 6,
       KCC_HCOUNT = 7, KCC_ERR = 8, KCC_WORDS = 16 };

void kc_launch_records(hipStream_t st, const KcCtx& c, int64_t n_all, uint8_t* level, int32_t* endpos, uint32_t* max_span);
void kc_launch_draft(hipStream_t st, const uint8_t* draft, uint32_t G, uint8_t* code, uint8_t* flag);
void kc_launch_compact(hipStream_t st, const uint8_t* flag, const uint32_t* fpos, uint32_t G, uint32_t* flagged);
void kc_launch_regions(hipStream_t st, const KcCtx& c, uint32_t nc, const uint32_t* fpos, uint32_t* flagged, int32_t* work,
                       uint32_t* nd_ctg, int32_t* nd_se, uint32_t* kr_ctg, int32_t* kr_se, uint32_t reg_cap, uint32_t* counters);
void kc_launch_inserts(hipStream_t st, const KcCtx& c, const uint32_t* kr_ctg, const int32_t* kr_se, uint32_t n_kr,
                       const uint32_t* nd_ctg, const int32_t* nd_se, uint32_t n_nd, uint32_t* ins);
void kc_launch_slots(hipStream_t st, const uint8_t* slot_info, uint32_t S, uint8_t* sbase, uint8_t* sflag, uint16_t* scount,
                     uint32_t* lhead);
void kc_launch_nodepth(hipStream_t st, const KcCtx& c, const uint32_t* nd_ctg, const int32_t* nd_se, uint32_t n_nd);
void kc_launch_split(hipStream_t st, const KcCtx& c, const uint32_t* kr_ctg, const int32_t* kr_se, uint32_t n_kr, int32_t* work,
                     const uint32_t* work_off, uint32_t* n_parts, const uint32_t* part_off, uint32_t* pt_ctg, int32_t* pt_se,
                     uint32_t* pt_len);
void kc_launch_winner(hipStream_t st, const KcCtx& c, const uint32_t* pt_ctg, const int32_t* pt_se, const uint32_t* pt_len,
                      const uint32_t* woff, uint32_t n_parts, int64_t n_all, uint8_t* wpool, uint8_t* has_winner);
// the same vote on what the replayed region iterator hands out per part (np1_replay.h); rp_n2 == nullptr: first pass, has_winner 2 = needs the second loop;
// rp_brk (optional): per part, the number of records after which the first loop left through the max_count_kmer break (0 = it did not)
void kc_launch_winner_replay(hipStream_t st, const KcCtx& c, const uint32_t* pt_ctg, const int32_t* pt_se, const uint32_t* pt_len, const uint32_t* woff, uint32_t n_parts,
                             int64_t n_all, uint8_t* wpool, uint8_t* has_winner, const uint32_t* rp_first, const uint32_t* rp_list, const long long* rp_stale,
                             const int32_t* rp_n2, uint32_t* rp_brk);
void kc_launch_apply(hipStream_t st, const KcCtx& c, const uint32_t* pt_ctg, const int32_t* pt_se, const uint32_t* pt_len,
                     const uint32_t* woff, uint32_t n_parts, const uint8_t* wpool, const uint8_t* has_winner);
void kc_launch_result(hipStream_t st, const uint8_t* sbase, const uint8_t* sflag, uint32_t S, uint16_t* slot_res);

// snp_valid (task 4): what happens around the votes of the two rounds (np1_kmer_kernels.hip)
void sv_launch_val_sizes(hipStream_t st, const uint32_t* pt_len, uint32_t n_parts, uint32_t* vsz);
void sv_launch_ranges(hipStream_t st, const uint32_t* pt_ctg, uint32_t n_parts, uint32_t* range);   // [first, last) part of every contig
void sv_launch_round1(hipStream_t st, const KcCtx& c, uint32_t nc, const uint32_t* range, const int32_t* pt_se, const uint32_t* pt_len,
                      const uint32_t* woff, uint32_t n_parts, const uint8_t* wpool, const uint8_t* has_winner, int32_t* fail_se,
                      uint32_t* fail_cnt);
void sv_launch_round2_parts(hipStream_t st, const KcCtx& c, uint32_t nc, const uint32_t* range, const int32_t* pt_se, uint32_t n_parts,
                            const uint32_t* voff, const int32_t* fail_se, const uint32_t* fail_cnt, int32_t* val, uint32_t* p2_ctg,
                            int32_t* p2_se, uint32_t* p2_len);
void sv_launch_round2_apply(hipStream_t st, const KcCtx& c, uint32_t nc, const uint32_t* range, uint32_t n_parts, const uint32_t* voff,
                            const uint32_t* p2_ctg, const int32_t* p2_se, const uint32_t* p2_len, const uint32_t* woff2,
                            const uint8_t* wpool, const uint8_t* has_winner);

// generic exclusive scans of np1_kernels.hip reused here
void launch_scan_u8(hipStream_t st, const uint8_t* v, uint64_t n, uint32_t* out, uint64_t* tmp, uint64_t* total);
void launch_scan_u32(hipStream_t st, const uint32_t* v, uint64_t n, uint32_t* out, uint64_t* tmp, uint64_t* total);

}  // namespace np1k
