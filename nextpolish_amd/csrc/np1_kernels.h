// Device-side layout constants and launch wrappers shared by np1_kernels.hip and np1_device.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "np1_core.h"
#include "np1_desc.h"

namespace np1k {

uint64_t scan_tmp_words(uint64_t n);

void launch_prep(hipStream_t st, const ReadsDev& R, int64_t n_reads, const uint32_t* ctg_off, int trim, int32_t* qs,
                 int32_t* qe, int32_t* span, uint32_t* ins, uint32_t* counters);
void launch_scan_slots(hipStream_t st, const uint32_t* ins, uint64_t G, uint32_t* soff, uint64_t* tmp, uint64_t* total);
void launch_scan_rows(hipStream_t st, const uint32_t* cap_bytes, uint64_t n, uint64_t* rowoff, uint64_t* tmp, uint64_t* total);
void launch_scan_keep(hipStream_t st, const uint16_t* slot_res, uint64_t S, uint32_t* opos, uint64_t* tmp, uint64_t* total);
void launch_slotinfo(hipStream_t st, const uint8_t* draft, uint32_t G, const uint32_t* ctg_off, uint32_t nc,
                     const uint32_t* soff, uint8_t* slot_info, uint32_t* slot_g);
void launch_rowcap(hipStream_t st, const ReadsDev& R, int64_t n_reads, const uint32_t* ctg_off, const uint32_t* soff,
                   const int32_t* qs, const int32_t* qe, const int32_t* span, uint32_t* rbase, uint32_t* cap_bytes);
void launch_rows(hipStream_t st, const ReadsDev& R, int64_t n_reads, const uint32_t* ctg_off, const uint32_t* soff,
                 const int32_t* qs, const int32_t* qe, const uint32_t* rbase, const uint64_t* rowoff, uint8_t* rows,
                 uint4* meta, uint32_t* chunk_first, uint32_t* chunk_last, unsigned long long* votes);
inline size_t vote_hbm_list_words() { return (size_t)(np1k::VOTE_E_ALL - 2) * 64; }   // per chunk, for launch_vote with E > 160
void launch_vote(hipStream_t st, int E, const uint4* meta, const uint8_t* rows, const uint8_t* slot_info, uint32_t S,
                 const uint32_t* chunk_first, const uint32_t* chunk_last, uint32_t n_chunks, const uint32_t* redo_in,
                 uint32_t n_redo_in, uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool, uint32_t pool_cap,
                 uint32_t* counters, uint32_t* heads, uint32_t* redo_out, uint32_t redo_ci, uint32_t flag_single, uint32_t* hbm_lists = nullptr);
void launch_desc(hipStream_t st, const ReadsDev& R, int64_t n_reads, const uint32_t* ctg_off, const uint32_t* soff,
                 const int32_t* qs, const int32_t* qe, uint32_t* desc, uint32_t* ovf_pool, uint32_t ovf_cap,
                 uint32_t* chunk_first, uint32_t* chunk_last, uint32_t* counters);
// cigar_off / seq_off / ctg of a dense record stream on the device (cigoff and seqoff get n + 1 entries)
void launch_widen_u16(hipStream_t st, const uint16_t* src, uint32_t* dst, uint64_t n);
// the compact upload form of the per-record fields (np1_priv.h: np1_stream::Compact) -> pos, n_cigar, l_qseq and the operation pool;
// cigoff = the running sums of n_cigar (launch_record_offsets, between the two halves).  work: 3 * (n + 1) + nx + 2 words of 8 bytes.
// the 4-bit upload form of the draft (np1_priv.h: np1_stream::draft4) -> its characters; exceptions patched with launch_unpack_seq2's kernel
void launch_unpack_draft4(hipStream_t st, const uint8_t* d4, uint64_t G, uint8_t* draft, const uint64_t* esc_at, const uint8_t* esc_val, uint64_t n_esc);
struct CompactDev {
    const uint32_t* plain; const int32_t* x_lq; const uint32_t* x_ncig; const uint32_t* x_cigar; const uint8_t* dpos; const int32_t* x_pos;
    uint32_t common_lq; uint64_t n, nx, n_xpos;
};
void launch_expand_records(hipStream_t st, const CompactDev& c, int32_t* pos, uint32_t* ncig, int32_t* lq, uint64_t* work, uint64_t* tmp, uint64_t* total);
void launch_expand_cigars(hipStream_t st, const CompactDev& c, const uint32_t* ncig, const uint64_t* cigoff, uint32_t* cigar, uint64_t* work, uint64_t* tmp, uint64_t* total);
// the 2-bit upload form of the packed bases (np1_priv.h: np1_stream::seq2) expanded to the 4-bit codes the kernels read, then the
// exception bytes put in place; seq must hold 2 * n2 bytes
void launch_unpack_seq2(hipStream_t st, const uint8_t* seq2, uint64_t n2, uint8_t* seq, const uint64_t* esc_at, const uint8_t* esc_val, uint64_t n_esc);   // the 16-bit upload form of the CIGAR operation counts
void launch_record_offsets(hipStream_t st, const uint32_t* ncig, const int32_t* lq, const uint64_t* read_begin, uint32_t nc, uint64_t n, uint64_t* cigoff,
                           uint64_t* seqoff, uint32_t* ctg, uint64_t* tmp, uint64_t* total);
// default fused kernel (descriptors + packed bases staged through LDS); levels as launch_tile
int launch_tile3(hipStream_t st, int level, const ReadsDev& R, const uint32_t* soff, const uint32_t* desc,
                 const uint32_t* ovf_pool, const uint32_t* chunk_first, const uint32_t* chunk_last, uint32_t n_chunks, const uint32_t* redo_in,
                 uint32_t n_redo_in, const uint8_t* slot_info, const uint32_t* slot_g, uint32_t S, uint32_t max_lq,
                 uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool, uint32_t pool_cap, uint32_t* counters,
                 uint32_t* heads, uint32_t heads_cap, uint32_t* redo_out, uint32_t redo_ci, uint32_t flag_single,
                 unsigned long long* votes);
// k_tile9 (default since round 4, np1_tile9.h): four slots per lane, agreeing records counted per window, the rest deferred and tallied in
// record order; the vote chunks of a wave that runs out of room land in redo_out for launch_tile3(level 1)
int launch_tile9(hipStream_t st, const ReadsDev& R, const uint32_t* soff, const uint32_t* desc, const uint32_t* ovf_pool, const uint32_t* chunk_first,
                 const uint32_t* chunk_last, uint32_t n_chunks, const uint8_t* slot_info, const uint32_t* slot_g, uint32_t S, uint32_t max_lq,
                 uint16_t* slot_res, uint32_t* slot_rec, uint32_t* pool, uint32_t pool_cap, uint32_t* counters, uint32_t* heads, uint32_t heads_cap,
                 uint32_t* redo_out, uint32_t redo_ci, uint32_t flag_single, unsigned long long* votes, unsigned long long* dbg = nullptr);
// intra-contig tiling: single-state map of the vote, and a tile's join facts (out[4]: left halo has a single-state slot, right halo has one,
// output offset of the tile's first own base, of the first base behind it)
void launch_single_map(hipStream_t st, const uint16_t* slot_res, uint32_t S, uint8_t* single);
void launch_join_info(hipStream_t st, const uint32_t* soff, const uint8_t* single, const uint32_t* opos, uint32_t i_elo, uint32_t i_a, uint32_t i_b, uint32_t i_ehi,
                      uint32_t skip, uint32_t* out);
void launch_dp(hipStream_t st, const uint32_t* heads, uint32_t* counters, uint32_t cnt0, uint32_t n_shards,
               uint32_t heads_region, uint32_t* pool, const uint32_t* slot_rec, uint16_t* slot_res, int K, long long Rfix,
               double min_ratio, uint32_t grid, bool fp = false, double rate = 0.0);   // fp: general-rate path (doubles, whole-contig runs)
void launch_fixfirst(hipStream_t st, const uint32_t* ctg_off, uint32_t nc, const uint32_t* soff, const uint8_t* slot_info,
                     uint16_t* slot_res);
void launch_emit(hipStream_t st, const uint16_t* slot_res, const uint8_t* slot_info, const uint32_t* opos, uint32_t S,
                 uint32_t mask, uint8_t* out);
void launch_contig_bounds(hipStream_t st, const uint32_t* ctg_off, uint32_t nc, const uint32_t* soff, const uint32_t* opos,
                          uint32_t* out_bounds);

}  // namespace np1k
