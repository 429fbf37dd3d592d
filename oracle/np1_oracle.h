/* oracle/np1_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's short-read polishing hot path
 * (nextpolish1.so: score_chain and kmer_count), operating on an already decoded
 * record stream instead of htslib iterators.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library -- as the checker, never
 * as the thing measured or shipped.
 *
 * Parity status: PINNED.  The restatement is checked record-for-record against the
 * real reference compiled from /root/reference by oracle/Makefile (`make ref` ->
 * oracle/_ref/nextpolish1) on fuzzed and synthetic BAM+FASTA inputs
 * (tests/test_oracle_vs_ref.py) and against golden vectors generated from that
 * binary (tests/golden/, generator: tests/golden/make_golden.py).
 */
#ifndef NP1_ORACLE_H
#define NP1_ORACLE_H
#include <stdint.h>

/* Same field order / natural alignment as the reference `Configure`
 * (reference: source/lib/config.h:25-67). */
typedef struct {
    uint8_t trim_len_edge, ext_len_edge, min_map_quality;
    double indel_balance_factor_sgs, min_count_ratio_skip;
    uint8_t min_len_ldr, min_len_inter_kmer, max_len_kmer, max_count_kmer;
    uint8_t min_depth_snp, min_count_snp;
    int8_t min_count_snp_link;
    double ploidy, indel_balance_factor_lgs, max_indel_factor_lgs, max_snp_factor_lgs, min_snp_factor_sgs;
    int32_t region_count;
    uint32_t count_read_ins_sgs, max_ins_len_sgs;
    int32_t max_ins_fold_sgs, max_variant_count_lgs;
    double max_clip_ratio_sgs, max_clip_ratio_lgs;
    int32_t trace_polish_open, read_tlen, read_len;
    char *fastafn, *bamfn, *thirdbamfn;
} np1o_configure;

/* BAI of one reference sequence, as htslib holds it after hts_idx_load: bins in ascending order with their chunks and their `loff`
 * (hts.c update_loff), 5 levels, 16 kb windows. */
typedef struct np1o_index {
    int32_t n_bins;
    const uint32_t* bin;
    const uint64_t* loff;
    const uint32_t* chunk_first;   /* n_bins + 1 */
    const uint64_t* chunk_u;
    const uint64_t* chunk_v;
} np1o_index;

/* One contig + its records in BAM file order (a slice of a decoded stream). */
typedef struct {
    const char* draft;      /* raw FASTA characters, case preserved */
    int32_t length;
    int64_t n_reads;
    const int32_t* pos;
    const uint16_t* flag;
    const uint32_t* n_cigar;
    const int32_t* l_qseq;
    const uint8_t* mapq;
    const int32_t* isize;
    const uint64_t* cigar_off;   /* absolute index into cigar[] */
    const uint64_t* seq_off;     /* absolute byte offset into seq[] */
    const uint64_t* qual_off;    /* absolute byte offset into qual[] (kmer_count only) */
    const uint32_t* cigar;
    const uint8_t* seq;
    const uint8_t* qual;
    int32_t has_next;            /* 1: arrays hold one more record (index n_reads) = the next record in BAM
                                    file order after this contig's */
    /* optional, only for streams read from a BAM + BAI pair: what the reference's region iterator (contig.c:982-1043 over
     * htslib's hts_itr_query / hts_itr_next) sees -- per record the BGZF virtual offset of its first byte and of the byte behind
     * it, and the index of this reference sequence.  NULL: kmer_count / snp_valid take "records in file order" instead. */
    const uint64_t* voff;
    const uint64_t* voff_end;
    const struct np1o_index* idx;
} np1o_contig;

/* Fills *cfg with the defaults of config_init (reference: source/lib/config.c:8-38). */
void np1o_default_config(np1o_configure* cfg);

/* score_chain / kmer_count for one contig.  Returns a malloc'd NUL-terminated polished
 * string (caller frees with np1o_free) and its length in *out_len. */
char* np1o_score_chain(const np1o_contig* c, const np1o_configure* cfg, int32_t* out_len);
char* np1o_kmer_count(const np1o_contig* c, const np1o_configure* cfg, int32_t* out_len);
char* np1o_snp_valid(const np1o_contig* c, const np1o_configure* cfg, int32_t* out_len);   /* task 4, source/lib/snpvalid.c */
/* task 3, source/lib/snpphase.c: `sr` = the short-read records of the contig, `lr` = its long-read records (both with
 * qualities).  NULL with *out_len = -1 where the reference's own result rests on a null / uninitialised read. */
char* np1o_snp_phase(const np1o_contig* sr, const np1o_contig* lr, const np1o_configure* cfg, int32_t* out_len);
void np1o_snp_phase_stats(int64_t out[10]);   /* what the stages of the last np1o_snp_phase call did (np1_oracle.c: g_sp_stats) */
/* the change list of the calling thread's last task call made with cfg->trace_polish_open set (reference: source/lib/contig.c:743-797):
 * 4 words per point (pos, index, curbase, base); returns the number of points; the array stays the oracle's */
int32_t np1o_last_points(const int32_t** out);
void np1o_free(void* p);

/* Algorithmic update count of score_chain's pileup (one per slot vote), for throughput reports. */
int64_t np1o_last_update_count(void);

#endif
