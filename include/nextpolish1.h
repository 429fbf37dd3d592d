/* include/nextpolish1.h -- C ABI of the MI355X-native short-read polishing core.
 *
 * Part 1 is the DROP-IN surface: the exact symbols, struct layouts and ownership rules
 * the reference's ctypes caller binds in nextpolish1.so
 *   (reference: source/lib/nextpolish1.py:27-100  -- ctypes structs and prototypes,
 *               source/lib/config.h:25-70, source/lib/contig.h:10-25,
 *               source/lib/scorechain.h, source/lib/kmercount.h).
 * Part 2 is this library's own batch interface (np1_ prefix): decoded record streams
 * resident in HBM, one launch sequence for many contigs, used by the CLI, bench.py and
 * the multi-GPU driver.  Plain pointers and sizes only; no torch / HIP types.
 *
 * All compute entry points run on the GPU and fail loudly (message on stderr, exit(1)
 * for Part 1 like the reference; negative return + np1_last_error() for Part 2) when no
 * HIP device is usable.  There is no CPU fallback in this library.
 */
#ifndef NEXTPOLISH1_H
#define NEXTPOLISH1_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ Part 1: drop-in */

/* reference: source/lib/config.h:25-67 (field order and natural alignment are ABI;
 * the Python caller mutates the struct in place after config_init, nextpolish1.py:102-133) */
typedef struct {
    uint8_t trim_len_edge;
    uint8_t ext_len_edge;
    uint8_t min_map_quality;
    double indel_balance_factor_sgs;
    double min_count_ratio_skip;
    uint8_t min_len_ldr;
    uint8_t min_len_inter_kmer;
    uint8_t max_len_kmer;
    uint8_t max_count_kmer;
    uint8_t min_depth_snp;
    uint8_t min_count_snp;
    int8_t min_count_snp_link;
    double ploidy;
    double indel_balance_factor_lgs;
    double max_indel_factor_lgs;
    double max_snp_factor_lgs;
    double min_snp_factor_sgs;
    int32_t region_count;
    uint32_t count_read_ins_sgs;
    uint32_t max_ins_len_sgs;
    int32_t max_ins_fold_sgs;
    int32_t max_variant_count_lgs;
    double max_clip_ratio_sgs;
    double max_clip_ratio_lgs;
    int32_t trace_polish_open;
    int32_t read_tlen;
    int32_t read_len;
    char* fastafn;
    char* bamfn;
    char* thirdbamfn;
} Configure;

/* reference: source/lib/contig.h:10-22 */
typedef struct {
    int32_t pos;
    int16_t index;
    char curbase;
    char base;
} PolishPoint;

typedef struct {
    char* contig;          /* calloc'd, NUL terminated; caller copies then calls polishresult_destory */
    PolishPoint* data;     /* non-NULL only when trace_polish_open */
    int32_t length;
    int32_t datalength;
} PolishResult;

/* reference: source/lib/config.c:8-56 (defaults, insert-size probe of the first 10000 records,
 * bamfn/thirdbamfn become NULL when the file is not accessible) */
Configure* config_init(const char* fastafn, const char* bamfn, const char* thirdbamfn);
/* reference: source/lib/config.c:58-68 (sic: "destory") */
void config_destory(Configure* config);

/* reference: source/lib/scorechain.c:3-15 -- polishes one contig with the score-chain pass */
PolishResult* score_chain(const char* tigname, Configure* configure);
/* reference: source/lib/kmercount.c:93-126 -- re-votes lowercase regions by spanning-read haplotypes */
PolishResult* kmer_count(const char* tigname, Configure* configure);
/* reference: source/lib/snpphase.c:87-134 (task 3): heterozygous sites from the short reads (configure->bamfn) and the long
 * reads (configure->thirdbamfn) of one contig, low-depth correction with both, read-backed phasing; result with the
 * long-read-only evidence in lower case.  source/lib/snpvalid.c:3-36 (task 4): two vote rounds over the lowercase regions.
 * source/lib/lgspolish.c (task 5): resolved by the ctypes caller at import time (nextpolish1.py:95-100) but refused by the
 * caller itself (nextpolish1.py:338-340); here it reports that and exit(1). */
PolishResult* snp_phase(const char* tigname, Configure* configure);
PolishResult* snp_valid(const char* tigname, Configure* configure);
PolishResult* lgspolish(const char* tigname, Configure* configure);
/* reference: source/lib/contig.c:25-30 */
void polishresult_destory(PolishResult* polishresult);

/* ------------------------------------------------------------------ Part 2: batch interface */

const char* np1_last_error(void);

/* Host-side decoded record stream = whole contigs + their BAM records (see DESIGN.md "data layout"). */
typedef struct np1_stream np1_stream;

typedef struct {
    int64_t n_contigs, n_reads;
    const int32_t* ctg_len;       /* [n_contigs] */
    const uint32_t* ctg_off;      /* [n_contigs+1] offsets into draft */
    const uint64_t* read_begin;   /* [n_contigs+1] */
    const char* draft;            /* concatenated FASTA characters */
    int64_t draft_len;
    const int32_t* pos;
    const uint32_t* ctg;
    const uint16_t* flag;
    const uint32_t* n_cigar;
    const int32_t* l_qseq;
    const uint64_t* cigar_off;
    const uint64_t* seq_off;
    const uint8_t* mapq;
    const int32_t* isize;
    const uint64_t* qual_off;
    const uint32_t* cigar;
    int64_t cigar_len;
    const uint8_t* seq;
    int64_t seq_len;
    const uint8_t* qual;
    int64_t qual_len;             /* 0 when qualities were not loaded */
} np1_stream_view;

/* names == NULL / n_names == 0: every contig of the FASTA index, in index order. */
np1_stream* np1_stream_load(const char* fasta, const char* bam, const char* const* names, int n_names, int with_qual);
/* Deep copy of caller-owned arrays into a new stream (tests, adapters from other decoders). */
np1_stream* np1_stream_build(const np1_stream_view* v, const char* const* contig_names);
void np1_stream_get_view(const np1_stream* s, np1_stream_view* out);
const char* np1_stream_contig_name(const np1_stream* s, int64_t i);
/* BGZF virtual offsets of the records of a stream loaded from a file: start of each record and the byte behind it (htslib's
 * bgzf_tell convention).  Returns the record count, or 0 for a stream built in memory. */
int64_t np1_stream_voffs(const np1_stream* s, const uint64_t** beg, const uint64_t** end);
uint64_t np1_stream_algorithmic_bytes(const np1_stream* s, int with_qual);
int np1_stream_write_files(const np1_stream* s, const char* fasta, const char* bam, int bgzf_level);
/* same, with raw BAM optional fields per record (aux_pool[aux_off[i] .. aux_off[i+1])): test data with SA tags */
int np1_stream_write_files_aux(const np1_stream* st, const char* fasta, const char* bam, int level, const uint8_t* aux_pool,
                               const uint64_t* aux_off);
/* several streams as ONE FASTA(+.fai) + ONE coordinate-sorted BAM(+.bai): contigs of streams[0], then of streams[1], ... */
int np1_streams_write_files(np1_stream* const* streams, int n, const char* fasta, const char* bam, int bgzf_level);
/* the same with a quality model for streams that carry no qualities: 0 = none (0xff bytes), 1 = Illumina-like binned qualities
 * (2 / 12 / 23 / 37, np_synth.cpp:synth_binned_qualities): what a BAM of a current instrument looks like to the BGZF inflate;
 * 2 = uniformly random in [25, 40] (incompressible: the decoder's worst case) */
int np1_streams_write_files_q(np1_stream* const* streams, int n, const char* fasta, const char* bam, int bgzf_level, int qual_model);
void np1_stream_free(np1_stream* s);

/* Synthetic workload.  Field meanings: nextpolish_amd/csrc/np_synth.h */
typedef struct {
    uint64_t seed;
    int32_t n_contigs;
    const int32_t* contig_len;
    double depth;
    int32_t read_len;
    double frag_mean, frag_sd;
    double draft_sub, draft_indel;
    double draft_lower;
    double read_sub, read_indel;
    double softclip_rate;
    double dup_rate, supp_rate, sec_rate, unmapped_rate;
    double lowmapq_rate;
    double weird_rate;
    int32_t with_qual;
} np1_synth_params;
void np1_synth_defaults(np1_synth_params* p);
np1_stream* np1_stream_synth(const np1_synth_params* p, const char* contig_name_prefix);
/* Long-read workload (nextpolish2 path): random drafts + noisy reads of log-normal length with known CIGARs */
typedef struct {
    uint64_t seed;
    int32_t n_contigs;
    const int32_t* contig_len;
    double depth, mean_len;
    double sub, ins, dele;
    int32_t max_indel;
    double clip_rate;
} np1_synth_long_params;
np1_stream* np1_stream_synth_long(const np1_synth_long_params* p, const char* contig_name_prefix);
/* Either workload over contigs that are handed in (names, sequences, lengths; p = np1_synth_params for long_reads = 0, np1_synth_long_params for 1;
 * their n_contigs / contig_len are not used): the reads of the next step of a multi-step run, aligned to the assembly the step before wrote. */
np1_stream* np1_stream_synth_on(const void* p, int long_reads, const char* const* names, const char* const* seqs, const int64_t* lens, int n);
/* Diploid workload of task 3 (snp_phase): per contig one draft with its own errors, short read pairs and long reads drawn from two
 * haplotypes that differ by substitutions (het_sub per base) and small indels (het_indel); sr_holes = stretches per contig no short
 * fragment touches.  Both streams carry qualities.  Returns 0 and the two streams (same contigs, same drafts). */
typedef struct {
    uint64_t seed;
    int32_t n_contigs;
    const int32_t* contig_len;
    double sr_depth, lr_depth;
    int32_t read_len;
    double frag_mean, lr_len;
    double het_sub, het_indel, draft_err, sr_err, lr_err;
    int32_t sr_holes;
} np1_diploid_params;
int np1_stream_synth_diploid(const np1_diploid_params* p, const char* contig_name_prefix, np1_stream** sr, np1_stream** lr);
/* Test hook: the BGZF block decoder (own raw-DEFLATE implementation) on one stream; 1 = accepted and dst filled. */
int np1_debug_inflate(const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_len);
/* test hook: CRC-32 of a BGZF block as the reader / writer compute it (must equal zlib's crc32) */
uint32_t np1_debug_crc32(const uint8_t* src, uint64_t len);

/* Device context: one per process per GPU; created lazily AFTER any fork (the reference's callers
 * fork worker pools after config_init, nextpolish1.py:219-223). */
typedef struct np1_ctx np1_ctx;
int np1_device_count(void);
np1_ctx* np1_ctx_create(int device);
void np1_ctx_destroy(np1_ctx* ctx);

/* A batch resident in HBM. */
typedef struct np1_batch np1_batch;
np1_batch* np1_batch_upload(np1_ctx* ctx, const np1_stream* s);
void np1_batch_free(np1_batch* b);
/* Streaming use: an empty batch object bound to a context, (re)filled from successive host streams.  Its HBM buffers only
 * grow, so nothing is allocated in steady state; np1_batch_reload leaves the H2D copies in flight on the context's stream.
 * Round 5: the bytes of an UNPINNED stream are taken while the call runs (through the library's own pinned ring: the stream may be freed
 * right after); np1_stream_pin copies the arrays an upload moves into one page-locked arena of the stream, from which the copies run
 * asynchronously at full PCIe rate (the stream must then outlive the pass; np1_stream_free waits for the device).  The GPU never reads
 * the caller's pageable memory in place (csrc/np_hostcopy.h).  np1_batch_results_fetch moves the polished strings of the whole batch into a pinned host buffer with one D2H
 * copy on the same stream and waits for it; np1_batch_results_ptr()[bounds[c] .. bounds[c+1]) is contig c. */
np1_batch* np1_batch_create(np1_ctx* ctx);
int np1_batch_reload(np1_batch* b, const np1_stream* s);
int np1_stream_pin(np1_stream* s);
/* bytes one np1_batch_reload of this stream moves over PCIe (bases as 2 bits + exceptions, operation counts as 16 bits, offsets rebuilt on the device) */
uint64_t np1_stream_upload_bytes(np1_stream* s);
int np1_batch_results_fetch(np1_batch* b);
const char* np1_batch_results_ptr(np1_batch* b);
const uint32_t* np1_batch_results_bounds(np1_batch* b);

/* Streamed polishing (SURVEY.md 8d timing scope 1): batches flow  pinned host arrays -> H2D -> score_chain kernels -> D2H
 * on `lanes` device lanes (one HIP stream + one reusable HBM batch + one host thread each), so the copies of one batch
 * overlap the kernels of another.  np1_pipe_run polishes the given streams (batch k = streams[k]) and keeps the polished
 * strings inside the pipe until the next run: np1_pipe_result(p, k, c, &len) is contig c of batch k.
 * task: 1 = score_chain, 2 = kmer_count.  Returns 0 on success (np1_last_error otherwise). */
typedef struct np1_pipe np1_pipe;
np1_pipe* np1_pipe_open(int device, int lanes);
int np1_pipe_run(np1_pipe* p, np1_stream* const* streams, int n, const Configure* cfg, int task);
const char* np1_pipe_result(np1_pipe* p, int batch, int64_t contig, int64_t* len);
/* Resident mode (kernel-path measurements): np1_pipe_upload keeps one HBM batch per stream (dealt round-robin over the
 * lanes, replacing any earlier set); np1_pipe_run_resident polishes every resident batch `passes` times, the lanes
 * working concurrently, outputs staying on the device; np1_pipe_resident_batch(p, k) is batch k for np1_batch_* calls.
 * A resident batch holds its INPUTS in HBM; the buffers a pass works in (slot arrays, descriptors, DP records, output: ~2.6 x the
 * inputs) belong to the lane and are lent to the batch for the pass, so a draft of any size stays resident next to `lanes` work
 * sets.  np1_pipe_run_resident_timed: one instrumented pass of batch k on lane 0 (per-stage HIP-event milliseconds into
 * stage_ms[np1_stage_count()]); its result lengths stay readable through np1_batch_result_len. */
int np1_pipe_upload(np1_pipe* p, np1_stream* const* streams, int n);
int np1_pipe_run_resident(np1_pipe* p, const Configure* cfg, int task, int passes);
int np1_pipe_run_resident_timed(np1_pipe* p, int k, const Configure* cfg, float* stage_ms);
np1_batch* np1_pipe_resident_batch(np1_pipe* p, int k);
void np1_pipe_close(np1_pipe* p);
/* BGZF blocks the device-side ingest of np1_pipe_run_files / _phase_files handed back to the host decoder since the pipe was opened (0 on well-formed files) */
uint64_t np1_pipe_host_inflated_blocks(np1_pipe* p);
/* device-side ingest since the pipe was opened (or the last call with reset != 0): out = {BGZF block decoder ms, CRC pass ms, compressed bytes in,
 * inflated bytes out, launches}, HIP-event times on the lanes' streams summed over the lanes */
void np1_pipe_ingest_stats(np1_pipe* p, double out[5], int reset);
/* From files: contigs of the FASTA index (all when names == NULL) are packed in index order into batches of at most
 * batch_bp draft bases; host threads load batch k+1.. (BGZF inflate + record split) while the lanes polish batch k.
 * Every finished contig is handed to `sink` in index order (user, name, sequence, length).  Returns 0 on success. */
typedef void (*np1_sink_fn)(void* user, const char* name, const char* seq, int64_t len);
int np1_pipe_run_files(np1_pipe* p, const char* fasta, const char* bam, const char* const* names, int n_names, int64_t batch_bp,
                       const Configure* cfg, int task, np1_sink_fn sink, void* user);
/* Task 3 (snp_phase) from files: per batch the short-read BAM goes through the device-side ingest (compressed blocks -> HBM ->
 * inflate + record split on the GPU), the long-read BAM through the host loader, then one np1_batch_snp_phase pass on lane 0;
 * a loader thread stages the next batch meanwhile.  Contigs reach `sink` in request order. */
int np1_pipe_run_phase_files(np1_pipe* p, const char* fasta, const char* bam_sr, const char* bam_lr, const char* const* names, int n_names,
                             int64_t batch_bp, const Configure* cfg, np1_sink_fn sink, void* user);

/* Names of the timed stages of one score_chain pass, in launch order (for profiles / roofline). */
#define NP1_MAX_STAGES 16
int np1_stage_count(void);
const char* np1_stage_name(int i);

/* Runs score_chain over every contig of the batch on the context's stream; outputs stay on the device.
 * stage_ms: NULL, or float[NP1_MAX_STAGES] receiving per-stage HIP-event milliseconds (forces a sync).
 * Returns 0 on success. */
int np1_batch_score_chain(np1_batch* b, const Configure* cfg, float* stage_ms);
/* kmer_count (task 2) over the same batch; the stream must have been loaded with base qualities
 * (np1_stream_load(..., with_qual = 1)) and cfg->read_tlen must be set (config_init does).  Returns 0 on success;
 * results are fetched with np1_batch_result_len / np1_batch_result_copy like for score_chain. */
int np1_batch_kmer_count(np1_batch* b, const Configure* cfg, float* stage_ms);
/* Makes np1_batch_kmer_count of this batch replay the reference's region iterator (reference: source/lib/contig.c:982-1043 over htslib's
 * hts_itr_query / hts_itr_next): per part the records the first loop of ss_kmer_correct gets, the record left in its buffer and the
 * passes of the second loop come from the BAM index and the records' virtual offsets instead of "records in file order" (DESIGN.md
 * section 3).  `s` = the stream the batch was uploaded from, read from `bam` (np1_stream_load); it has to stay alive until the pass is
 * done.  Experimental in round 2: equal to the compiled reference through the host model, not yet validated on a GPU. */
int np1_batch_enable_replay(np1_batch* b, const np1_stream* s, const char* bam);
/* task 4 on an uploaded batch (reference: source/lib/snpvalid.c:3-36 snp_valid; needs a stream loaded with qualities) */
int np1_batch_snp_valid(np1_batch* b, const Configure* cfg, float* stage_ms);
/* task 3 on two uploaded batches of the same contigs (reference: source/lib/snpphase.c:87-134 snp_phase): `sr` = the short-read
 * records (the reference's -b BAM), `lr` = the long-read records (its third BAM), both loaded with qualities and uploaded on the
 * same context.  cfg->read_len / read_tlen as config_init sets them.  The result lands in `sr` (np1_batch_result_*).  Inputs for
 * which the reference itself reads through a null or unset pointer fail with an error message instead of a result. */
int np1_batch_snp_phase(np1_batch* sr, np1_batch* lr, const Configure* cfg);
/* Intra-contig tiling (np1_tile.cpp): one contig of any length polished as independent tiles of tile_bp draft bases with
 * a halo of halo_bp on each side, each tile reading its own region of the BAM through the index; the join is exact (a tile is redone
 * with a doubled halo when a halo holds no slot the chain restarts behind).  The reference takes contigs up to 2^31 bases
 * (source/nextPolish:101-102) in one score_chain call (source/lib/scorechain.c:3-15); this is that call for contigs beyond one HBM batch.
 * first_tile / tile_stride: this call polishes tiles first_tile, first_tile + tile_stride, ... and returns their pieces concatenated in
 * tile order (0, 1: all of them = the polished contig).  Tiles are independent, so the tiles of one dominant contig can be dealt over
 * ranks: a call with first_tile = k and a stride of at least the number of tiles returns the piece of tile k alone, and the pieces of all
 * tiles in tile order are the polished contig (the callers of this repository still deal whole contigs over ranks: DESIGN.md section 11).
 * *out: malloc'd, NUL-terminated (np1_free_string).
 * stats (optional, 4 words): tiles, tiles recomputed with a wider halo, records read, records of the largest tile.
 * Pieces: np1_batch_keep_single (before the run) makes a one-contig batch remember which slots left the vote with one state;
 * np1_batch_tile_join gives {left halo has such a slot, right halo has one, output offset of the tile's first own base, of the first base
 * behind it}; np1_batch_result_range copies polished characters [o0, o1) out. */
int np1_score_chain_tiled(np1_ctx* ctx, const char* fasta, const char* bam, const char* name, const Configure* cfg, int64_t tile_bp, int64_t halo_bp,
                          int64_t first_tile, int64_t tile_stride, char** out, int64_t* out_len, uint64_t* stats);
void np1_free_string(char* s);
/* The same with the contig opened ONCE: np1_tiler_open reads the FASTA index entry, the contig's draft and the BAM index; np1_tiler_run
 * polishes tiles first_tile, first_tile + tile_stride, ... (NP1_TILE_PREFETCH=1: the records of tile t + 1 are read while the device runs tile t) and returns their
 * pieces joined in tile order, piece_len[i] (optional; room for every tile of the call) = the length of the i-th piece -- what a rank that
 * takes every world-th tile of a dominant contig calls once (nextpolish_amd/nextpolish1.py: write_tile_pieces). */
typedef struct np1_tiler np1_tiler;
np1_tiler* np1_tiler_open(const char* fasta, const char* bam, const char* name);
int64_t np1_tiler_length(const np1_tiler* t);
int np1_tiler_run(np1_tiler* t, np1_ctx* ctx, const Configure* cfg, int64_t tile_bp, int64_t halo_bp, int64_t first_tile, int64_t tile_stride,
                  char** out, int64_t* out_len, int64_t* piece_len, uint64_t* stats);
void np1_tiler_close(np1_tiler* t);
/* score_chain over a whole FASTA index with tiling on: contigs longer than tile_bp tile by tile, the others through the pipe in batches;
 * every contig reaches `sink` in index order (what `nextpolish1 scorechain` does when NP1_TILE_BP is set) */
int np1_run_files_tiled(np1_pipe* pipe, int device, const char* fasta, const char* bam, int64_t batch_bp, int64_t tile_bp, int64_t halo_bp,
                        const Configure* cfg, np1_sink_fn sink, void* user);
int np1_batch_keep_single(np1_batch* b, int on);
int np1_batch_tile_join(np1_batch* b, uint32_t i_elo, uint32_t i_a, uint32_t i_b, uint32_t i_ehi, uint32_t skip, uint32_t out[4]);
int np1_batch_result_range(np1_batch* b, uint32_t o0, uint32_t o1, char* dst);
/* Blocks until the batch's work is complete. */
int np1_batch_sync(np1_batch* b);
/* Polished length of contig i (valid after a completed run), and copy-out of its NUL-terminated string. */
int64_t np1_batch_result_len(np1_batch* b, int64_t contig);
int np1_batch_result_copy(np1_batch* b, int64_t contig, char* dst, int64_t cap);
/* total slot votes ("updates") of the last run and HBM bytes held by the batch */
int64_t np1_batch_update_count(np1_batch* b);
int64_t np1_batch_device_bytes(np1_batch* b);
/* device work counters of the last score_chain run (diagnostics; layout = np1_core.h CNT_*), up to n words */
int np1_batch_debug_counters(np1_batch* b, uint32_t* out, int n);
/* raw per-slot arrays of the last run (diagnostics): kind 0 = slot_info u8, 1 = slot_res u16, 2 = slot_rec u32;
 * copies min(n, slots) elements and returns the slot count */
int64_t np1_batch_debug_slots(np1_batch* b, int kind, void* out, int64_t n);

/* Device memory of both libraries comes from a caching allocator in front of hipMalloc / hipFree (csrc/np_devalloc.h; NP_DEVCACHE_MB, NP_PINCACHE_MB):
 * np1_alloc_stats: {hits, misses, runtime frees, blocks fenced at release, idle bytes, bytes handed out, peak idle bytes, bound} of the device cache;
 * np1_alloc_trim: every idle block back to the runtime; np1_diag_report: the library's streams with their state and the cache counters, to `fd`. */
void np1_alloc_stats(uint64_t out[8]);
void np1_alloc_trim(void);
void np1_diag_report(int fd);

/* calgs (reference: source/lib/calgs.c:8-24): sum of sequence lengths of a FASTA/FASTQ (gz aware) */
uint64_t calgs(const char* file);

#ifdef __cplusplus
}
#endif
#endif
