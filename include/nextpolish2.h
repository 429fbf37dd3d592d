/* nextpolish2.h -- C ABI of the MI355X long-read consensus library (drop-in for the reference's nextpolish2.so).
 *
 * The reference caller binds exactly these symbols with ctypes (reference: source/lib/nextpolish2.py:18-65) and
 * indexes two of the structs directly (`REFS.contents.ref[i].n`, `c_seq.contents.data[i].seq/.len`,
 * nextpolish2.py:92-96,139-151), so `refs_`, `ref_`, `consensus_trimed` and `consensus_trimed_data` are ABI and keep
 * the reference layouts (source/lib/ctg_cns.h:92-101,165-184).  `ctg_cns_cfg` is opaque to the caller
 * (nextpolish2.py:18-19 declares an empty Structure); its contents are private to this library.
 *
 * Error convention = the reference's: fatal conditions print to stderr and exit(1) (ctg_cns.c:2272-2275,3534-3537).
 * No HIP context is created by read_ref / ctg_cns_init (the caller forks its worker pool after them,
 * nextpolish2.py:184-194); the device is initialised lazily by the first ctg_cns_core of a process.
 */
#ifndef NEXTPOLISH2_AMD_H
#define NEXTPOLISH2_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-contig assembly QV track parsed from the FASTA comment (ctg_cns.h:157-162; set_ref_qv, ctg_cns.c:2233-2267) */
typedef struct ref_qv {
    uint32_t ide : 12;
    uint32_t ort : 10;
    uint32_t irt : 10;
    uint32_t p;
} ref_qv;

/* one accepted contig (ctg_cns.h:164-170): name, 2-bit packed bases (16 per word, first base in the top bits,
 * bseq.c:87-103), optional QV track, length */
typedef struct {
    char* n;
    uint32_t* s;
    ref_qv* qv;
    uint32_t qv_l;
    uint32_t length;
} ref_;

typedef struct {   /* ctg_cns.h:172-176 */
    ref_* ref;
    uint32_t i;    /* number of contigs held */
    uint32_t i_m;  /* capacity */
} refs_;

typedef struct {   /* ctg_cns.h:92-96 */
    unsigned int len;
    float identity;   /* never set by the reference either (calloc'd) */
    char* seq;
} consensus_trimed;

typedef struct {   /* ctg_cns.h:98-101 */
    consensus_trimed* data;
    int i_m;       /* number of pieces (>1 only when the contig was split) */
} consensus_trimed_data;

typedef struct ctg_cns_cfg ctg_cns_cfg;   /* opaque (reference: ctg_cns.c:3337-3353) */

/* reference: read_ref, ctg_cns.c:2269-2295.  Loads the contigs named in accept_names (all when n == 0) from a
 * (gzipped) FASTA/FASTQ in file order; sorts accept_names in place like the reference (qsort + strcmp).
 * Non-ACGT letters pack as the reference packs them (value 4 OR-ed into the 2-bit stream, bseq.c:91). */
refs_* read_ref(char* fasta, char** accept_names, int n);
void refs_destroy(refs_* refs);   /* ctg_cns.c:2199-2208 */

/* reference: bseq.c:87-103 and :105-124 */
void seq2bit1(uint32_t* s, uint32_t len, char* seq);
void bit2seq1(uint32_t* s, uint32_t len, char* seq);

/* reference: ctg_cns_init, ctg_cns.c:3355-3381.  window: consensus window in bp (0 -> 40 M; otherwise must exceed
 * 4 x the 1 Mb window overlap); read_type 1 = ONT, 2 = PacBio CLR, 3 = HiFi; split: 0/1/2 as in the reference;
 * ide/ort/irt: thresholds of the reference-QV based split logic. */
ctg_cns_cfg* ctg_cns_init(int window, int read_type, int split, float ide, float ort, float irt);
void ctg_cns_destroy(ctg_cns_cfg* cfg);   /* ctg_cns.c:3383-3397 */

/* reference: ctg_cns_core, ctg_cns.c:3399-3623.  Consensus of one contig from the sorted, indexed BAMs listed (one
 * path per line) in bam_list.  Returns calloc'd pieces; the caller copies and calls free_consensus_trimed_data. */
consensus_trimed_data* ctg_cns_core(ctg_cns_cfg* cfg, ref_* ref, char* bam_list);
void free_consensus_trimed_data(consensus_trimed_data* d);   /* ctg_cns.c:2150-2156 */

/* ---- additions of this library (not in the reference ABI) ---- */
/* last error text of this thread for the np2_* calls below ("" if none) */
const char* np2_last_error(void);
/* device index the process will use / uses (pid mod device count, or NP2_DEVICE) */
int np2_device_index(void);
/* diagnostics: this library's streams with their state and the allocator caches' counters, written to `fd` */
void np2_diag_report(int fd);

#ifdef __cplusplus
}
#endif
#endif
