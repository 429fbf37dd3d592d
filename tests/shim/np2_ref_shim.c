/* LD_PRELOAD shim around the exported helpers of the compiled reference library (oracle/_ref/nextpolish2.so):
 * TEST INFRASTRUCTURE.  The reference's split-read structural layer is only observable end to end; its helpers are
 * non-static, so inside the shared object they are called through the PLT and can be interposed.  The shim forwards
 * every call to the real function and logs the stage results to the file named by NP2_SHIM_LOG, which lets the
 * tests compare this repository's implementation stage by stage.  Struct layouts restated from the reference
 * header source/lib/ctg_cns.h:186-248 (gap_, gaps, gap_cluster, gap_clusters, sup_alns, ld_regs, pos). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct { uint32_t s, e; } pos;
typedef struct { pos gap; uint32_t p_id, s_id; uint32_t p_s, s_s; uint32_t dl_m, l; uint8_t* dseq; } gap_;
typedef struct { uint32_t i, i_m; gap_* gap; } gaps;
typedef struct { pos r; uint32_t median; uint32_t i_m; gap_* gap[120]; } gap_cluster;
typedef struct { uint32_t i, i_m; gap_cluster* clusters; } gap_clusters;
typedef struct { uint32_t i, i_m; pos* reg; } ld_regs;

/* the reference library is dlopen'ed RTLD_LOCAL by ctypes, so RTLD_NEXT does not see it: resolve through its handle */
static void* real_sym(const char* name) {
    static void* h = NULL;
    if (!h) {
        const char* p = getenv("NP2_REF_SO");
        h = p ? dlopen(p, RTLD_LAZY | RTLD_NOLOAD) : NULL;
        if (!h) { fprintf(stderr, "np2_ref_shim: set NP2_REF_SO to the loaded reference library\n"); abort(); }
    }
    void* s = dlsym(h, name);
    if (!s) { fprintf(stderr, "np2_ref_shim: %s not found\n", name); abort(); }
    return s;
}

static FILE* logf_(void) {
    static FILE* f = NULL;
    if (!f) {
        const char* p = getenv("NP2_SHIM_LOG");
        f = p ? fopen(p, "a") : stderr;
        if (!f) f = stderr;
    }
    return f;
}

int update_gap_cluster(gaps* gs, gap_clusters* clusters, uint16_t* ref_ds, const int w, const int d, const int32_t ref_s) {
    static int (*real)(gaps*, gap_clusters*, uint16_t*, int, int, int32_t) = NULL;
    if (!real) real = real_sym("update_gap_cluster");
    const uint32_t n_gaps = gs->i;
    const int t = real(gs, clusters, ref_ds, w, d, ref_s);
    FILE* f = logf_();
    fprintf(f, "update_gap_cluster gaps %u w %d d %d ref_s %d -> clusters %u total %d\n", n_gaps, w, d, ref_s, d < 10 ? 0 : clusters->i, t);
    if (d >= 10)
        for (uint32_t i = 0; i < clusters->i; ++i) fprintf(f, "  cluster %u i_m %u median %u\n", i, clusters->clusters[i].i_m, clusters->clusters[i].median);
    fflush(f);
    return t;
}

void generate_gapseqs(gap_clusters* clusters, void* tags_list, const int32_t s_) {
    static void (*real)(gap_clusters*, void*, int32_t) = NULL;
    if (!real) real = real_sym("generate_gapseqs");
    real(clusters, tags_list, s_);
    FILE* f = logf_();
    for (uint32_t i = 0; i < clusters->i; ++i) {
        const gap_cluster* c = &clusters->clusters[i];
        uint32_t l2 = 0;
        for (uint32_t j = 0; j < c->i_m; ++j) l2 += c->gap[j]->l == 2;
        fprintf(f, "generate_gapseqs cluster %u r %u %u i_m %u usable %u\n", i, c->r.s, c->r.e, c->i_m, l2);
        for (uint32_t j = 0; j < c->i_m; ++j)
            fprintf(f, "    gap %u l %u read %u..%u p_id %u s_id %u\n", j, c->gap[j]->l, c->gap[j]->gap.s, c->gap[j]->gap.e, c->gap[j]->p_id, c->gap[j]->s_id);
    }
    fflush(f);
}

uint32_t update_align_tags(gap_clusters* clusters, void* sup_alns, void* tags_list, uint32_t seq_count, char* rfseq, void* aln, const int32_t ref_s,
                           const int32_t ref_e, void* msa) {
    static uint32_t (*real)(gap_clusters*, void*, void*, uint32_t, char*, void*, int32_t, int32_t, void*) = NULL;
    if (!real) real = real_sym("update_align_tags");
    const uint32_t r = real(clusters, sup_alns, tags_list, seq_count, rfseq, aln, ref_s, ref_e, msa);
    fprintf(logf_(), "update_align_tags streams %u -> %u\n", seq_count, r);
    fflush(logf_());
    return r;
}

void update_ld_regs(ld_regs* regs, const uint16_t* r, const int32_t l, const int w, const int d, const int32_t s) {
    static void (*real)(ld_regs*, const uint16_t*, int32_t, int, int, int32_t) = NULL;
    if (!real) real = real_sym("update_ld_regs");
    real(regs, r, l, w, d, s);
    FILE* f = logf_();
    fprintf(f, "update_ld_regs l %d w %d d %d s %d -> %u regions\n", l, w, d, s, regs->i);
    for (uint32_t i = 0; i < regs->i; ++i) fprintf(f, "  ld %u %u %u\n", i, regs->reg[i].s, regs->reg[i].e);
    fflush(f);
}

void update_ld_regs_with_refqv(ld_regs* regs, const uint16_t* r, void* ref, const int32_t w, const int32_t s_t, const int32_t e_t, const int32_t d_t,
                                  const uint32_t ide_t, const uint32_t ort_t, const uint32_t irt_t) {
    static void (*real)(ld_regs*, const uint16_t*, void*, int32_t, int32_t, int32_t, int32_t, uint32_t, uint32_t, uint32_t) = NULL;
    if (!real) real = real_sym("update_ld_regs_with_refqv");
    real(regs, r, ref, w, s_t, e_t, d_t, ide_t, ort_t, irt_t);
    FILE* f = logf_();
    fprintf(f, "update_ld_regs_with_refqv w %d d_t %d ide_t %u ort_t %u irt_t %u -> %u regions\n", w, d_t, ide_t, ort_t, irt_t, regs->i);
    for (uint32_t i = 0; i < regs->i; ++i) fprintf(f, "  ld %u %u %u\n", i, regs->reg[i].s, regs->reg[i].e);
    fflush(f);
}

void update_split_p(ld_regs* split_ps, gap_clusters* clusters, ld_regs* regs, const int32_t s, const int32_t l, void* ref) {
    static void (*real)(ld_regs*, gap_clusters*, ld_regs*, int32_t, int32_t, void*) = NULL;
    if (!real) real = real_sym("update_split_p");
    real(split_ps, clusters, regs, s, l, ref);
    FILE* f = logf_();
    fprintf(f, "update_split_p -> %u split points\n", split_ps->i);
    for (uint32_t i = 0; i < split_ps->i; ++i) fprintf(f, "  split %u %u %u\n", i, split_ps->reg[i].s, split_ps->reg[i].e);
    fflush(f);
}
