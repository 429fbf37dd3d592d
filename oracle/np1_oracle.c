/* oracle/np1_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see np1_oracle.h).
 *
 * CPU restatement of nextpolish1.so's score_chain and kmer_count on a decoded record
 * stream.  Every routine cites the reference lines it restates; data structures mirror
 * the reference's per-base lists so that first-seen ordering, uint16 counters and the
 * traversal quirks fall out naturally.  Written from the behaviour described in
 * SURVEY.md appendix A/A2 and checked against the compiled reference (oracle/_ref).
 */
#include "np1_oracle.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define BASE_DEL 3
#define FLAG_ZERO 1
#define FLAG_COVERAGE 2
#define MAX_MAPQ 60

static int64_t g_updates;
int64_t np1o_last_update_count(void) { return g_updates; }
void np1o_free(void* p) { free(p); }

/* ---- nt16 <-> char tables (reference: source/lib/base.c:5-15) */
static const char basetostr[] = "=ACMGRSVTWYHKDBN";
static uint8_t strtobase(uint8_t c) {
    /* the reference table has 100 entries indexed by the upper-cased character */
    switch (c) {
        case '=': return 0;  case 'A': return 1;  case 'C': return 2;  case 'M': return 3;
        case 'G': return 4;  case 'R': return 5;  case 'S': return 6;  case 'V': return 7;
        case 'T': return 8;  case 'W': return 9;  case 'Y': return 10; case 'H': return 11;
        case 'K': return 12; case 'D': return 13; case 'B': return 14;
        default: return 15;
    }
}

/* ---- per-slot state (reference: source/lib/base.h:28-48) */
typedef struct { uint16_t kmer, count; } okmer;
typedef struct { uint8_t base; uint16_t kmer; double score; } oscore;
typedef struct {
    uint8_t base, flag;
    uint16_t refkmer, count;
    uint32_t nk, capk;
    okmer* k;
    uint32_t ns;
    oscore s[16];   /* states are keyed by the 4-bit base => at most 16 */
} oslot;
typedef struct { oslot m; oslot* ins; int32_t nins; } obase;

typedef struct {
    const np1o_contig* in;
    const np1o_configure* cfg;
    obase* b;
    int32_t L;
    int32_t inslength;
    int filter_kind;   /* 1: contig_read_fliter1 (score_chain); 0: contig_read_fliter (kmer_count); 2: contig_read_fliter2 (long reads) */
    int shift;         /* context shift of the pileup (contig.c:360-363): BASE_SHIFT = 4, snp_phase's first pileup uses 16 */
    uint8_t insflag;   /* pass 1 makes insertion columns only behind bases carrying one of these marks; 0 = everywhere */
} octg;

static void slot_init(oslot* s) {   /* base.c:17-32 */
    memset(s, 0, sizeof(*s));
    s->base = 3;
}
static void slot_free(oslot* s) { free(s->k); }

static void slot_add_data(oslot* s, uint16_t kmer) {   /* base.c:60-71, seqlist.c:84-101 (uint16 scan index) */
    okmer* hit = NULL;
    for (uint16_t i = 0; i < s->nk; i++)
        if (s->k[i].kmer == kmer) { hit = &s->k[i]; break; }
    if (!hit) {
        if (s->nk == s->capk) {
            s->capk += s->capk / 2 + 1;
            s->k = (okmer*)realloc(s->k, s->capk * sizeof(okmer));
        }
        s->k[s->nk].kmer = kmer;
        s->k[s->nk].count = 1;
        s->nk++;
    } else {
        hit->count++;
    }
    s->count++;
    g_updates++;
}

static oscore* slot_max_score(oslot* s) {   /* base.c:185-197: first strictly greatest */
    oscore* q = NULL;
    if (s->ns) {
        q = &s->s[0];
        for (uint32_t i = 0; i < s->ns; i++)
            if (s->s[i].score > q->score) q = &s->s[i];
    }
    return q;
}
static oscore* slot_find_score(oslot* s, uint8_t base) {
    for (uint32_t i = 0; i < s->ns; i++)
        if (s->s[i].base == base) return &s->s[i];
    return NULL;
}
static oscore* slot_get_score(oslot* s, uint16_t kmer) {   /* base.c:171-178 */
    if (kmer) return slot_find_score(s, kmer & 0xf);
    return slot_max_score(s);
}
static void slot_add_score(oslot* s, uint16_t kmer, double score) {   /* base.c:159-169 */
    oscore* r = slot_find_score(s, kmer & 0xf);
    if (!r) r = &s->s[s->ns++];
    r->base = kmer & 0xf;
    r->kmer = kmer;
    r->score = score;
}
static double slot_coverage(oslot* s, uint16_t base) {   /* base.c:79-89 */
    uint32_t count = 0;
    for (uint32_t i = 0; i < s->nk; i++)
        if ((s->k[i].kmer & 0xf) == base) count += s->k[i].count;
    return count / (double)s->count;
}

/* ---- traversal in (base, insert-column) order (reference: source/lib/contig.c:385-422) */
static oslot* ctg_next(octg* c, int32_t* i, int32_t* j) {
    if (*i + 1 >= c->L) { *i = c->L; return &c->b[c->L - 1].m; }
    obase* p = &c->b[*i];
    if (p->ins == NULL || p->nins == *j) { (*i)++; *j = 0; return &c->b[*i].m; }
    (*j)++;
    return &p->ins[*j - 1];
}
static oslot* ctg_prev(octg* c, int32_t* i, int32_t* j) {
    if (*i - 1 < 0) { *i = -1; return &c->b[0].m; }
    obase* p = &c->b[*i];
    if (*j == 0) {
        (*i)--;
        p = &c->b[*i];
        if (p->ins != NULL) *j = p->nins;
    } else {
        (*j)--;
    }
    if (*j == 0) return &p->m;
    return &p->ins[*j - 1];
}
#define IN_RANGE(i, j, end) ((i) < (end) || ((i) == (end) && (j) == 0))

/* ---- record helpers */
static inline uint8_t seqi(const uint8_t* s, int32_t i) { return s[i >> 1] >> ((~i & 1) << 2) & 0xf; }
#define OP(c) ((c) & 0xf)
#define OPLEN(c) ((int32_t)((c) >> 4))
enum { CMATCH = 0, CINS = 1, CDEL = 2, CREF_SKIP = 3, CSOFT = 4, CHARD = 5, CPAD = 6, CEQUAL = 7, CDIFF = 8 };

static int32_t read_endpos(const np1o_contig* in, int64_t r) {   /* htslib bam_endpos */
    if (!(in->flag[r] & 4) && in->n_cigar[r] > 0) {
        const uint32_t* cg = in->cigar + in->cigar_off[r];
        int32_t l = 0;
        for (int k = 0; k < in->n_cigar[r]; k++) {
            uint32_t op = OP(cg[k]);
            if (op == CMATCH || op == CDEL || op == CREF_SKIP || op == CEQUAL || op == CDIFF) l += OPLEN(cg[k]);
        }
        return in->pos[r] + (l > 0 ? l : 1);
    }
    return in->pos[r] + 1;
}

static double read_cliprate(const np1o_contig* in, int64_t r) {   /* contig.c:632-646 */
    if (in->n_cigar[r] == 0) return 0;   /* the reference reads out of bounds here; such reads never vote */
    const uint32_t* cg = in->cigar + in->cigar_off[r];
    int32_t addlen = 0;
    if (OP(cg[0]) == CSOFT) addlen += OPLEN(cg[0]);
    uint32_t last = cg[in->n_cigar[r] - 1];
    if (OP(last) == CSOFT) addlen += OPLEN(last);
    return in->l_qseq[r] > 0 ? addlen / (double)in->l_qseq[r] : 0;
}

static uint8_t read_filter(octg* c, int64_t r) {
    const np1o_contig* in = c->in;
    uint8_t result = 0;
    if (c->filter_kind == 2) {   /* contig_read_fliter2, contig.c:679-686 */
        if ((in->flag[r] & 0xD04) == 0 && read_cliprate(in, r) <= c->cfg->max_clip_ratio_lgs) result = 1;
        return result;
    }
    if (c->filter_kind == 1) {   /* contig_read_fliter1, contig.c:667-677 */
        if ((in->flag[r] & 0xC04) == 0) result = 1;
        return result;
    }
    if ((in->flag[r] & 0xC04) == 0) {   /* contig_read_fliter, contig.c:648-665 */
        int32_t length = in->isize[r] >= 0 ? in->isize[r] : -in->isize[r];
        double cliprate = read_cliprate(in, r);
        if ((length > 0 && length < c->cfg->read_tlen) || cliprate < c->cfg->max_clip_ratio_sgs) {
            result = 1;
            if (in->mapq[r] >= c->cfg->min_map_quality && (cliprate < c->cfg->max_clip_ratio_sgs + 0.05)) result = 2;
        }
    }
    return result;
}

/* usable query window (reference: source/lib/contig.c:333-358).  The reference's two
 * homopolymer loops have no bounds checks; running off either end can only produce
 * qstart > qend (the record then contributes nothing), which is what we return. */
static void cut_read(octg* c, int64_t r, int32_t* qstart, int32_t* qend) {
    const np1o_contig* in = c->in;
    const uint32_t* cg = in->cigar + in->cigar_off[r];
    const uint8_t* seq = in->seq + in->seq_off[r];
    int32_t lq = in->l_qseq[r], trim = c->cfg->trim_len_edge, addlen = 0;
    if (OP(cg[0]) == CSOFT) addlen = OPLEN(cg[0]);
    int32_t qs = trim + addlen;
    uint32_t last = cg[in->n_cigar[r] - 1];
    addlen = 0;
    if (OP(last) == CSOFT) addlen = OPLEN(last);
    int32_t qe = lq - trim - addlen - 1;
    if (trim > 0) {
        int dead = 0;
        for (;;) {   /* while (seqi(qs) == seqi(qs-1)) qs++ */
            if (qs >= lq) { dead = 1; break; }
            if (seqi(seq, qs) != seqi(seq, qs - 1)) break;
            qs++;
        }
        while (!dead) {   /* while (seqi(qe) == seqi(qe+1)) qe-- */
            if (qe < 0 || qe + 1 >= lq) { dead = (qe < qs); break; }
            if (seqi(seq, qe) != seqi(seq, qe + 1)) break;
            qe--;
        }
        if (dead || qs > qe) { qs = 1; qe = 0; }
    }
    *qstart = qs;
    *qend = qe;
}

static inline uint16_t left_kmer(uint16_t kmer, uint8_t base) { return (uint16_t)((kmer & 0xff) << 4 | base); }
static inline uint16_t left_kmer_s(uint16_t kmer, uint8_t base, int shift) { return (uint16_t)((uint32_t)(kmer & 0xff) << shift | base); }

/* PASS 1: insertion columns (reference: source/lib/contig.c:202-245, flag argument 0) */
static void parse_read_insert(octg* c, int64_t r, int32_t start, int32_t end) {
    const np1o_contig* in = c->in;
    if (!in->n_cigar[r]) return;
    const uint32_t* cg = in->cigar + in->cigar_off[r];
    int32_t pos = in->pos[r];
    for (int i = 0; i < in->n_cigar[r]; ++i) {
        switch (OP(cg[i])) {
            case CMATCH: case CDEL: pos += OPLEN(cg[i]); break;
            case CINS:
                if (pos > start && pos <= end && (c->insflag == 0 || (c->b[pos - 1].m.flag & c->insflag))) {
                    int32_t len = OPLEN(cg[i]);
                    obase* b = &c->b[pos - 1];
                    if (b->nins < len) {
                        b->ins = (oslot*)realloc(b->ins, (size_t)len * sizeof(oslot));
                        for (int32_t j = b->nins; j < len; j++, c->inslength++) {
                            slot_init(&b->ins[j]);
                            b->ins[j].flag = b->m.flag;
                        }
                        b->nins = len;
                    }
                }
                break;
        }
    }
}

/* PASS 2: per-read pileup of 3-base contexts (reference: source/lib/contig.c:247-331) */
static void parse_read(octg* c, int64_t r, int32_t start, int32_t end) {
    const np1o_contig* in = c->in;
    if (!in->n_cigar[r]) return;
    uint16_t kmer = 0;
    int32_t pos = in->pos[r], qpos = 0, qstart, qend, i, j, k, len;
    uint8_t curcigar, lastcigar = CINS;
    const uint32_t* cg = in->cigar + in->cigar_off[r];
    const uint8_t* seq = in->seq + in->seq_off[r];
    cut_read(c, r, &qstart, &qend);
    for (i = 0; i < in->n_cigar[r]; ++i) {
        len = OPLEN(cg[i]);
        curcigar = OP(cg[i]);
        switch (curcigar) {
            case CMATCH: case CDEL:
                for (j = 0; j < len; j++, pos++) {
                    if (pos >= start && pos <= end && qpos >= qstart && qpos <= qend) {
                        if (lastcigar != CINS && pos > start && (qpos > qstart || (qpos == qstart && lastcigar == CDEL))) {
                            obase* pb = &c->b[pos - 1];
                            for (k = 0; k < pb->nins; k++) {
                                kmer = left_kmer_s(kmer, BASE_DEL, c->shift);
                                slot_add_data(&pb->ins[k], kmer);
                            }
                        }
                        if (curcigar == CDEL) kmer = left_kmer_s(kmer, BASE_DEL, c->shift);
                        else kmer = left_kmer_s(kmer, seqi(seq, qpos), c->shift);
                        slot_add_data(&c->b[pos].m, kmer);
                    }
                    if (curcigar != CDEL) qpos++;
                    lastcigar = curcigar;
                }
                break;
            case CINS:
                if (pos) {
                    obase* pb = (pos - 1 < c->L) ? &c->b[pos - 1] : NULL;
                    for (j = 0; j < len; j++, qpos++) {
                        if (pos > start && pos <= end && qpos >= qstart && qpos <= qend) {
                            kmer = left_kmer_s(kmer, seqi(seq, qpos), c->shift);
                            slot_add_data(&pb->ins[j], kmer);
                        }
                    }
                    if (pos > start && pos <= end && qpos > qstart && qpos <= qend + 1) {
                        for (; j < pb->nins; j++) {
                            kmer = left_kmer_s(kmer, BASE_DEL, c->shift);
                            slot_add_data(&pb->ins[j], kmer);
                        }
                    }
                    lastcigar = curcigar;
                } else {
                    qpos += len;
                    qstart += len;
                    lastcigar = curcigar;
                }
                break;
            case CHARD: case CSOFT:
                qpos += len;
                break;
        }
        if (pos > end) break;
    }
}

/* region iteration = htslib overlap query in file order (hts.c hts_itr_next):
 * records with pos < end+1 and endpos > start; stop at the first record with pos >= end+1.
 * first_candidate(): records are position sorted, so nothing before the first record whose
 * pos could reach `start` matters; we simply scan from a lower bound computed with max_span. */
typedef struct { int64_t lo; int32_t max_span; } oscan;
static int32_t stream_max_span(const np1o_contig* in) {
    int32_t m = 1;
    for (int64_t r = 0; r < in->n_reads; r++) {
        int32_t s = read_endpos(in, r) - in->pos[r];
        if (s > m) m = s;
    }
    return m;
}
static int64_t lower_bound_pos(const np1o_contig* in, int32_t p) {   /* first r with pos[r] >= p */
    int64_t lo = 0, hi = in->n_reads;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (in->pos[mid] < p) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static void create_insert(octg* c, int32_t start, int32_t end, int32_t max_span) {   /* contig.c:170-180 */
    const np1o_contig* in = c->in;
    for (int64_t r = lower_bound_pos(in, start - max_span); r < in->n_reads; r++) {
        if (in->pos[r] >= end + 1) break;
        if (read_endpos(in, r) <= start) continue;
        if (read_filter(c, r) >= 1) parse_read_insert(c, r, start, end);
    }
}

static void parse_region(octg* c, int32_t start, int32_t end, uint8_t level, int32_t max_span) {   /* contig.c:688-704 */
    const np1o_contig* in = c->in;
    for (int64_t r = lower_bound_pos(in, start - max_span); r < in->n_reads; r++) {
        if (in->pos[r] >= end + 1) break;
        if (read_endpos(in, r) <= start) continue;
        if (read_filter(c, r) == level) parse_read(c, r, start, end);
    }
}

static void as_read(octg* c, int32_t start, int32_t end) {   /* contig.c:373-383 */
    uint16_t kmer = 0;
    int32_t i = start, j = 0;
    oslot* p = &c->b[start].m;
    while (IN_RANGE(i, j, end)) {
        p->refkmer = kmer = left_kmer(kmer, p->base);
        slot_add_data(p, kmer);
        p = ctg_next(c, &i, &j);
    }
}

static void calculate_score(oslot* cur, oslot* last, double rate) {   /* contig.c:424-454 */
    cur->ns = 0;
    double score = 0;
    uint16_t temp, count, total = cur->count;
    if (total > 1) total--;
    for (uint32_t i = 0; i < cur->nk; i++) {
        okmer* p = &cur->k[i];
        temp = p->kmer >> 4;
        if ((temp & 0xf) == 0) score = slot_max_score(last)->score;
        else score = slot_get_score(last, temp)->score;
        count = p->count;
        if (p->kmer == cur->refkmer && cur->count > 1) count--;
        score += count - total * rate;
        oscore* q = slot_get_score(cur, p->kmer);
        if (q == NULL || q->score < score) slot_add_score(cur, p->kmer, score);
    }
}

static void region_score(octg* c, int32_t start, int32_t end, double rate) {   /* contig.c:456-471 */
    int32_t i = start, j = 0;
    oslot temp;
    slot_init(&temp);
    oslot *p = &temp, *q = &c->b[start].m;
    for (uint32_t t = 0; t < q->nk; t++) slot_add_score(&temp, q->k[t].kmer >> 4, 0);
    while (IN_RANGE(i, j, end)) {
        calculate_score(q, p, rate);
        p = q;
        q = ctg_next(c, &i, &j);
    }
}

static void region_correct(octg* c, int32_t start, int32_t end) {   /* contig.c:473-496 */
    oslot* base = &c->b[end].m;
    oscore* score = slot_max_score(base);
    int32_t i = end, j = 0;
    while (i > start || (i == start && j == 0)) {
        base->base = score->base;
        if (base->count == 1) base->flag |= FLAG_ZERO; else base->flag &= (uint8_t)~FLAG_ZERO;
        if (slot_coverage(base, base->base) < c->cfg->min_count_ratio_skip) base->flag |= FLAG_COVERAGE;
        else base->flag &= (uint8_t)~FLAG_COVERAGE;
        base = ctg_prev(c, &i, &j);
        score = slot_get_score(base, score->kmer >> 4);
    }
}

/* ---- low-quality regions (reference: source/lib/contig.c:498-620) */
typedef struct { int32_t* v; int32_t n, cap; } ilist;
static void il_push(ilist* l, int32_t x) {
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 64; l->v = (int32_t*)realloc(l->v, l->cap * sizeof(int32_t)); }
    l->v[l->n++] = x;
}
static void brim_no_ext(octg* c, uint8_t flag, int32_t bstart, int32_t bend, int32_t* start, int32_t* end) {
    (void)flag;
    int32_t e = c->cfg->ext_len_edge;
    *start = *start >= bstart + e ? *start - e : bstart;
    *end = *end <= bend - e ? *end + e : bend;
}
static void brim_with_ext(octg* c, uint8_t flag, int32_t bstart, int32_t bend, int32_t* start, int32_t* end) {
    brim_no_ext(c, flag, bstart, bend, start, end);
    int32_t p = *start + 1;
    while (*start > bstart && (c->b[p].m.base == c->b[p - 1].m.base || (c->b[p - 1].m.flag & flag) != 0)) {
        (*start)--;
        p--;
    }
    p = *end - 1;
    while (*end < bend && (c->b[p].m.base == c->b[p + 1].m.base || (c->b[p + 1].m.flag & flag) != 0)) {
        (*end)++;
        p++;
    }
}
typedef void (*brimfn)(octg*, uint8_t, int32_t, int32_t, int32_t*, int32_t*);

static ilist get_region(octg* c, int32_t start, int32_t end, uint16_t gap, uint16_t con, uint8_t flag, brimfn brim) {
    ilist result = {0, 0, 0};
    int32_t i = start, j = 0, qstart = -1, qend = -1;
    uint16_t pgap = 0, pcon = 0;
    oslot* p = &c->b[start].m;
    while (IN_RANGE(i, j, end)) {
        if ((p->flag & flag) != 0) {
            if (qstart == -1) { qstart = i; pcon = 1; }
            else if (pgap == 0) pcon++;
            else pcon = 1;
            pgap = 0;
            qend = i;
        } else if (qstart != -1) {
            pgap++;
            if (pgap > gap) {
                if (pcon > con) {
                    brim(c, flag, start, end, &qstart, &qend);
                    il_push(&result, qstart);
                    il_push(&result, qend);
                    if (qend > i) { i = qend; j = 0; }
                }
                qstart = qend = -1;
            }
        }
        p = ctg_next(c, &i, &j);
    }
    if (qstart != -1) {
        brim(c, flag, start, end, &qstart, &qend);
        il_push(&result, qstart);
        il_push(&result, qend);
    }
    return result;
}

static void merge_region(ilist* l) {   /* contig.c:595-620, literal */
    if (l->n == 0) return;
    int32_t *pstart = l->v, *pend = pstart + 1, *qstart = pstart, *qend = pend, length = 2;
    for (int i = 0; i < l->n; i += 2) {
        if (*pstart >= *qend) {
            qstart += 2;
            qend = qstart + 1;
            if (qstart != pstart) *qstart = *pstart;
            if (qend != pend) *qend = *pend;
            length += 2;
        } else {
            while (*pstart < *qstart) qstart -= 2;
            qend = qstart + 1;
            *qend = *pend;
        }
        pstart += 2;
        pend = pstart + 1;
    }
    l->n = length;
}

/* contig_score_correct (reference: source/lib/contig.c:706-734) */
static void score_correct(octg* c, int32_t start, int32_t end, int32_t flag, double rate, int32_t max_span) {
    int32_t level = flag & 0xf, insert = (flag >> 4) & 0xf;
    if ((insert & 0x1) == 0) create_insert(c, start, end, max_span);
    as_read(c, start, end);
    parse_region(c, start, end, (uint8_t)level, max_span);
    region_score(c, start, end, rate);
    region_correct(c, start, end);
    if (level == 2) {
        ilist nd = get_region(c, start, end, 0, 0, 1, brim_no_ext);
        if (nd.n != 0) {
            merge_region(&nd);
            for (int i = 0; i < nd.n; i += 2) {
                parse_region(c, nd.v[i], nd.v[i + 1], 1, max_span);
                region_score(c, nd.v[i], nd.v[i + 1], c->cfg->indel_balance_factor_sgs);
                region_correct(c, nd.v[i], nd.v[i + 1]);
            }
        }
        free(nd.v);
    }
}

/* contig_get_contig (reference: source/lib/contig.c:736-799).  With cfg->trace_polish_open the change list (PolishPoint: pos, index,
 * curbase, base -- contig.c:743-797) of the call is kept per thread and read back with np1o_last_points; every task ends here. */
static __thread int32_t* g_points = NULL;      /* 4 words per point: pos, index, curbase, base */
static __thread int32_t g_npoints = 0, g_cappoints = 0;
static void point_add(int32_t pos, int32_t index, char cur, char was) {
    if (g_npoints == g_cappoints) {
        g_cappoints = g_cappoints ? 2 * g_cappoints : 1024;
        g_points = (int32_t*)realloc(g_points, (size_t)g_cappoints * 4 * sizeof(int32_t));
    }
    int32_t* p = g_points + 4 * (size_t)g_npoints++;
    p[0] = pos; p[1] = index; p[2] = (uint8_t)cur; p[3] = (uint8_t)was;
}
int32_t np1o_last_points(const int32_t** out) {
    *out = g_points;
    return g_npoints;
}
static char* get_contig(octg* c, int32_t start, int32_t end, uint8_t flag, int32_t* out_len) {
    int32_t i = start, j = 0, length = 0;
    char* result = (char*)calloc(1, (size_t)c->L + (size_t)c->inslength + 1), *q = result;
    oslot* p = &c->b[start].m;
    uint8_t sign = 0;
    const int trace = c->cfg->trace_polish_open != 0;
    g_npoints = 0;
    while (IN_RANGE(i, j, end)) {
        const char was = (char)toupper((unsigned char)c->in->draft[i]);
        if (p->base == 3) {
            if ((p->flag & flag) != 0) sign = 1;
            if (trace && j == 0) point_add(i, j, '.', was);
        } else {
            *q = basetostr[p->base];
            if (trace) {
                if (j != 0) point_add(i, j, *q, '.');
                else if (*q != was) point_add(i, j, *q, was);
            }
            if (sign || (p->flag & flag) != 0) { *q += 32; sign = 0; }
            q++;
            length++;
        }
        p = ctg_next(c, &i, &j);
    }
    result[length] = '\0';
    *out_len = length;
    return result;
}

static octg* ctg_init(const np1o_contig* in, const np1o_configure* cfg) {   /* contig.c:81-102 */
    octg* c = (octg*)calloc(1, sizeof(octg));
    c->in = in;
    c->cfg = cfg;
    c->shift = 4;
    c->L = in->length;
    c->b = (obase*)calloc((size_t)(in->length > 0 ? in->length : 1), sizeof(obase));
    for (int32_t i = 0; i < in->length; i++) {
        slot_init(&c->b[i].m);
        uint8_t ch = (uint8_t)in->draft[i];
        if (ch >= 97 && ch <= 122) { ch -= 32; c->b[i].m.flag |= FLAG_ZERO; }
        c->b[i].m.base = strtobase(ch);
    }
    return c;
}
static void ctg_free(octg* c) {
    for (int32_t i = 0; i < c->L; i++) {
        for (int32_t j = 0; j < c->b[i].nins; j++) slot_free(&c->b[i].ins[j]);
        free(c->b[i].ins);
        slot_free(&c->b[i].m);
    }
    free(c->b);
    free(c);
}

void np1o_default_config(np1o_configure* r) {   /* config.c:8-38 */
    memset(r, 0, sizeof(*r));
    r->trim_len_edge = 2; r->ext_len_edge = 2; r->min_map_quality = 0;
    r->indel_balance_factor_sgs = 0.5; r->min_count_ratio_skip = 0.8;
    r->min_len_ldr = 3; r->min_len_inter_kmer = 5; r->max_len_kmer = 50; r->max_count_kmer = 50;
    r->min_depth_snp = 3; r->min_count_snp = 5; r->min_count_snp_link = 5;
    r->ploidy = 2; r->indel_balance_factor_lgs = 0.33; r->max_indel_factor_lgs = 0.21;
    r->max_snp_factor_lgs = 0.53; r->min_snp_factor_sgs = 0.34;
    r->region_count = 10000; r->count_read_ins_sgs = 10000; r->max_ins_len_sgs = 10000;
    r->max_ins_fold_sgs = 5; r->max_variant_count_lgs = 150000;
    r->max_clip_ratio_sgs = 0.15; r->max_clip_ratio_lgs = 0.4;
}

/* score_chain (reference: source/lib/scorechain.c:3-15) */
char* np1o_score_chain(const np1o_contig* in, const np1o_configure* cfg, int32_t* out_len) {
    g_updates = 0;
    if (in->length <= 0) { *out_len = 0; return (char*)calloc(1, 1); }
    octg* c = ctg_init(in, cfg);
    c->filter_kind = 1;
    int32_t max_span = stream_max_span(in);
    score_correct(c, 0, c->L - 1, 0x1, cfg->indel_balance_factor_sgs, max_span);
    char* out = get_contig(c, 0, c->L - 1, FLAG_ZERO | FLAG_COVERAGE, out_len);
    ctg_free(c);
    return out;
}

/* ================= kmer_count (reference: source/lib/kmercount.c) ================= */

static void create_insert_region(octg* c, ilist* regs, int32_t max_span) {   /* contig.c:182-200 */
    c->filter_kind = 0;
    for (int i = 0; i < regs->n; i += 2) create_insert(c, regs->v[i], regs->v[i + 1], max_span);
}

static int32_t get_length(octg* c, int32_t start, int32_t end) {   /* contig.c:801-809 */
    int32_t i = start, j = 0, length = 0;
    while (IN_RANGE(i, j, end)) { length++; ctg_next(c, &i, &j); }
    return length;
}

static ilist split_region(octg* c, ilist* regs, uint8_t flag, uint8_t max) {   /* kmercount.c:128-173 */
    ilist result = {0, 0, 0}, temp = {0, 0, 0};
    for (int i = 0; i < regs->n; i += 2) {
        int32_t* p = &regs->v[i];
        il_push(&result, p[0]);
        if (p[1] - p[0] > max) {
            int32_t j = p[0], k = 0, qstart = -1, qend = -1;
            oslot* q = &c->b[j].m;
            temp.n = 0;
            while (IN_RANGE(j, k, p[1])) {
                if ((q->flag & flag) != 0) break;
                q = ctg_next(c, &j, &k);
            }
            while (IN_RANGE(j, k, p[1])) {
                if ((q->flag & flag) == 0) {
                    if (qstart == -1) qstart = j;
                    qend = j;
                } else if (qstart != -1) {
                    il_push(&temp, qstart);
                    il_push(&temp, qend);
                    qstart = qend = -1;
                }
                q = ctg_next(c, &j, &k);
            }
            for (j = 0; j < temp.n; j += 2) {
                k = (temp.v[j] + temp.v[j + 1]) >> 1;
                il_push(&result, k);
                il_push(&result, k);
            }
        }
        il_push(&result, p[1]);
    }
    free(temp.v);
    return result;
}

typedef struct { uint8_t* region; int32_t length, qual, mapqual, num; } okscore;

/* set when the walk would touch insertion columns that were never created (a snp_valid second-round pair reaching outside its
 * region): the reference dereferences a null list there */
static int g_undefined = 0;

/* ss_parse_read_kmer with left = right = -1 (reference: source/lib/kmercount.c:365-465); flagzero != 0: the FLAG_ZERO marks of
 * the covered slots are left alone (snp_valid's first round) */
static int32_t parse_read_kmer(octg* c, int64_t r, int32_t start, int32_t end, okscore* ks, int32_t left, int32_t right, int flagzero) {
    const np1o_contig* in = c->in;
    int32_t result = 0;
    if (!in->n_cigar[r]) return 0;
    int32_t pos = in->pos[r], qpos = 0, qstart, qend, i, j, k, len, del = 0;
    uint8_t curcigar, lastcigar = CINS;
    const uint32_t* cg = in->cigar + in->cigar_off[r];
    const uint8_t* seq = in->seq + in->seq_off[r];
    const uint8_t* qual = in->qual + in->qual_off[r];
    cut_read(c, r, &qstart, &qend);
    ks->mapqual = in->mapq[r];
    for (i = 0; i < in->n_cigar[r]; ++i) {
        len = OPLEN(cg[i]);
        curcigar = OP(cg[i]);
        switch (curcigar) {
            case CMATCH: case CDEL:
                for (j = 0; j < len; j++, pos++) {
                    if (pos >= start && pos <= end && qpos >= qstart && qpos <= qend) {
                        if (lastcigar != CINS && pos > start && (qpos > qstart || (qpos == qstart && lastcigar == CDEL))) {
                            obase* pb = &c->b[pos - 1];
                            for (k = 0; k < pb->nins; k++) {
                                ks->region[ks->length++] = BASE_DEL;
                                if (!flagzero) pb->ins[k].flag &= (uint8_t)~FLAG_ZERO;
                                del++;
                            }
                        }
                        if (curcigar == CDEL) {
                            ks->region[ks->length++] = BASE_DEL;
                        } else {
                            ks->region[ks->length++] = seqi(seq, qpos);
                            ks->qual += qual[qpos];
                        }
                        if (!flagzero) c->b[pos].m.flag &= (uint8_t)~FLAG_ZERO;
                    }
                    if (left == pos || right == pos) {   /* kmercount.c:416-420: the read agrees with the draft at the two anchor columns */
                        if (pos < c->L && qpos < in->l_qseq[r] && seqi(seq, qpos) == c->b[pos].m.base) result++;
                    }
                    if (curcigar != CDEL) qpos++;
                    lastcigar = curcigar;
                }
                break;
            case CINS:
                if (pos) {
                    obase* pb = (pos - 1 < c->L) ? &c->b[pos - 1] : NULL;
                    for (j = 0; j < len; j++, qpos++) {
                        if (pos > start && pos <= end && qpos >= qstart && qpos <= qend) {
                            if (!pb || j >= pb->nins) { g_undefined = 1; return result; }
                            ks->region[ks->length++] = seqi(seq, qpos);
                            ks->qual += qual[qpos];
                            if (!flagzero) pb->ins[j].flag &= (uint8_t)~FLAG_ZERO;
                        }
                    }
                    if (pos > start && pos <= end && qpos > qstart && qpos <= qend + 1) {
                        for (; j < pb->nins; j++) {
                            ks->region[ks->length++] = BASE_DEL;
                            if (!flagzero) pb->ins[j].flag &= (uint8_t)~FLAG_ZERO;
                            del++;
                        }
                    }
                    lastcigar = curcigar;
                } else {
                    qpos += len;
                    qstart += len;
                    lastcigar = curcigar;
                }
                break;
            case CHARD: case CSOFT:
                qpos += len;
                break;
        }
        if (pos > end) break;
    }
    if (ks->length > 0 && ks->length != del) ks->qual /= ks->length - del;
    else ks->qual = 0;
    return result;
}

typedef struct { okscore* v; int32_t n, cap; } kslist;

/* ss_kmer_get_region (reference: source/lib/kmercount.c:332-363); returns nothing, mutates ks */
static void kmer_get_region(octg* c, int64_t r, int32_t start, int32_t end, int32_t length, kslist* rd, okscore* ks, int flagzero) {
    if (ks->region == NULL) ks->region = (uint8_t*)calloc(1, (size_t)length + 8);
    parse_read_kmer(c, r, start, end, ks, -1, -1, flagzero);
    if (ks->length == length) {
        okscore* hit = NULL;
        for (int32_t i = 0; i < rd->n; i++)
            if (memcmp(rd->v[i].region, ks->region, (size_t)rd->v[i].length) == 0) { hit = &rd->v[i]; break; }
        if (!hit) {
            ks->num = 1;
            if (rd->n == rd->cap) { rd->cap = rd->cap ? rd->cap * 2 : 16; rd->v = (okscore*)realloc(rd->v, rd->cap * sizeof(okscore)); }
            rd->v[rd->n++] = *ks;
            ks->region = NULL;
        } else {
            hit->num++;
            hit->mapqual += ks->mapqual;
            hit->qual += ks->qual;
        }
    } else {
        ks->mapqual = 0;
    }
}
static void ks_clean(okscore* ks, int32_t length) {   /* kmercount.c:24-33 */
    if (ks->region == NULL) ks->region = (uint8_t*)calloc(1, (size_t)length + 8);
    ks->length = 0; ks->qual = 0; ks->mapqual = 0; ks->num = 0;
}
static int ks_compare(const okscore* a, const okscore* b) {   /* kmercount.c:63-88 */
    if (a == b) return 0;
    if (a->num != b->num) return a->num > b->num ? 1 : -1;
    if (a->mapqual != b->mapqual) return a->mapqual > b->mapqual ? 1 : -1;
    if (a->qual != b->qual) return a->qual > b->qual ? 1 : -1;
    return 0;
}

/* ---- replay of the reference's region iterator (contig.c:982-1043 over htslib 1.9 hts.c:2086-2189,2614-2655) on the decoded
 * stream: records are addressed by their BGZF virtual offsets, the chunk lists come from the BAI.  Used when the caller hands in
 * voff / voff_end / idx (a stream read from a file). */
typedef struct { uint64_t u, v; } opair;
typedef struct {
    opair* off;
    int n_off, i;
    uint64_t curr_off;
    int finished;
    int32_t beg, end, curr_end;
} oitr;
typedef struct {      /* one of ss_kmer_correct's two iterators with the variables it is threaded through */
    oitr* it;
    int32_t iterend;
    uint64_t saved_off;
    int32_t saved_end;
    uint64_t fpos;    /* file position */
    int64_t buffer;   /* record last read into the bam1_t: -1 none yet, -2 another contig's */
} ochan;

static int idx_find(const np1o_index* x, uint32_t bin) {
    int lo = 0, hi = x->n_bins - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        if (x->bin[mid] == bin) return mid;
        if (x->bin[mid] < bin) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
static int pair_cmp(const void* a, const void* b) {
    uint64_t x = ((const opair*)a)->u, y = ((const opair*)b)->u;
    return x < y ? -1 : x > y;
}
#define BIN_FIRST(l) (((1 << (((l) << 1) + (l))) - 1) / 7)
#define BIN_PARENT(b) (((b) - 1) >> 3)
static oitr* itr_query(const np1o_index* x, int32_t beg, int32_t end) {   /* hts_itr_query, tid >= 0 */
    const int n_lvls = 5, min_shift = 14, n_bins_all = 37449;
    oitr* it = (oitr*)calloc(1, sizeof(oitr));
    if (beg < 0) beg = 0;
    it->beg = beg; it->end = end; it->i = -1;
    if (x->n_bins == 0) { it->finished = 1; return it; }
    int bin = BIN_FIRST(n_lvls) + (beg >> min_shift), k;
    do {
        k = idx_find(x, (uint32_t)bin);
        if (k >= 0) break;
        int first = (BIN_PARENT(bin) << 3) + 1;
        if (bin > first) --bin; else bin = BIN_PARENT(bin);
    } while (bin);
    if (bin == 0) k = idx_find(x, 0);
    const uint64_t min_off = k >= 0 ? x->loff[k] : 0;
    uint64_t max_off;
    bin = BIN_FIRST(n_lvls) + ((end - 1) >> min_shift) + 1;
    if (bin >= n_bins_all) bin = 0;
    for (;;) {
        while (bin % 8 == 1) bin = BIN_PARENT(bin);
        if (bin == 0) { max_off = (uint64_t)-1; break; }
        k = idx_find(x, (uint32_t)bin);
        if (k >= 0 && x->chunk_first[k + 1] > x->chunk_first[k]) { max_off = x->chunk_u[x->chunk_first[k]]; break; }
        bin++;
    }
    /* reg2bins + the chunks of the bins that exist */
    int n_off = 0, cap = 64;
    opair* off = (opair*)malloc((size_t)cap * sizeof(opair));
    {
        int64_t e = end;
        int l, t, s = min_shift + (n_lvls << 1) + n_lvls;
        if (beg < e) {
            if (e >= 1LL << s) e = 1LL << s;
            for (--e, l = 0, t = 0; l <= n_lvls; s -= 3, t += 1 << ((l << 1) + l), ++l) {
                int b = t + (int)(beg >> s), ee = t + (int)(e >> s);
                for (int bb = b; bb <= ee; ++bb) {
                    k = idx_find(x, (uint32_t)bb);
                    if (k < 0) continue;
                    for (uint32_t j = x->chunk_first[k]; j < x->chunk_first[k + 1]; ++j)
                        if (x->chunk_v[j] > min_off && x->chunk_u[j] < max_off) {
                            if (n_off == cap) { cap *= 2; off = (opair*)realloc(off, (size_t)cap * sizeof(opair)); }
                            off[n_off].u = x->chunk_u[j]; off[n_off].v = x->chunk_v[j]; ++n_off;
                        }
                }
            }
        }
    }
    if (n_off == 0) { free(off); it->finished = 1; return it; }
    qsort(off, (size_t)n_off, sizeof(opair), pair_cmp);
    int i, l;
    for (i = 1, l = 0; i < n_off; ++i) if (off[l].v < off[i].v) off[++l] = off[i];
    n_off = l + 1;
    for (i = 1; i < n_off; ++i) if (off[i - 1].v >= off[i].u) off[i - 1].v = off[i].u;
    for (i = 1, l = 0; i < n_off; ++i) {
        if (off[l].v >> 16 == off[i].u >> 16) off[l].v = off[i].v;
        else off[++l] = off[i];
    }
    it->n_off = l + 1;
    it->off = off;
    return it;
}
static void itr_free(oitr* it) { if (it) { free(it->off); free(it); } }

/* record that starts at (or, for an offset written before a block boundary was crossed, first behind) a virtual offset; -1 when
 * that is behind this contig's records */
static int64_t rec_at(const np1o_contig* in, uint64_t v) {
    int64_t lo = 0, hi = in->n_reads;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (in->voff[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo < in->n_reads ? lo : -1;
}
static int64_t itr_next(const np1o_contig* in, ochan* ch) {   /* hts_itr_next + the bam readrec */
    oitr* it = ch->it;
    if (it == NULL || it->finished) return -1;
    int64_t ret = -1;
    for (;;) {
        if (it->curr_off == 0 || (it->i >= 0 && it->curr_off >= it->off[it->i].v)) {
            if (it->i == it->n_off - 1) break;
            if (it->i < 0 || it->off[it->i].v != it->off[it->i + 1].u) { ch->fpos = it->off[it->i + 1].u; it->curr_off = ch->fpos; }
            ++it->i;
        }
        const int64_t r = rec_at(in, ch->fpos);
        if (r < 0) {                      /* behind the contig: another contig's record (read, then rejected) or the end of the file */
            if (in->has_next) ch->buffer = -2;
            break;
        }
        ch->fpos = in->voff_end[r];
        it->curr_off = ch->fpos;
        ch->buffer = r;
        const int32_t beg = in->pos[r], end = read_endpos(in, r);
        if (beg >= it->end) break;
        if (end > it->beg && it->end > beg) { it->curr_end = end; return r; }
    }
    it->finished = 1;
    return ret;
}
/* contig_next_iter (contig.c:1010-1043) with contig_update_iter (contig.c:982-1008); flag as there */
static int64_t chan_next(octg* c, ochan* ch, int32_t start, int32_t end, int32_t* nextposend, int flag) {
    const np1o_contig* in = c->in;
    if (flag >= 1) {
        if (ch->it != NULL) {
            if (end < ch->iterend) { ch->it->beg = start; ch->it->end = end + 1; ch->it->finished = 0; }
            else { itr_free(ch->it); ch->it = NULL; }
        }
        if (ch->it == NULL) {
            ch->it = itr_query(in->idx, start, end + 1);
            if (ch->it->off) {
                const int64_t r = rec_at(in, ch->it->off[0].v);
                ch->iterend = r >= 0 ? in->pos[r] : c->L;
            }
        }
        if (ch->it->curr_off) {
            ch->it->curr_off = ch->saved_off;
            ch->it->curr_end = ch->saved_end;
            ch->fpos = ch->it->curr_off;
            if (ch->it->curr_off == 0) ch->it->i = -1;
        } else {
            ch->saved_off = 0;
        }
        if (flag == 1) { int32_t t = ch->it->beg; ch->it->beg = ch->it->end; ch->it->end = t; }
    }
    const int64_t ret = ch->it->off ? itr_next(in, ch) : -1;
    if (ret >= 0 && ch->it->curr_end <= *nextposend) { ch->saved_off = ch->it->curr_off; ch->saved_end = ch->it->curr_end; }
    else *nextposend = -1;
    return ret;
}

/* ss_kmer_correct (reference: source/lib/kmercount.c:175-261); the values of `regs` are read pairwise, n_vals of them are valid
 * (an odd count makes the last pair read one stored value further, as the reference does); nodepth != NULL collects the
 * regions no record spans; flagzero: see parse_read_kmer.
 * Spanning query = swapped-interval iterator (contig.c:1130-1135): records with pos < start and
 * endpos > end+1 in file order, iteration ends at the first record with pos >= start; that
 * terminating record is the "stale read" the reference's fallback loop keeps re-parsing. */
static void kmer_correct(octg* c, ilist* regs, int32_t max_span, ilist* nodepth, int flagzero) {
    const np1o_contig* in = c->in;
    kslist rd = {0, 0, 0};
    c->filter_kind = 0;
    const int replay = in->idx != NULL && in->voff != NULL && in->voff_end != NULL;
    ochan ch1, ch2;
    memset(&ch1, 0, sizeof(ch1));
    memset(&ch2, 0, sizeof(ch2));
    ch1.buffer = ch2.buffer = -1;
    for (int ri = 0; ri < regs->n; ri += 2) {
        int32_t start = regs->v[ri], end = regs->v[ri + 1];
        int32_t length = get_length(c, start, end), count = 0;
        okscore ks;
        memset(&ks, 0, sizeof(ks));
        int have_ks = 0;
        if (replay) {   /* the two loops of kmercount.c:196-219 on the replayed iterators */
            const int32_t nextposend = ri + 2 < regs->n ? regs->v[ri + 3] : -1;
            int32_t np1 = nextposend;
            int flag = 1;
            int64_t r;
            while ((r = chan_next(c, &ch1, start, end, &np1, flag)) >= 0) {
                if (read_filter(c, r) == 2) {
                    kmer_get_region(c, r, start, end, length, &rd, &ks, flagzero);
                    have_ks = 1;
                    if (ks.mapqual == MAX_MAPQ) {
                        count++;
                        if (count >= c->cfg->max_count_kmer) break;
                    }
                    ks_clean(&ks, length);
                }
                flag = 0;
            }
            if (rd.n == 0) {
                flag = 1;
                np1 = nextposend;
                while (chan_next(c, &ch2, start, end, &np1, flag) >= 0) {
                    if (ch1.buffer >= 0 && read_filter(c, ch1.buffer) == 1) {   /* (kmercount.c:214: the FIRST iterator's record) */
                        kmer_get_region(c, ch1.buffer, start, end, length, &rd, &ks, flagzero);
                        have_ks = 1;
                        ks_clean(&ks, length);
                    }
                    flag = 0;
                }
            }
            goto vote;
        }
        int64_t r0 = lower_bound_pos(in, start - max_span);
        int64_t rstop = lower_bound_pos(in, start);   /* first record with pos >= start */
        int64_t n_span = 0;
        for (int64_t r = r0; r < rstop; r++) {
            if (!(read_endpos(in, r) > end + 1)) continue;
            n_span++;
            if (read_filter(c, r) == 2) {
                kmer_get_region(c, r, start, end, length, &rd, &ks, flagzero);
                have_ks = 1;
                if (ks.mapqual == MAX_MAPQ) {
                    count++;
                    if (count >= c->cfg->max_count_kmer) break;
                }
                ks_clean(&ks, length);
            }
        }
        /* the record the first loop stopped on: first record with pos >= start; when the contig has none, the last record read --
         * the contig's last one: the chunk list of the query ends with the contig's records (the index closes a chunk where the
         * reference id changes, hts.c hts_idx_push), so the reader never gets as far as another contig's record */
        int64_t stale = -1;
        if (rstop < in->n_reads) stale = rstop;
        else if (in->n_reads > 0) stale = in->n_reads - 1;
        if (rd.n == 0 && stale >= 0) {
            /* bug-compatible fallback (kmercount.c:212-217): one pass per spanning record, always
             * testing and parsing the stale record */
            for (int64_t t = 0; t < n_span; t++) {
                if (read_filter(c, stale) == 1) {
                    kmer_get_region(c, stale, start, end, length, &rd, &ks, flagzero);
                    have_ks = 1;
                    ks_clean(&ks, length);
                }
            }
        }
    vote:
        if (rd.n > 0) {
            if (flagzero) {   /* contig_clean_flag(start, end, FLAG_ZERO_N), contig.c:823-831 */
                int32_t i = start, j = 0;
                oslot* p = &c->b[start].m;
                while (IN_RANGE(i, j, end)) { p->flag &= (uint8_t)~FLAG_ZERO; p = ctg_next(c, &i, &j); }
            }
            okscore* best = NULL;
            if (count == c->cfg->max_count_kmer) {
                int32_t want = MAX_MAPQ * count;
                for (int32_t i = 0; i < rd.n; i++)
                    if (rd.v[i].mapqual == want) { best = &rd.v[i]; break; }
            }
            if (best == NULL) {
                best = &rd.v[0];
                for (int32_t i = 0; i < rd.n; i++)
                    if (ks_compare(best, &rd.v[i]) < 0) best = &rd.v[i];
            }
            /* contig_update_contig, contig.c:811-821 */
            int32_t i = start, j = 0;
            oslot* p = &c->b[start].m;
            uint8_t* q = best->region;
            while (IN_RANGE(i, j, end)) {
                p->base = *q;
                p = ctg_next(c, &i, &j);
                q++;
            }
        } else if (nodepth) {
            il_push(nodepth, start);
            il_push(nodepth, end);
        }
        if (have_ks) free(ks.region);
        for (int32_t i = 0; i < rd.n; i++) free(rd.v[i].region);
        rd.n = 0;
    }
    free(rd.v);
    itr_free(ch1.it);
    itr_free(ch2.it);
}

char* np1o_kmer_count(const np1o_contig* in, const np1o_configure* cfg, int32_t* out_len) {   /* kmercount.c:93-126 */
    g_updates = 0;
    if (in->length <= 0) { *out_len = 0; return (char*)calloc(1, 1); }
    octg* c = ctg_init(in, cfg);
    c->filter_kind = 0;
    int32_t max_span = stream_max_span(in);
    ilist nodepth = get_region(c, 0, c->L - 1, 0, cfg->min_len_ldr, 0x1, brim_no_ext);
    ilist kreg = get_region(c, 0, c->L - 1, cfg->min_len_inter_kmer, 0, 0x1, brim_with_ext);
    if (kreg.n > 0) {
        merge_region(&kreg);
        create_insert_region(c, &kreg, max_span);
    }
    if (nodepth.n > 0) {
        merge_region(&nodepth);
        create_insert_region(c, &nodepth, max_span);
        for (int i = 0; i < nodepth.n; i += 2)
            score_correct(c, nodepth.v[i], nodepth.v[i + 1], 0x12, cfg->indel_balance_factor_sgs, max_span);
    }
    free(nodepth.v);
    if (kreg.n > 0) {
        ilist parts = split_region(c, &kreg, 0x1, cfg->max_len_kmer);
        kmer_correct(c, &parts, max_span, NULL, 0);
        free(parts.v);
    }
    free(kreg.v);
    char* out = get_contig(c, 0, c->L - 1, FLAG_ZERO, out_len);
    ctg_free(c);
    return out;
}

/* ================= snp_valid (reference: source/lib/snpvalid.c) ================= */

/* fts_spilt_region (snpvalid.c:38-66): appends to `result` (which is NOT started with `start`) */
static void fts_split_region(octg* c, int32_t start, int32_t end, uint8_t flag, ilist* result) {
    int32_t i = start, j = 0, qstart = -1, qend = -1;
    oslot* p = &c->b[start].m;
    while (IN_RANGE(i, j, end)) {
        if ((p->flag & flag) == 0) {
            if (qstart == -1) qstart = i;
            qend = i;
        } else if (qstart != -1) {
            int32_t count = 2;
            if (qstart == start) { qend = start; count--; }
            int32_t mid = (qstart + qend) / 2;
            for (int32_t k = 0; k < count; k++) {
                il_push(result, mid);
                if (qstart != qend) mid++;
            }
            qstart = qend = -1;
        }
        p = ctg_next(c, &i, &j);
    }
    il_push(result, end);
}

char* np1o_snp_valid(const np1o_contig* in, const np1o_configure* cfg, int32_t* out_len) {   /* snpvalid.c:3-36 */
    g_updates = 0;
    if (in->length <= 0) { *out_len = 0; return (char*)calloc(1, 1); }
    octg* c = ctg_init(in, cfg);
    c->filter_kind = 0;
    int32_t max_span = stream_max_span(in);
    ilist kreg = get_region(c, 0, c->L - 1, cfg->min_len_inter_kmer, 0, FLAG_ZERO, brim_with_ext);
    if (kreg.n > 0) {
        merge_region(&kreg);
        create_insert_region(c, &kreg, max_span);
    }
    if (kreg.n > 0) {
        /* the reference keeps ONE list object: first the parts of ss_spilt_region, then -- its length reset to 0, its
         * contents left in place -- the values fts_spilt_region appends for the regions nothing spanned.  An odd number of
         * values makes the last pair take its end from whatever the list held at that index before. */
        ilist nodepth = split_region(c, &kreg, FLAG_ZERO, cfg->max_len_kmer);
        kreg.n = 0;
        kmer_correct(c, &nodepth, max_span, &kreg, 1);
        int32_t n_old = nodepth.n;
        nodepth.n = 0;
        if (kreg.n > 0) {
            for (int i = 0; i < kreg.n; i += 2) fts_split_region(c, kreg.v[i], kreg.v[i + 1], FLAG_ZERO, &nodepth);
            g_undefined = 0;
            if (nodepth.n & 1)       /* the value behind the last one: a leftover of the first list, or never written (zero-filled) */
                il_push(&nodepth, nodepth.n < n_old ? nodepth.v[nodepth.n] : 0);
            kmer_correct(c, &nodepth, max_span, NULL, 0);
            if (g_undefined) {
                free(nodepth.v); free(kreg.v); ctg_free(c);
                *out_len = -1;
                return NULL;
            }
        }
        free(nodepth.v);
    }
    free(kreg.v);
    char* out = get_contig(c, 0, c->L - 1, 0, out_len);
    ctg_free(c);
    return out;
}

/* ================= snp_phase (task 3; reference: source/lib/snpphase.c) =================
 * Two record streams: `sr` = the short-read BAM (contig->fp), `lr` = the long-read BAM (contig->tfp).  c->in says which of
 * them the record helpers look at, as the reference switches fp / read_fliter.  Region iteration is the same model as above:
 * overlap query (contig_next_iter with flag 2) or the swapped-interval "spanning" query (flag 1); the iterator re-use and the
 * saved offsets of contig.c:982-1043 only skip records that cannot qualify.
 * Where the reference's result rests on reading through a null / uninitialised pointer, g_undefined is set and
 * np1o_snp_phase returns NULL with *out_len = -1. */
#include <math.h>
#define FLAG_DEPTH 4
#define FLAG_SNP 8
#define FLAG_THIRD 16
#define FLAG_INSERT 32
#define FLAG_LEFT 64
#define FLAG_RIGHT 128
#define SNP_NUM 2
#define BASE_QUAL 41
#define READ_MAPQ 60

/* how often each stage of the last np1o_snp_phase call did something (so that tests can tell a fuzz run reached them):
 * 0 sites found, 1 sites kept, 2 sites with insertion columns, 3 low-count sites asked of the long reads, 4 long-read votes,
 * 5 sites settled to one base, 6 low-depth regions, 7 links counted, 8 sites re-written by the phase chain, 9 long-read links */
static int64_t g_sp_stats[10];
void np1o_snp_phase_stats(int64_t out[10]) { memcpy(out, g_sp_stats, sizeof(g_sp_stats)); }

typedef struct {   /* snpphase.h:8-16 */
    int32_t pos, left, right;
    int16_t length, total;
    uint8_t* region[SNP_NUM];
    kslist link;
} osnp;
typedef struct { osnp** v; int32_t n, cap; } snplist;

static osnp* snp_new(int32_t right, int32_t length) {   /* snpphase.c:3-13 */
    osnp* s = (osnp*)calloc(1, sizeof(osnp));
    for (int i = 0; i < SNP_NUM; i++) s->region[i] = (uint8_t*)calloc(1, (size_t)length + 8);
    s->right = right;
    s->length = (int16_t)length;
    return s;
}
static void snp_free(osnp* s) {
    if (!s) return;
    for (int i = 0; i < SNP_NUM; i++) free(s->region[i]);
    free(s->link.v);
    free(s);
}
static int32_t snp_get_index(osnp* s, const uint8_t* region) {   /* snpphase.c:40-48 */
    for (int i = 0; i < SNP_NUM; i++)
        if (memcmp(s->region[i], region, (size_t)s->length) == 0) return i;
    return -1;
}
static void kl_push(kslist* l, const okscore* k) {
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 16; l->v = (okscore*)realloc(l->v, (size_t)l->cap * sizeof(okscore)); }
    l->v[l->n++] = *k;
}
static int32_t kl_index_by_region(kslist* l, const okscore* item) {   /* seqlist_get_index + ks_compare_region: the LIST element's length decides */
    for (uint16_t i = 0; i < l->n; i++)
        if (memcmp(l->v[i].region, item->region, (size_t)l->v[i].length) == 0) return i;
    return -1;
}

static int32_t slot_nlargest(oslot* s, okmer** maxn, int32_t n) {   /* base.c:91-121 */
    int32_t count = 0;
    if (s->nk > 0) {
        okmer* p = s->k;
        maxn[0] = p;
        p++;
        count++;
        for (uint32_t i = 1; i < s->nk; i++, p++) {
            for (int32_t j = count - 1; j >= 0; j--) {
                if (p->count > maxn[j]->count) {
                    if (j < n - 1) maxn[j + 1] = maxn[j];
                    maxn[j] = p;
                } else {
                    if (j < n - 1) maxn[j + 1] = p;
                    break;
                }
            }
            if (count < n) count++;
        }
    }
    return count;
}
static void slot_merge_kmer(oslot* s) {   /* base.c:123-146 */
    int32_t count = 0;
    okmer* temp[16];
    if (s->nk > 0) {
        memset(temp, 0, sizeof(temp));
        okmer *p = s->k, *q = p;
        for (uint32_t i = 0; i < s->nk; i++, p++) {
            int32_t index = p->kmer & 0xf;
            if (temp[index] == 0) {
                count++;
                q->kmer = (uint16_t)index;
                q->count = p->count;
                temp[index] = q;
                q++;
            } else {
                temp[index]->count = (uint16_t)(temp[index]->count + p->count);
            }
        }
        s->nk = (uint32_t)count;
    }
}
static int32_t ks_nlargest(kslist* l, okscore** maxn, int32_t n) {   /* snpphase.c:873-903 */
    int32_t count = 0;
    okscore* p = l->v;
    maxn[0] = p;
    if (l->n > 0) {
        p++;
        count++;
        for (int i = 1; i < l->n; i++, p++) {
            for (int j = count - 1; j >= 0; j--) {
                if (ks_compare(p, maxn[j]) > 0) {
                    if (j < n - 1) maxn[j + 1] = maxn[j];
                    maxn[j] = p;
                } else {
                    if (j < n - 1) maxn[j + 1] = p;
                    break;
                }
            }
            if (count < n) count++;
        }
    }
    return count;
}

static int32_t check_snps(octg* c, int32_t count, double rate, int32_t flag) {   /* snpphase.c:205-214 */
    if (rate < c->cfg->min_snp_factor_sgs && flag) return 0;
    if (rate == 0 || (count >= c->cfg->min_count_snp && flag == 0 && rate < c->cfg->min_snp_factor_sgs)) return 2;
    return 1;
}

/* ts_find_snps (snpphase.c:136-203) */
static void sp_find_snps(octg* c, int32_t start, int32_t end, snplist* out) {
    int32_t i = start, j = 0, k, count, flag = 0, flag1 = 0, lasti = start, lastj = 0;
    oslot* p = &c->b[start].m;
    okmer* maxn[SNP_NUM];
    double rate;
    while (IN_RANGE(i, j, end)) {
        if (p->count == 0) p->flag |= FLAG_ZERO; else p->flag &= (uint8_t)~FLAG_ZERO;
        if (p->count <= c->cfg->min_depth_snp) p->flag |= FLAG_DEPTH; else p->flag &= (uint8_t)~FLAG_DEPTH;
        flag = 0;
        if (p->count > 0) {
            count = slot_nlargest(p, maxn, SNP_NUM);
            rate = count == 1 ? 0 : maxn[1]->count / (double)maxn[0]->count;
            flag = check_snps(c, p->count, rate, maxn[0]->kmer == p->base);
            if (flag == 2) {
                p->base = (uint8_t)maxn[0]->kmer;
            } else if (flag == 1) {
                if (j == 0 || (c->b[i].m.flag & FLAG_SNP) == 0) {
                    c->b[i].m.flag |= FLAG_SNP;
                    osnp* s = snp_new(c->L - 1, 1);
                    s->left = lasti;
                    s->pos = i;
                    flag1 = 1;
                    for (k = 0; k < count; k++) s->region[k][0] = (uint8_t)maxn[k]->kmer;
                    if (count < SNP_NUM) s->region[count][0] = p->base;
                    if (out->n == out->cap) { out->cap = out->cap ? out->cap * 2 : 64; out->v = (osnp**)realloc(out->v, (size_t)out->cap * sizeof(osnp*)); }
                    out->v[out->n++] = s;
                }
            }
        }
        if (flag != 1 && (c->b[i].m.flag & FLAG_SNP) == 0 && (c->b[i].ins == NULL || j == c->b[i].nins)) {
            lasti = i;
            if (flag1) {
                for (; lastj < out->n; lastj++) out->v[lastj]->right = lasti;
                flag1 = 0;
            }
        }
        p = ctg_next(c, &i, &j);
    }
}

static void update_flag(octg* c, int32_t start, int32_t end, uint8_t flag) {   /* contig.c:833-841 */
    int32_t i = start, j = 0;
    oslot* p = &c->b[start].m;
    while (IN_RANGE(i, j, end)) { p->flag |= flag; p = ctg_next(c, &i, &j); }
}
static void update_contig(octg* c, int32_t start, int32_t end, const uint8_t* region, uint16_t index) {   /* contig.c:811-821 */
    int32_t i = start, j = 0;
    oslot* p = &c->b[start].m;
    const uint8_t* q = region;
    while (i < end || (i == end && j == index)) {
        p->base = *q;
        p = ctg_next(c, &i, &j);
        q++;
    }
}
static void clean_region(octg* c, int32_t start, int32_t end) {   /* contig.c:620-630 */
    int32_t i = start, j = 0;
    oslot* p = &c->b[start].m;
    while (IN_RANGE(i, j, end)) { p->nk = 0; p->count = 0; p = ctg_next(c, &i, &j); }
}

/* ss_kmer_get_region in full (kmercount.c:332-363): anchors left / right, the length adjustment `flag`, the counter */
static void sp_get_region(octg* c, int64_t r, int32_t start, int32_t end, int32_t length, kslist* rd, okscore* ks, int32_t* count,
                          int32_t left, int32_t right, int32_t flag, int flagzero) {
    int32_t check = left != -1 ? 2 : 0;
    int32_t result = parse_read_kmer(c, r, start, end, ks, left, right, flagzero);
    if (ks->length == length && result >= check) {
        ks->length += flag;
        okscore* hit = NULL;
        for (uint16_t i = 0; i < rd->n; i++)
            if (memcmp(rd->v[i].region, ks->region, (size_t)rd->v[i].length) == 0) { hit = &rd->v[i]; break; }
        if (!hit) {
            ks->num = 1;
            kl_push(rd, ks);
            ks->region = NULL;
        } else {
            hit->num++;
            hit->mapqual += ks->mapqual;
            hit->qual += ks->qual;
        }
        if (count) (*count)++;
    } else {
        ks->mapqual = 0;
    }
}
static okscore* ks_new(int32_t length) {
    okscore* ks = (okscore*)calloc(1, sizeof(okscore));
    ks->region = (uint8_t*)calloc(1, (size_t)length + 8);
    return ks;
}

/* ts_fliter_snps (snpphase.c:216-349) */
static void sp_filter_snps(octg* c, snplist* sl, const np1o_contig* sr, const np1o_contig* lr, int32_t span_sr, int32_t span_lr) {
    if (sl->n <= 0) return;
    kslist rd = {0, 0, 0};
    okscore* ks = NULL;
    okscore* maxn[SNP_NUM];
    int32_t kept = 0;
    for (int32_t si = 0; si < sl->n && !g_undefined; si++) {
        osnp* q = sl->v[si];
        obase* bb = &c->b[q->pos];
        oslot* base = &bb->m;
        int32_t length = 1, start = q->pos, end = q->pos, total = 0, flag = 0, flag1, j;
        if (bb->ins && bb->nins > 0) {
            end = q->pos + 1;
            length += bb->nins + 1;
            g_sp_stats[2]++;
            c->in = sr;
            c->filter_kind = 0;
            int64_t r0 = lower_bound_pos(sr, start - span_sr), rstop = lower_bound_pos(sr, start);
            for (int64_t r = r0; r < rstop; r++) {
                if (!(read_endpos(sr, r) > end + 1)) continue;
                if (read_filter(c, r) == 2) {
                    if (!ks) ks = ks_new(length);
                    sp_get_region(c, r, start, end, length, &rd, ks, &total, -1, -1, -1, 0);
                    ks_clean(ks, length);
                }
            }
            flag = 1;
        } else {
            ks = ks_new(length);
            total = base->count;
        }
        if (total <= c->cfg->min_count_snp) {
            if (!ks) { g_undefined = 1; break; }   /* snpphase.c:269: memset through the null KmerScore */
            g_sp_stats[3]++;
            if (length == 1) {
                for (uint32_t t = 0; t < base->nk; t++) {
                    ks->region[0] = (uint8_t)base->k[t].kmer;
                    ks->num = base->k[t].count;
                    ks->mapqual = READ_MAPQ * ks->num;
                    ks->qual = BASE_QUAL * ks->num;
                    kl_push(&rd, ks);
                    ks->region = (uint8_t*)calloc(1, (size_t)length + 8);
                }
            }
            memset(ks->region, BASE_DEL, (size_t)length);
            flag1 = kl_index_by_region(&rd, ks);
            c->in = lr;
            c->filter_kind = 2;
            int64_t r0 = lower_bound_pos(lr, start - span_lr), rstop = lower_bound_pos(lr, start);
            for (int64_t r = r0; r < rstop; r++) {
                if (!(read_endpos(lr, r) > end + 1)) continue;
                if (read_filter(c, r) == 1) {
                    { int32_t before = total; sp_get_region(c, r, start, end, length, &rd, ks, &total, q->left, q->right, -1, 1); g_sp_stats[4] += total - before; }
                    ks_clean(ks, length);
                }
            }
            flag = 1;
            if (flag1 == -1) {
                memset(ks->region, BASE_DEL, (size_t)length);
                flag1 = kl_index_by_region(&rd, ks);
                if (flag1 != -1) {
                    free(rd.v[flag1].region);
                    for (int32_t t = flag1; t + 1 < rd.n; t++) rd.v[t] = rd.v[t + 1];
                    rd.n--;
                }
            }
        }
        if (flag) {
            if (rd.n == 0) { g_undefined = 1; break; }   /* ts_get_nlargest on an empty list: maxn[1] is never set */
            flag1 = ks_nlargest(&rd, maxn, SNP_NUM);
            double rate = flag1 == 1 ? 0 : maxn[1]->num / (double)maxn[0]->num;
            memset(ks->region, BASE_DEL, (size_t)length);
            ks->region[0] = base->base;
            flag = check_snps(c, total, rate, memcmp(maxn[0]->region, ks->region, (size_t)maxn[0]->length) == 0);
            if (flag == 1) {
                if (length != q->length)
                    for (int t = 0; t < SNP_NUM; t++) q->region[t] = (uint8_t*)realloc(q->region[t], (size_t)length + 8);
                q->length = (int16_t)maxn[0]->length;
                for (j = 0; j < flag1; j++) memcpy(q->region[j], maxn[j]->region, (size_t)q->length);
                if (flag1 < SNP_NUM) memcpy(q->region[flag1], ks->region, (size_t)q->length);
                { sl->v[si] = NULL; sl->v[kept++] = q; }
            } else {
                if (flag == 2) {
                    g_sp_stats[5]++;
                    base->base = maxn[0]->region[0];
                    update_contig(c, start, end, maxn[0]->region, (uint16_t)-1);
                }
                c->b[start].m.flag &= 0xf7;
                snp_free(q);
                sl->v[si] = NULL;
            }
        } else {
            { sl->v[si] = NULL; sl->v[kept++] = q; }
        }
        if (ks) { free(ks->region); free(ks); }
        for (int32_t t = 0; t < rd.n; t++) free(rd.v[t].region);
        rd.n = 0;
        ks = NULL;
    }
    if (!g_undefined) sl->n = kept;
    free(rd.v);
}

/* ts_region_correct (snpphase.c:843-871) */
static void sp_region_correct(octg* c, int32_t start, int32_t end) {
    oslot* base = &c->b[end].m;
    oscore* score = slot_max_score(base);
    int32_t i = end, j = 0;
    okmer* maxn[2];
    while (i > start || (i == start && j == 0)) {
        if (!score) { g_undefined = 1; return; }
        if ((base->flag & FLAG_ZERO) || (j == 0 && score->base != BASE_DEL)) base->base = score->base;
        slot_merge_kmer(base);
        if (base->nk >= 2) {
            slot_nlargest(base, maxn, 2);
            double rate = maxn[1]->count / (double)maxn[0]->count;
            if (maxn[0]->kmer != base->base || rate > c->cfg->max_indel_factor_lgs) {
                if (base->base == BASE_DEL || j != 0 || maxn[0]->kmer != base->base || rate > c->cfg->max_snp_factor_lgs) base->flag |= FLAG_THIRD;
                else base->flag &= (uint8_t)~FLAG_THIRD;
            }
        }
        base = ctg_prev(c, &i, &j);
        score = slot_get_score(base, score->kmer >> 4);
    }
}

/* ts_correct_lower_depth (snpphase.c:797-841) */
static void sp_correct_lower_depth(octg* c, ilist* nodepth, const np1o_contig* sr, const np1o_contig* lr, int32_t span_sr, int32_t span_lr) {
    c->in = sr;
    c->filter_kind = 0;
    c->shift = 4;
    for (int i = 0; i < nodepth->n; i += 2) {
        int32_t s = nodepth->v[i], e = nodepth->v[i + 1];
        clean_region(c, s, e);
        as_read(c, s, e);
        parse_region(c, s, e, 2, span_sr);
    }
    c->in = lr;
    c->filter_kind = 2;
    for (int i = 0; i < nodepth->n; i += 2) parse_region(c, nodepth->v[i], nodepth->v[i + 1], 1, span_lr);
    for (int i = 0; i < nodepth->n && !g_undefined; i += 2) {
        region_score(c, nodepth->v[i], nodepth->v[i + 1], c->cfg->indel_balance_factor_lgs);
        sp_region_correct(c, nodepth->v[i], nodepth->v[i + 1]);
    }
}

/* ts_find_snp_region (snpphase.c:559-613) */
static ilist sp_find_snp_region(octg* c, snplist* sl, int32_t gap, uint32_t flag) {
    ilist result = {0, 0, 0};
    int32_t temp;
    uint32_t flag1 = FLAG_LEFT | FLAG_RIGHT, flag2;
    osnp *qstart = NULL, *qend = NULL;
    for (int32_t i = 0; i < sl->n; i++) {
        osnp* p = sl->v[i];
        flag2 = c->b[p->pos].m.flag;
        if ((flag2 & flag) || (flag2 & flag1)) {
            if (qstart == NULL) {
                qend = qstart = p;
            } else if (flag || (flag2 & FLAG_RIGHT)) {
                if (flag) temp = p->pos - qend->pos;
                else temp = p->right - qend->left;
                if (temp < gap) {
                    qend = p;
                } else {
                    if (qstart != qend) {
                        if (flag) {
                            il_push(&result, qstart->pos);
                            il_push(&result, qend->pos + 1);
                        } else {
                            il_push(&result, qstart->left);
                            il_push(&result, qend->right);
                        }
                    }
                    if (flag || (flag2 & FLAG_LEFT)) qend = qstart = p;
                    else qend = qstart = NULL;
                }
            }
        }
    }
    if (qstart && qstart != qend) {
        if (flag) {
            il_push(&result, qstart->pos);
            il_push(&result, qend->pos);
        } else {
            il_push(&result, qstart->left);
            il_push(&result, qend->right);
        }
    }
    return result;
}

/* ts_snps_parse_read (snpphase.c:615-776): the haplotype strings of one record at the marked columns, appended to `ld`; their
 * bytes live one behind the other in the buffer ks->region points into */
static void sp_parse_read(octg* c, int64_t r, int32_t start, int32_t end, uint32_t flagbrim, kslist* ld, okscore* ks) {
    const np1o_contig* in = c->in;
    if (!in->n_cigar[r]) return;
    int32_t pos = in->pos[r], qpos = 0, qstart, qend, i, j, k, len, del = 0, curpos = 0, sign = 0, comfirmindex = 0,
            totallength = c->cfg->max_variant_count_lgs;
    uint8_t curcigar, lastcigar = CINS, *q = ks->region;
    const uint32_t* cg = in->cigar + in->cigar_off[r];
    const uint8_t* seq = in->seq + in->seq_off[r];
    const uint8_t* qual = in->qual + in->qual_off[r];
    const uint16_t flag = FLAG_LEFT | FLAG_RIGHT;
    cut_read(c, r, &qstart, &qend);
    ks->mapqual = in->mapq[r];
    for (i = 0; i < in->n_cigar[r]; ++i) {
        len = OPLEN(cg[i]);
        curcigar = OP(cg[i]);
        if (totallength - len < 0) break;
        switch (curcigar) {
            case CMATCH: case CDEL:
                for (j = 0; j < len; j++, pos++) {
                    if (pos >= start && pos <= end && qpos >= qstart && qpos <= qend) {
                        oslot* base = &c->b[pos].m;
                        if (lastcigar != CINS && pos > start && (qpos > qstart || (qpos == qstart && lastcigar == CDEL))) {
                            obase* pb = &c->b[pos - 1];
                            if (pb->ins != NULL && curpos) {
                                for (k = 0; k < pb->nins; k++) {
                                    ks->region[ks->length++] = BASE_DEL;
                                    totallength--;
                                    del++;
                                }
                            }
                        }
                        if (flagbrim == 0 || (base->flag & flag)) {
                            if (base->flag & FLAG_SNP) {
                                if (curpos == 0) {
                                    ks->region = q;
                                    ks->length = 0;
                                    ks->num = pos;
                                    ks->qual = 0;
                                    del = 0;
                                    curpos = 1;
                                    if (flagbrim == 0) sign = 1;
                                } else {
                                    sign++;
                                }
                            } else if (flagbrim) {
                                if (base->base == seqi(seq, qpos)) sign++;
                            } else {
                                sign++;
                            }
                            if (curpos) {
                                if (curcigar == CDEL) {
                                    ks->region[ks->length++] = BASE_DEL;
                                } else {
                                    ks->region[ks->length++] = seqi(seq, qpos);
                                    ks->qual += qual[qpos];
                                }
                                totallength--;
                                if (ks->num != pos || c->b[pos].ins == NULL) {
                                    if (ks->num != pos) {
                                        if (ks->length != del) ks->qual /= (double)(ks->length - del);
                                        else ks->qual = 0;
                                        ks->length--;
                                    }
                                    kl_push(ld, ks);
                                    q += ks->length;
                                    curpos = 0;
                                }
                            }
                            if (ks->num != pos) {
                                if (base->flag & FLAG_SNP) {
                                    ks->region = q;
                                    ks->length = 1;
                                    ks->num = pos;
                                    ks->qual = qual[qpos];
                                    del = 0;
                                    curpos = 1;
                                    if (flagbrim == 0) {
                                        comfirmindex++;
                                        sign = 1;
                                    }
                                } else if (base->flag & FLAG_RIGHT) {
                                    if (sign == 2) {
                                        comfirmindex = ld->n;
                                    } else if (comfirmindex >= 0 && comfirmindex < ld->n) {
                                        for (k = comfirmindex; k < ld->n; k++, comfirmindex++) ld->v[k].length = 0;
                                    }
                                    curpos = 0;
                                    sign = 0;
                                    if (base->flag & FLAG_LEFT) sign++;
                                }
                            }
                        }
                    }
                    if (curcigar != CDEL) qpos++;
                    lastcigar = curcigar;
                }
                break;
            case CINS:
                if (curpos) {
                    if (pos) {
                        obase* pb = (pos - 1 < c->L) ? &c->b[pos - 1] : NULL;
                        for (j = 0; j < len; j++, qpos++) {
                            if (pos > start && pos <= end && qpos >= qstart && qpos <= qend) {
                                ks->region[ks->length++] = seqi(seq, qpos);
                                ks->qual += qual[qpos];
                                totallength--;
                            }
                        }
                        if (pos > start && pos <= end && qpos > qstart && qpos <= qend + 1) {
                            if (!pb || pb->ins == NULL) { g_undefined = 1; return; }   /* snpphase.c:750: p->length of a null list */
                            for (; j < pb->nins; j++) {
                                ks->region[ks->length++] = BASE_DEL;
                                totallength--;
                                del++;
                            }
                        }
                    } else {
                        qpos += len;
                        qstart += len;
                        lastcigar = curcigar;
                    }
                } else {
                    qpos += len;
                }
                lastcigar = curcigar;
                break;
            case CHARD: case CSOFT:
                qpos += len;
                break;
        }
    }
}

static int32_t sp_list_find(snplist* sl, int32_t pos) {   /* snpphase.c:68-85 */
    int32_t i = 0, j = sl->n - 1, mid, qpos;
    while (i <= j) {
        mid = (i + j) / 2;
        qpos = sl->v[mid]->pos;
        if (qpos == pos) return mid;
        if (qpos < pos) i = mid + 1; else j = mid - 1;
    }
    return -1;
}

/* ts_tranfer_link (snpphase.c:423-448) */
static void sp_transfer_link(osnp** snp, okscore** ks, int16_t* total, int32_t flag) {
    if (ks[0]->length == snp[0]->length && ks[1]->length == snp[1]->length && flag == 4) {
        int32_t index = snp_get_index(snp[0], ks[0]->region);
        if (index != -1) {
            index++;
            ks[1]->length = index << 4;
            index = snp_get_index(snp[1], ks[1]->region);
            if (index != -1) {
                index++;
                ks[1]->length += index;
                okscore* p = NULL;
                for (uint16_t i = 0; i < snp[1]->link.n; i++)
                    if (snp[1]->link.v[i].length == ks[1]->length) { p = &snp[1]->link.v[i]; break; }
                if (p == NULL) {
                    ks[1]->num = 1;
                    kl_push(&snp[1]->link, ks[1]);
                } else {
                    p->num++;
                    p->mapqual += ks[1]->mapqual;
                    p->qual += ks[1]->qual;
                }
                (*total)++;
                g_sp_stats[7]++;
            }
        }
    }
}

/* ts_snps_deal_linkdata (snpphase.c:778-795) */
static void sp_deal_linkdata(octg* c, kslist* ld, snplist* sl, uint32_t flag) {
    if (ld->n > 1) {
        okscore* ks[2];
        osnp* snps[2];
        for (int32_t i = 1; i < ld->n; i++) {
            okscore* p = &ld->v[i];
            if (p->length && (p - 1)->length &&
                (flag == 0 || ((c->b[p->num].m.flag & FLAG_RIGHT) && (c->b[(p - 1)->num].m.flag & FLAG_LEFT)))) {
                int32_t index = sp_list_find(sl, p->num);
                if (index < 1) { g_undefined = 1; return; }   /* snpslist->data[index - 1] in front of the array */
                snps[0] = sl->v[index - 1];
                snps[1] = sl->v[index];
                ks[0] = p - 1;
                ks[1] = p;
                sp_transfer_link(snps, ks, &snps[1]->total, 4);
            }
        }
    }
}

/* ts_find_snps_link (snpphase.c:351-421) */
static void sp_find_snps_link(octg* c, snplist* sl, const np1o_contig* sr, const np1o_contig* lr, int32_t span_sr, int32_t span_lr) {
    if (sl->n <= 1) return;
    ilist reg = sp_find_snp_region(c, sl, c->cfg->read_len, FLAG_SNP);
    kslist ld = {0, 0, 0};
    okscore* ks = ks_new(c->cfg->max_variant_count_lgs > 0 ? c->cfg->max_variant_count_lgs : 1);
    uint8_t* pks = ks->region;
    c->in = sr;
    c->filter_kind = 0;
    for (int i = 0; i < reg.n && !g_undefined; i += 2) {
        int32_t s = reg.v[i], e = reg.v[i + 1];
        for (int64_t r = lower_bound_pos(sr, s - span_sr); r < sr->n_reads && !g_undefined; r++) {
            if (sr->pos[r] >= e + 1) break;
            if (read_endpos(sr, r) <= s) continue;
            if (read_filter(c, r) == 2) {
                sp_parse_read(c, r, s, e, 0, &ld, ks);
                if (!g_undefined) sp_deal_linkdata(c, &ld, sl, 0);
                ld.n = 0;
                ks->length = 0;
                ks->region = pks;
            }
        }
    }
    for (int32_t i = 1; i < sl->n; i++) {
        if (sl->v[i]->total <= c->cfg->min_count_snp_link) {
            osnp *a = sl->v[i - 1], *b = sl->v[i];
            c->b[a->left].m.flag |= FLAG_LEFT;
            c->b[a->pos].m.flag |= FLAG_LEFT;
            c->b[a->right].m.flag |= FLAG_RIGHT;
            c->b[b->left].m.flag |= FLAG_LEFT;
            c->b[b->pos].m.flag |= FLAG_RIGHT;
            c->b[b->right].m.flag |= FLAG_RIGHT;
        }
    }
    free(reg.v);
    reg = sp_find_snp_region(c, sl, c->cfg->max_variant_count_lgs, 0);
    c->in = lr;
    c->filter_kind = 2;
    for (int i = 0; i < reg.n && !g_undefined; i += 2) {
        int32_t s = reg.v[i], e = reg.v[i + 1];
        for (int64_t r = lower_bound_pos(lr, s - span_lr); r < lr->n_reads && !g_undefined; r++) {
            if (lr->pos[r] >= e + 1) break;
            if (read_endpos(lr, r) <= s) continue;
            if (read_filter(c, r) == 1) {
                sp_parse_read(c, r, s, e, 1, &ld, ks);
                { int64_t before = g_sp_stats[7]; if (!g_undefined) sp_deal_linkdata(c, &ld, sl, 1); g_sp_stats[9] += g_sp_stats[7] - before; }
                ld.n = 0;
                ks->length = 0;
                ks->region = pks;
            }
        }
    }
    free(reg.v);
    free(ld.v);
    free(pks);
    free(ks);
}

/* ts_snps_score (snpphase.c:450-516): chain score over the links between neighbouring sites; evaluation order as written */
static void sp_snps_score(octg* c, snplist* sl) {
    if (sl->n <= 1) return;
    oslot* base[SNP_NUM];
    uint16_t link[2][SNP_NUM + 1], temp;
    double score;
    int32_t i, j, k;
    base[0] = &c->b[sl->v[0]->pos].m;
    base[0]->ns = 0;
    for (i = 1; i <= SNP_NUM; i++) slot_add_score(base[0], (uint16_t)i, 0);
    for (i = 1; i < sl->n; i++) {
        osnp* q = sl->v[i];
        base[0] = &c->b[sl->v[i - 1]->pos].m;
        base[1] = &c->b[q->pos].m;
        base[1]->ns = 0;
        if (q->link.n) {
            memset(link, 0, sizeof(link));
            for (j = 0; j < q->link.n; j++) {
                okscore* ks = &q->link.v[j];
                temp = (uint16_t)(ks->length >> 4);
                oscore* ps0 = slot_get_score(base[0], temp);
                if (!ps0) { g_undefined = 1; return; }
                score = ps0->score;
                score += ks->num * log10((ks->mapqual + ks->qual) / (double)ks->num + 2) - q->total / c->cfg->ploidy;
                oscore* pscore = slot_get_score(base[1], (uint16_t)ks->length);
                if (pscore == NULL || pscore->score < score) {
                    if (link[0][temp]) {
                        if (slot_get_score(base[1], link[0][temp])->score >= score) continue;
                        link[1][link[0][temp]] = 0;
                    }
                    if (pscore != NULL) link[0][pscore->kmer >> 4] = 0;
                    slot_add_score(base[1], (uint16_t)ks->length, score);
                    link[0][temp] = ks->length & 0xf;
                    link[1][ks->length & 0xf] = temp;
                }
            }
            k = 1;
            for (j = 1; j <= SNP_NUM; j++) {
                if (link[1][j] == 0) {
                    for (; k <= SNP_NUM; k++) {
                        if (link[0][k] == 0) {
                            temp = (uint16_t)((k << 4) + j);
                            oscore* ps0 = slot_get_score(base[0], (uint16_t)k);
                            if (!ps0) { g_undefined = 1; return; }
                            score = ps0->score;
                            score -= q->total / c->cfg->ploidy;
                            slot_add_score(base[1], temp, score);
                            break;
                        }
                    }
                }
            }
        } else {
            for (j = 1; j <= SNP_NUM; j++) slot_add_score(base[1], (uint16_t)j, 0);
        }
    }
}

/* ts_snps_correct (snpphase.c:518-557) */
static void sp_snps_correct(octg* c, snplist* sl) {
    if (sl->n <= 1) return;
    oscore* score = NULL;
    for (int32_t i = sl->n - 1; i > 0; i--) {
        osnp* q = sl->v[i];
        if (q->link.n > 0) {
            oslot* base;
            if (score == NULL) {
                base = &c->b[q->pos].m;
                score = slot_max_score(base);
                if (!score) { g_undefined = 1; return; }
                if (q->length == 1) base->base = q->region[score->base - 1][0];
                else update_contig(c, q->pos, q->pos + 1, q->region[score->base - 1], (uint16_t)-1);
            }
            int32_t index = (score->kmer >> 4) - 1;
            if (index < 0 || index >= SNP_NUM) { g_undefined = 1; return; }
            q = sl->v[i - 1];
            base = &c->b[q->pos].m;
            g_sp_stats[8]++;
            if (q->length == 1) base->base = q->region[index][0];
            else update_contig(c, q->pos, q->pos + 1, q->region[index], (uint16_t)-1);
            if (q->link.n > 0) {
                score = slot_get_score(base, (uint16_t)(index + 1));
                if (!score) { g_undefined = 1; return; }
            } else {
                score = NULL;
            }
        }
    }
}

/* snp_phase (snpphase.c:87-134) */
char* np1o_snp_phase(const np1o_contig* sr, const np1o_contig* lr, const np1o_configure* cfg, int32_t* out_len) {
    g_updates = 0;
    g_undefined = 0;
    memset(g_sp_stats, 0, sizeof(g_sp_stats));
    if (sr->length <= 0) { *out_len = 0; return (char*)calloc(1, 1); }
    octg* c = ctg_init(sr, cfg);
    int32_t span_sr = stream_max_span(sr), span_lr = stream_max_span(lr);
    c->filter_kind = 0;
    create_insert(c, 0, c->L - 1, span_sr);
    c->shift = 16;
    parse_region(c, 0, c->L - 1, 2, span_sr);
    c->shift = 4;
    snplist sl = {0, 0, 0};
    sp_find_snps(c, 0, c->L - 1, &sl);
    g_sp_stats[0] = sl.n;
    ilist nodepth = get_region(c, 0, c->L - 1, cfg->ext_len_edge, 0, FLAG_DEPTH, brim_no_ext);
    if (nodepth.n > 0) {
        merge_region(&nodepth);
        if (getenv("NP1O_SP_DEBUG")) { fprintf(stderr, "regions:"); for (int i = 0; i < nodepth.n; i += 2) fprintf(stderr, " [%d,%d]", nodepth.v[i], nodepth.v[i + 1]); fprintf(stderr, "\n"); }
        for (int i = 0; i < nodepth.n; i += 2) update_flag(c, nodepth.v[i], nodepth.v[i + 1], FLAG_INSERT);
    }
    c->in = lr;
    c->filter_kind = 2;
    c->insflag = FLAG_INSERT | FLAG_SNP;
    create_insert(c, 0, c->L - 1, span_lr);
    c->insflag = 0;
    const char* dbg_stop = getenv("NP1O_SP_STOP");
    const int stop = dbg_stop ? atoi(dbg_stop) : 99;
    if (stop >= 2) sp_filter_snps(c, &sl, sr, lr, span_sr, span_lr);
    g_sp_stats[1] = g_undefined ? 0 : sl.n;
    g_sp_stats[6] = nodepth.n / 2;
    if (stop >= 3 && !g_undefined && nodepth.n > 0) sp_correct_lower_depth(c, &nodepth, sr, lr, span_sr, span_lr);
    free(nodepth.v);
    if (stop >= 4 && !g_undefined && sl.n > 1) {
        sp_find_snps_link(c, &sl, sr, lr, span_sr, span_lr);
        if (!g_undefined) sp_snps_score(c, &sl);
        if (!g_undefined) sp_snps_correct(c, &sl);
    }
    for (int32_t i = 0; i < sl.n; i++) snp_free(sl.v[i]);
    free(sl.v);
    char* out = NULL;
    if (g_undefined) *out_len = -1;
    else out = get_contig(c, 0, c->L - 1, FLAG_THIRD, out_len);
    c->in = sr;
    ctg_free(c);
    return out;
}
