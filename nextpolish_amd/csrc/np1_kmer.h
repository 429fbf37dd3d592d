// kmer_count (task 2) bodies, host+device (reference: source/lib/kmercount.c:93-465 and the region helpers
// source/lib/contig.c:182-200,498-620,706-734,801-821).  kmer_count touches only the lowercase (FLAG_ZERO)
// neighbourhoods of a contig -- a fraction of a percent of the draft -- so the work is small and irregular:
// the GPU mapping is "one lane per region", every lane running the reference's per-region logic on the
// HBM-resident record stream (np1_kernels.hip: k_kc_*).  The same bodies compile for the host model the CPU
// test-suite checks against the oracle (tests/model).
#pragma once
#include <string.h>
#include "np1_core.h"

namespace np1k {

constexpr uint32_t KC_FLAG_ZERO = 1, KC_FLAG_COVERAGE = 2;
constexpr uint32_t ERR_KC_POOL = 64, ERR_KC_REGIONS = 128, ERR_KC_INCONSISTENT = 256, ERR_KC_UNDEFINED = 512;

// everything a region lane needs (plain pointers; device or host memory)
struct KcCtx {
    ReadsDev R;
    const uint8_t* mapq;
    const int32_t* isize;
    const uint64_t* qual_off;
    const uint8_t* qual;
    const uint8_t* level;        // per record: contig_read_fliter level 0/1/2
    const int32_t* endpos;       // per record: htslib bam_endpos
    const uint32_t* ctg_off;
    const uint64_t* read_begin;
    const uint8_t* draft_code;   // per draft base: nt16 code of the input draft (region finding, before insertion columns)
    const uint8_t* draft_flag;   // per draft base: FLAG_ZERO for lowercase input
    // slot space (after the insertion columns of the regions are known)
    const uint32_t* soff;
    uint8_t* sbase;
    uint8_t* sflag;
    uint16_t* srefk;
    uint16_t* scount;
    // per-slot context lists for the no-depth regions: singly linked, 2 words per entry {kmer | count<<16, next+1}
    uint32_t* lhead;
    uint32_t* lpool;
    uint32_t lcap;
    uint32_t* lcount;
    // DP state scratch: 16 states per slot of the region being solved, bump allocated
    long long* st_score;
    uint16_t* st_kmer;
    uint8_t* st_rank;            // 0xff = state absent
    uint32_t st_cap;             // in slots
    uint32_t* st_count;
    // candidate haplotypes of kmer_correct, bump allocated bytes
    uint8_t* hpool;
    uint32_t hcap;
    uint32_t* hcount;
    // parameters (reference: Configure)
    int32_t trim, ext_len_edge, min_len_ldr, min_len_inter_kmer, max_len_kmer, max_count_kmer, min_map_quality, read_tlen;
    double max_clip_ratio_sgs, min_count_ratio_skip;
    int K;                       // < 0: indel_balance_factor_sgs is no dyadic fraction, scores are doubles (rate below)
    long long Rfix;
    double rate;
    int32_t max_span;            // longest reference span of any record (lower bound for overlap scans)
    int32_t keep_zero_marks;     // snp_valid: a covering read leaves the FLAG_ZERO marks alone (ss_parse_read_kmer with flagzero = 1)
    // snp_phase (np1_phase.h): the region chain ends in ts_region_correct instead of contig_region_correct
    int32_t third_rule;
    double max_indel_factor_lgs, max_snp_factor_lgs;
    const uint32_t* sown;        // per slot: global index of the base it belongs to
    const uint16_t* bmark;       // per base: marks of its main slot | 0x100 when it owns insertion columns (link walk of snp_phase)
    const unsigned long long* bbits;   // one bit per base: the link walk has to look at it (a site / an anchor mark), 64 bases per word
    uint32_t* err;
};

NP1_HD uint32_t kc_bump(uint32_t* counter, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(counter, n);
#else
    uint32_t o = *counter;
    *counter += n;
    return o;
#endif
}

// ---- record helpers -----------------------------------------------------------------------------
// contig_read_cliprate + contig_read_fliter (contig.c:632-665)
NP1_HD uint32_t kc_filter_level(const ReadsDev& R, int64_t r, const uint8_t* mapq, const int32_t* isize, int32_t read_tlen,
                                double max_clip, int32_t min_mapq) {
    if ((R.flag[r] & 0xC04) != 0) return 0;
    const uint32_t ncig = R.n_cigar[r];
    double cliprate = 0;
    if (ncig > 0) {   // (the reference reads out of bounds for CIGAR-less records; they never vote anyway)
        const uint32_t* cg = R.cigar + R.cigar_off[r];
        int32_t addlen = 0;
        if (cig_op(cg[0]) == 4) addlen += cig_len(cg[0]);
        if (cig_op(cg[ncig - 1]) == 4) addlen += cig_len(cg[ncig - 1]);
        cliprate = R.l_qseq[r] > 0 ? addlen / (double)R.l_qseq[r] : 0;
    }
    const int32_t length = isize[r] >= 0 ? isize[r] : -isize[r];
    uint32_t result = 0;
    if ((length > 0 && length < read_tlen) || cliprate < max_clip) {
        result = 1;
        if ((int32_t)mapq[r] >= min_mapq && (cliprate < max_clip + 0.05)) result = 2;
    }
    return result;
}
NP1_HD int32_t kc_endpos(const ReadsDev& R, int64_t r) {   // htslib bam_endpos
    if (!(R.flag[r] & 4) && R.n_cigar[r] > 0) {
        const uint32_t* cg = R.cigar + R.cigar_off[r];
        int32_t l = 0;
        for (uint32_t k = 0; k < R.n_cigar[r]; ++k) {
            const uint32_t op = cig_op(cg[k]);
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) l += cig_len(cg[k]);
        }
        return R.pos[r] + (l > 0 ? l : 1);
    }
    return R.pos[r] + 1;
}
NP1_HD int64_t kc_lower_bound_pos(const ReadsDev& R, int64_t lo, int64_t hi, int32_t p) {   // first r in [lo,hi) with pos >= p
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (R.pos[mid] < p) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// trimmed query window of one record (contig.c:333-358), same bounded form as prep_record
NP1_HD void kc_cut_read(const ReadsDev& R, int64_t r, int trim, int32_t* qs_out, int32_t* qe_out) {
    const uint32_t ncig = R.n_cigar[r];
    const uint32_t* cg = R.cigar + R.cigar_off[r];
    const uint8_t* seq = R.seq + R.seq_off[r];
    const int32_t lq = R.l_qseq[r];
    int32_t qs = trim + (cig_op(cg[0]) == 4 ? cig_len(cg[0]) : 0);
    int32_t qe = lq - trim - (cig_op(cg[ncig - 1]) == 4 ? cig_len(cg[ncig - 1]) : 0) - 1;
    if (trim > 0) {
        bool dead = false;
        for (;;) {
            if (qs >= lq) { dead = true; break; }
            if (seq_nib(seq, qs) != seq_nib(seq, qs - 1)) break;
            ++qs;
        }
        while (!dead) {
            if (qe < 0 || qe + 1 >= lq) { dead = qe < qs; break; }
            if (seq_nib(seq, qe) != seq_nib(seq, qe + 1)) break;
            --qe;
        }
        if (dead) { qs = 1; qe = 0; }
    }
    if (qs > qe) { qs = 1; qe = 0; }
    *qs_out = qs;
    *qe_out = qe;
}

// ---- region-restricted pass-2 walk (contig.c:247-331 / kmercount.c:365-465 with a [start,end] window) ---------
// sink.vote(pos, col, sym, qpos, pad): col 0 = the base slot of local position pos, col j>0 = insertion column j-1
// after it; qpos = query index of a real base (-1 for DEL); pad = DEL written into an unused insertion column
template <class Sink>
NP1_HD void kc_walk(const KcCtx& c, int64_t r, uint32_t g0, int32_t start, int32_t end, Sink& sink) {
    const uint32_t ncig = c.R.n_cigar[r];
    if (!ncig) return;
    const uint32_t* cg = c.R.cigar + c.R.cigar_off[r];
    const uint8_t* seq = c.R.seq + c.R.seq_off[r];
    int32_t qs, qe;
    kc_cut_read(c.R, r, c.trim, &qs, &qe);
    int32_t pos = c.R.pos[r], qpos = 0;
    uint32_t last = 1;
    for (uint32_t i = 0; i < ncig; ++i) {
        const uint32_t op = cig_op(cg[i]);
        const int32_t len = cig_len(cg[i]);
        if (op == 0 || op == 2) {
            for (int32_t j = 0; j < len; ++j, ++pos) {
                if (pos >= start && pos <= end && qpos >= qs && qpos <= qe) {
                    if (last != 1 && pos > start && (qpos > qs || (qpos == qs && last == 2))) {
                        const uint32_t n = c.soff[g0 + pos] - c.soff[g0 + pos - 1] - 1;
                        for (uint32_t k = 0; k < n; ++k) sink.vote(pos - 1, k + 1, 3u, -1, true);
                    }
                    if (op == 2) sink.vote(pos, 0, 3u, -1, false);
                    else sink.vote(pos, 0, seq_nib(seq, qpos), qpos, false);
                }
                if (op != 2) ++qpos;
                last = op;
            }
        } else if (op == 1) {
            if (pos != 0) {
                const bool inr = pos > start && pos <= end;
                for (int32_t j = 0; j < len; ++j, ++qpos)
                    if (inr && qpos >= qs && qpos <= qe) sink.vote(pos - 1, (uint32_t)j + 1, seq_nib(seq, qpos), qpos, false);
                if (inr && qpos > qs && qpos <= qe + 1) {
                    const uint32_t n = c.soff[g0 + pos] - c.soff[g0 + pos - 1] - 1;
                    for (uint32_t j = (uint32_t)len; j < n; ++j) sink.vote(pos - 1, j + 1, 3u, -1, true);
                }
                last = 1;
            } else {
                qpos += len;
                qs += len;
                last = 1;
            }
        } else if (op == 4 || op == 5) {
            qpos += len;
        }
        if (pos > end) break;
    }
}

// ---- pass 1 restricted to a region (contig.c:182-245, flag argument 0): insertion columns ---------------------
NP1_HD void kc_insert_region(const KcCtx& c, uint32_t ctg, int32_t start, int32_t end, uint32_t* ins) {
    const uint32_t g0 = c.ctg_off[ctg];
    const int64_t rb = (int64_t)c.read_begin[ctg], re = (int64_t)c.read_begin[ctg + 1];
    for (int64_t r = kc_lower_bound_pos(c.R, rb, re, start - c.max_span); r < re; ++r) {
        if (c.R.pos[r] >= end + 1) break;
        if (c.endpos[r] <= start || c.level[r] < 1 || c.R.n_cigar[r] == 0) continue;
        const uint32_t* cg = c.R.cigar + c.R.cigar_off[r];
        int32_t pos = c.R.pos[r];
        for (uint32_t i = 0; i < c.R.n_cigar[r]; ++i) {
            const uint32_t op = cig_op(cg[i]);
            if (op == 0 || op == 2) pos += cig_len(cg[i]);
            else if (op == 1 && pos > start && pos <= end) np1_atomic_max(&ins[g0 + (uint32_t)pos - 1], (uint32_t)cig_len(cg[i]));
        }
    }
}

// ---- low-quality regions on the input draft (contig.c:498-620), before any insertion column exists -------------
// `flagged` = ascending local positions of the contig's lowercase bases.  Exact sparse restatement of the dense
// walk of contig_get_region(0, L-1, gap, con, FLAG_ZERO, brim).
NP1_HD void kc_brim(const uint8_t* code, const uint8_t* flag, int32_t ext, bool with_ext, int32_t bstart, int32_t bend,
                    int32_t* start, int32_t* end) {
    *start = *start >= bstart + ext ? *start - ext : bstart;     // contig_brim_no_extension
    *end = *end <= bend - ext ? *end + ext : bend;
    if (!with_ext) return;
    int32_t p = *start + 1;                                       // contig_brim_with_extension
    // (ext_len_edge = 0 and a region that starts on the last base: the reference reads data[L], contig.c:507-508 -- not here)
    while (*start > bstart && p <= bend && (code[p] == code[p - 1] || (flag[p - 1] & KC_FLAG_ZERO) != 0)) { --*start; --p; }
    p = *end - 1;      // (a region that ends at position 0 -- only possible with ext_len_edge = 0 -- makes the reference read data[-1],
                       // contig.c:512-513: no defined result; no read in front of the array here)
    while (*end < bend && p >= 0 && (code[p] == code[p + 1] || (flag[p + 1] & KC_FLAG_ZERO) != 0)) { ++*end; ++p; }
}

// returns the number of int32 values written to out (pairs), or -1 when out_cap is too small
NP1_HD int32_t kc_find_regions(const uint8_t* code, const uint8_t* flag, int32_t L, const uint32_t* flagged, uint32_t m,
                               uint32_t gap, uint32_t con, int32_t ext, bool with_ext, int32_t* out, int32_t out_cap) {
    int32_t n = 0;
    uint32_t k = 0;
    int64_t cursor = 0;
    const int32_t end = L - 1;
    while (k < m) {
        if ((int64_t)flagged[k] < cursor) { ++k; continue; }
        int32_t qstart = (int32_t)flagged[k], qend = qstart;
        uint16_t pcon = 1;
        int64_t last = flagged[k];
        ++k;
        for (;;) {
            const bool has_next = k < m;
            const int64_t next = has_next ? (int64_t)flagged[k] : (int64_t)1 << 40;
            if (!has_next || next - last - 1 > (int64_t)gap) {
                const int64_t close_i = last + (int64_t)gap + 1;   // first position where pgap exceeds gap
                if (close_i > end) {                               // the walk ends with the region still open: kept whatever pcon is
                    kc_brim(code, flag, ext, with_ext, 0, end, &qstart, &qend);
                    if (n + 2 > out_cap) return -1;
                    out[n++] = qstart; out[n++] = qend;
                    return n;
                }
                cursor = close_i + 1;
                if (pcon > con) {
                    kc_brim(code, flag, ext, with_ext, 0, end, &qstart, &qend);
                    if (n + 2 > out_cap) return -1;
                    out[n++] = qstart; out[n++] = qend;
                    if ((int64_t)qend > close_i) cursor = (int64_t)qend + 1;
                }
                break;
            }
            pcon = (next - last - 1 == 0) ? (uint16_t)(pcon + 1) : (uint16_t)1;
            qend = (int32_t)next;
            last = next;
            ++k;
        }
    }
    return n;
}

// ---- run-parallel form of the region walk (used by k_kc_regions; the CPU tests replay it against kc_find_regions) ----
struct KcRun { int32_t first_pos, s, e, close_i; uint32_t emit, first_k, last_k, pad; };   // one run of one pass
// the region of the run F[k0..k1] (inclusive) as the walk would close it when it starts at k0
NP1_HD void kc_run_region(const uint32_t* F, uint32_t k0, uint32_t k1, const uint8_t* code, const uint8_t* flag, int32_t L,
                                              uint32_t gap, uint32_t con, int32_t ext, bool with_ext, KcRun* out) {
    const int32_t end = L - 1;
    int32_t qstart = (int32_t)F[k0], qend = (int32_t)F[k1];
    uint32_t pcon = 1;   // adjacent flagged positions ending at the run's last one; a uint16 counter in the reference (it wraps)
    for (uint32_t k = k1; k > k0 && F[k] - F[k - 1] == 1u; --k) ++pcon;
    const int64_t close_i = (int64_t)F[k1] + (int64_t)gap + 1;
    const bool open_end = close_i > end;
    const bool emit = open_end || (pcon & 0xffffu) > con;
    if (emit) kc_brim(code, flag, ext, with_ext, 0, end, &qstart, &qend);
    out->first_pos = (int32_t)F[k0];
    out->s = qstart;
    out->e = qend;
    out->close_i = close_i > 0x7fffffff ? 0x7fffffff : (int32_t)close_i;
    out->emit = emit ? (open_end ? 2u : 1u) : 0u;
    out->first_k = k0;
    out->last_k = k1;
    out->pad = 0;
}

// the cursor the walk holds after a run it took (closing position + 1, or behind the extended region)
NP1_HD int64_t kc_reach(const KcRun& x) {
    return (x.emit == 1u && (int64_t)x.e > (int64_t)x.close_i) ? (int64_t)x.e + 1 : (int64_t)x.close_i + 1;
}
// Replays the reference's walk from run q, whose first position lies behind the cursor its predecessor leaves, until
// a run is again taken as computed; returns that run's index (n_runs when the runs are used up).  Swallowed runs lose
// their region, cut runs are recomputed from their first surviving position; `writer` stores the updated records (on
// the GPU every lane follows the same chain and one of them writes).  *ended: the walk ended with a region still open.
NP1_HD uint32_t kc_chain(const uint32_t* F, KcRun* runs, uint32_t n_runs, uint32_t q, const uint8_t* code, const uint8_t* flag, int32_t L,
                         uint32_t gap, uint32_t con, int32_t ext, bool with_ext, bool writer, bool* ended) {
    int64_t cursor = kc_reach(runs[q - 1]);
    for (; q < n_runs; ++q) {
        KcRun rr = runs[q];
        if ((int64_t)rr.first_pos >= cursor) break;     // taken as computed: back in step with the parallel result
        uint32_t k0 = rr.first_k;
        while (k0 <= rr.last_k && (int64_t)F[k0] < cursor) ++k0;
        if (k0 > rr.last_k) {                           // the whole run lies behind the cursor: no region, the cursor stays
            rr.emit = 0;
            rr.close_i = (int32_t)(cursor - 1);         // (a chain starting right behind it would see the same cursor)
            rr.e = rr.close_i;
            if (writer) runs[q] = rr;
            continue;
        }
        kc_run_region(F, k0, rr.last_k, code, flag, L, gap, con, ext, with_ext, &rr);
        rr.first_k = k0;
        if (writer) runs[q] = rr;
        cursor = kc_reach(rr);
        if (rr.emit == 2u) { *ended = true; break; }    // the walk ended with the region still open
    }
    return q;
}
// contig_merge_region with the last output region held in registers (kc_merge_regions below is the literal statement).
// Needs v[0] < v[1]: then the output never runs ahead of the input.  The GPU kernel walks the same steps with the inputs
// preloaded 64 at a time (kc_merge_wave).
NP1_HD int32_t kc_merge_fast(int32_t* v, int32_t n) {
    const int32_t nreg = n / 2;
    int32_t qi = 0, qs = v[0], qe = v[1], length = 2;
    for (int32_t i = 0; i < nreg; ++i) {
        const int32_t ps = v[2 * i], pe = v[2 * i + 1];
        if (ps >= qe) {
            ++qi;
            qs = ps;
            qe = pe;
            v[2 * qi] = qs;
            v[2 * qi + 1] = qe;
            length += 2;
        } else {
            while (ps < qs) { --qi; qs = v[2 * qi]; }
            qe = pe;
            v[2 * qi + 1] = qe;
        }
    }
    return length;
}

// contig_merge_region, literal (contig.c:595-620); returns the new number of values
NP1_HD int32_t kc_merge_regions(int32_t* v, int32_t n) {
    if (n == 0) return 0;
    int32_t *pstart = v, *pend = v + 1, *qstart = v, *qend = v + 1, length = 2;
    for (int32_t i = 0; i < n; i += 2) {
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
    return length;
}

// ---- per-slot context lists (base.c:60-71) on a shared pool ---------------------------------------------------
NP1_HD void kc_add_data(const KcCtx& c, uint32_t s, uint32_t kmer) {
    uint32_t idx = c.lhead[s], last = 0;
    while (idx) {
        uint32_t* e = c.lpool + 2ull * (idx - 1);
        if ((e[0] & 0xffffu) == kmer) {
            e[0] = (e[0] & 0xffffu) | ((((e[0] >> 16) + 1) & 0xffffu) << 16);
            c.scount[s] = (uint16_t)(c.scount[s] + 1);
            return;
        }
        last = idx;
        idx = e[1];
    }
    const uint32_t n = kc_bump(c.lcount, 1);
    if (n >= c.lcap) { np1_atomic_or(c.err, ERR_KC_POOL); return; }
    uint32_t* e = c.lpool + 2ull * n;
    e[0] = kmer | 1u << 16;
    e[1] = 0;
    if (last) c.lpool[2ull * (last - 1) + 1] = n + 1; else c.lhead[s] = n + 1;
    c.scount[s] = (uint16_t)(c.scount[s] + 1);
}

struct KcPileupSink {
    const KcCtx* c;
    uint32_t g0;
    uint32_t kmer;
    NP1_HD void vote(int32_t pos, uint32_t col, uint32_t sym, int32_t, bool) {
        kmer = ((kmer & 0xffu) << 4) | sym;
        kc_add_data(*c, c->soff[g0 + (uint32_t)pos] + col, kmer);
    }
};

// contig_as_read on [start,end] (contig.c:373-383): the draft votes once per slot, context restarted at `start`
NP1_HD void kc_as_read(const KcCtx& c, uint32_t g0, int32_t start, int32_t end) {
    uint32_t kmer = 0;
    const uint32_t s0 = c.soff[g0 + (uint32_t)start], s1 = c.soff[g0 + (uint32_t)end];
    for (uint32_t s = s0; s <= s1; ++s) {
        kmer = ((kmer & 0xffu) << 4) | c.sbase[s];
        c.srefk[s] = (uint16_t)kmer;
        kc_add_data(c, s, kmer);
    }
}

// contig_parse_region (contig.c:688-704): records of exactly `level` overlapping [start,end], in file order
NP1_HD void kc_parse_region(const KcCtx& c, uint32_t ctg, int32_t start, int32_t end, uint32_t level) {
    const uint32_t g0 = c.ctg_off[ctg];
    const int64_t rb = (int64_t)c.read_begin[ctg], re = (int64_t)c.read_begin[ctg + 1];
    for (int64_t r = kc_lower_bound_pos(c.R, rb, re, start - c.max_span); r < re; ++r) {
        if (c.R.pos[r] >= end + 1) break;
        if (c.endpos[r] <= start || c.level[r] != level) continue;
        KcPileupSink sink{&c, g0, 0};
        kc_walk(c, r, g0, start, end, sink);
    }
}

// ---- region DP (contig.c:424-496) with exact fixed-point scores ------------------------------------------------
// states of slot k of the region live at st_*[ (sb + k) * 16 + base ]
// Scores are exact integers (value * 2^K, rate = Rfix / 2^K) or, when the rate is no such fraction (K < 0), the reference's own
// doubles kept as bit patterns in the same 64-bit cells and evaluated in its order (score += count - total * rate, contig.c:448).
NP1_HD double kc_as_double(long long c) { double v; memcpy(&v, &c, 8); return v; }
NP1_HD long long kc_as_cell(double v) { long long c; memcpy(&c, &v, 8); return c; }
NP1_HD bool kc_less(long long a, long long b, bool fp) { return fp ? kc_as_double(a) < kc_as_double(b) : a < b; }
NP1_HD bool kc_equal(long long a, long long b, bool fp) { return fp ? kc_as_double(a) == kc_as_double(b) : a == b; }

struct KcStates {
    long long* sc;
    uint16_t* km;
    uint8_t* rk;
    bool fp = false;
    NP1_HD void clear() { for (int b = 0; b < 16; ++b) rk[b] = 0xff; }
    NP1_HD int first_max() const {   // first strict maximum in insertion order (base.c:185-197); -1 when empty
        int best = -1;
        for (int b = 0; b < 16; ++b)
            if (rk[b] != 0xff && (best < 0 || kc_less(sc[best], sc[b], fp) || (kc_equal(sc[b], sc[best], fp) && rk[b] < rk[best]))) best = b;
        return best;
    }
    NP1_HD uint32_t count() const { uint32_t n = 0; for (int b = 0; b < 16; ++b) n += rk[b] != 0xff; return n; }
};

// contig_region_score + contig_region_correct on [start,end]; returns false on an inconsistent pileup or scratch overflow
NP1_HD bool kc_region_solve(const KcCtx& c, uint32_t g0, int32_t start, int32_t end, int K, long long Rfix) {
    const bool fp = K < 0;
    const uint32_t s0 = c.soff[g0 + (uint32_t)start], s1 = c.soff[g0 + (uint32_t)end];
    const uint32_t n = s1 - s0 + 1;
    const uint32_t sb = kc_bump(c.st_count, n + 1);
    if ((uint64_t)sb + n + 1 > c.st_cap) { np1_atomic_or(c.err, ERR_KC_POOL); return false; }
    // seed: one zero-score state per distinct previous byte of the first slot's contexts (contig.c:459-464)
    KcStates seed{c.st_score + 16ull * sb, c.st_kmer + 16ull * sb, c.st_rank + 16ull * sb, fp};
    seed.clear();
    {
        uint32_t rank = 0;
        for (uint32_t idx = c.lhead[s0]; idx; idx = c.lpool[2ull * (idx - 1) + 1]) {
            const uint32_t t = (c.lpool[2ull * (idx - 1)] & 0xffffu) >> 4;
            const uint32_t b = t & 0xf;
            if (seed.rk[b] == 0xff) seed.rk[b] = (uint8_t)rank++;
            seed.sc[b] = 0;
            seed.km[b] = (uint16_t)t;
        }
    }
    bool ok = true;
    for (uint32_t k = 0; k < n; ++k) {   // forward (contig.c:424-454)
        const uint32_t s = s0 + k;
        KcStates prev{c.st_score + 16ull * (sb + k), c.st_kmer + 16ull * (sb + k), c.st_rank + 16ull * (sb + k), fp};
        KcStates cur{c.st_score + 16ull * (sb + k + 1), c.st_kmer + 16ull * (sb + k + 1), c.st_rank + 16ull * (sb + k + 1), fp};
        cur.clear();
        const uint32_t cnt_all = c.scount[s];
        const uint32_t tot = cnt_all > 1 ? cnt_all - 1 : cnt_all;
        const int pfm = prev.first_max();
        uint32_t ncur = 0;
        for (uint32_t idx = c.lhead[s]; idx; idx = c.lpool[2ull * (idx - 1) + 1]) {
            const uint32_t ent = c.lpool[2ull * (idx - 1)];
            const uint32_t kmer = ent & 0xffffu;
            uint32_t cnt = ent >> 16;
            const uint32_t t = kmer >> 4, p = t & 0xf;
            long long S0 = 0;
            if (p == 0) { if (pfm < 0) ok = false; else S0 = prev.sc[pfm]; }
            else if (prev.rk[p] != 0xff) S0 = prev.sc[p];
            else ok = false;
            if (kmer == c.srefk[s] && cnt_all > 1) cnt = (cnt - 1) & 0xffffu;
            long long v;
            if (fp) {   // a zero cell is +0.0: the seeds need no conversion
                const double prod = (double)(int)tot * c.rate;
                const double inc = (double)(int)cnt - prod;
                v = kc_as_cell(kc_as_double(S0) + inc);
            } else {
                v = S0 + ((long long)cnt << K) - (long long)tot * Rfix;
            }
            const uint32_t b = kmer & 0xf;
            if (kmer != 0) {
                if (cur.rk[b] == 0xff) { cur.rk[b] = (uint8_t)ncur++; cur.sc[b] = v; cur.km[b] = (uint16_t)kmer; }
                else if (kc_less(cur.sc[b], v, fp)) { cur.sc[b] = v; cur.km[b] = (uint16_t)kmer; }
            } else {
                const int fm = cur.first_max();
                if (fm < 0 || kc_less(cur.sc[fm], v, fp)) {
                    if (cur.rk[0] == 0xff) cur.rk[0] = (uint8_t)ncur++;
                    cur.sc[0] = v; cur.km[0] = 0;
                }
            }
        }
    }
    if (!ok) { np1_atomic_or(c.err, ERR_KC_INCONSISTENT); return false; }
    // backward (contig.c:473-496): from (end,0) down; base `start` and its insertion columns are not revisited when it
    // owns insertion columns (contig_data_pre, contig.c:402-422), unless the region is a single position
    uint32_t stop = s0;
    if (start != end) {
        const uint32_t nins0 = c.soff[g0 + (uint32_t)start + 1] - s0 - 1;
        if (nins0 > 0) stop = s0 + nins0 + 1;
    }
    KcStates lastst{c.st_score + 16ull * (sb + n), c.st_kmer + 16ull * (sb + n), c.st_rank + 16ull * (sb + n), fp};
    int b = lastst.first_max();
    for (uint32_t k = n; k-- > 0;) {
        const uint32_t s = s0 + k;
        if (s < stop) break;
        if (b < 0) { np1_atomic_or(c.err, ERR_KC_INCONSISTENT); return false; }
        KcStates cur{c.st_score + 16ull * (sb + k + 1), c.st_kmer + 16ull * (sb + k + 1), c.st_rank + 16ull * (sb + k + 1), fp};
        const uint32_t kk = cur.km[b];
        if (c.third_rule) {   // ts_region_correct (snpphase.c:843-871): long-read evidence marks, the base only where the rule lets it
            const bool col0 = s == c.soff[c.sown[s]];
            if ((c.sflag[s] & KC_FLAG_ZERO) || (col0 && b != 3)) c.sbase[s] = (uint8_t)b;
            // base_merge_kmer, IN PLACE like the reference (base.c:123-146): one entry per base symbol in first-seen order, 16-bit sums,
            // contexts dropped.  A slot shared by two touching regions is scored by the second one on this merged list.
            uint32_t mc[16], order[16], first_idx[16], nm = 0;
            for (int t = 0; t < 16; ++t) mc[t] = 0xffffffffu;
            for (uint32_t idx = c.lhead[s], prev_idx = 0; idx;) {
                uint32_t* e = c.lpool + 2ull * (idx - 1);
                const uint32_t sy = e[0] & 0xfu, nxt = e[1];
                if (mc[sy] == 0xffffffffu) {
                    mc[sy] = e[0] >> 16;
                    order[nm++] = sy;
                    first_idx[sy] = idx;
                    e[0] = sy | mc[sy] << 16;
                    prev_idx = idx;
                } else {
                    mc[sy] = (mc[sy] + (e[0] >> 16)) & 0xffffu;
                    c.lpool[2ull * (first_idx[sy] - 1)] = sy | mc[sy] << 16;
                    c.lpool[2ull * (prev_idx - 1) + 1] = nxt;   // unlink
                }
                idx = nxt;
            }
            if (nm >= 2) {   // stable top two (base.c:91-121)
                uint32_t m0 = order[0], m1 = order[1];
                if (mc[m1] > mc[m0]) { const uint32_t t = m0; m0 = m1; m1 = t; }
                for (uint32_t t = 2; t < nm; ++t) {
                    const uint32_t sy = order[t];
                    if (mc[sy] > mc[m1]) { if (mc[sy] > mc[m0]) { m1 = m0; m0 = sy; } else m1 = sy; }
                }
                const double rate = mc[m1] / (double)mc[m0];
                const uint32_t bb = c.sbase[s];
                if (m0 != bb || rate > c.max_indel_factor_lgs) {
                    if (bb == 3 || !col0 || m0 != bb || rate > c.max_snp_factor_lgs) c.sflag[s] = (uint8_t)(c.sflag[s] | 16u);
                    else c.sflag[s] = (uint8_t)(c.sflag[s] & ~16u);
                }
            }
            KcStates prev3{c.st_score + 16ull * (sb + k), c.st_kmer + 16ull * (sb + k), c.st_rank + 16ull * (sb + k), fp};
            const uint32_t arg3 = kk >> 4;
            if (arg3) b = prev3.rk[arg3 & 0xf] != 0xff ? (int)(arg3 & 0xf) : -1;
            else b = prev3.first_max();
            continue;
        }
        c.sbase[s] = (uint8_t)b;
        uint32_t fl = c.sflag[s];
        if (c.scount[s] == 1) fl |= KC_FLAG_ZERO; else fl &= ~KC_FLAG_ZERO;
        uint32_t cntb = 0;
        for (uint32_t idx = c.lhead[s]; idx; idx = c.lpool[2ull * (idx - 1) + 1]) {
            const uint32_t ent = c.lpool[2ull * (idx - 1)];
            if ((ent & 0xfu) == (uint32_t)b) cntb += ent >> 16;
        }
        if ((double)cntb / (double)c.scount[s] < c.min_count_ratio_skip) fl |= KC_FLAG_COVERAGE; else fl &= ~KC_FLAG_COVERAGE;
        c.sflag[s] = (uint8_t)fl;
        KcStates prev{c.st_score + 16ull * (sb + k), c.st_kmer + 16ull * (sb + k), c.st_rank + 16ull * (sb + k), fp};
        const uint32_t arg = kk >> 4;
        if (arg) b = prev.rk[arg & 0xf] != 0xff ? (int)(arg & 0xf) : -1;
        else b = prev.first_max();
    }
    return true;
}

// slot cursor in (base, insertion-column) order with the reference's end-of-contig rule (contig_data_next, contig.c:385-400)
struct KcCursor {
    const KcCtx* c;
    uint32_t g0;
    int32_t Lc, i, j;
    NP1_HD uint32_t slot() const { return c->soff[g0 + (uint32_t)i] + (uint32_t)j; }
    NP1_HD bool in(int32_t end) const { return i < end || (i == end && j == 0); }
    NP1_HD void next() {
        if (i + 1 >= Lc) { i = Lc; return; }
        const uint32_t n = c->soff[g0 + (uint32_t)i + 1] - c->soff[g0 + (uint32_t)i] - 1;
        if ((uint32_t)j == n) { ++i; j = 0; } else ++j;
    }
};

// contig_get_region(start, end, gap 0, con 0, FLAG_ZERO, no extension) over the slots of a region (the level-2
// fallback of contig_score_correct, contig.c:721-733); writes (start,end) pairs of local positions
NP1_HD int32_t kc_zero_subregions(const KcCtx& c, uint32_t ctg, int32_t start, int32_t end, int32_t* out, int32_t out_cap) {
    const uint32_t g0 = c.ctg_off[ctg];
    KcCursor cur{&c, g0, (int32_t)(c.ctg_off[ctg + 1] - g0), start, 0};
    int32_t n = 0, qstart = -1, qend = -1;
    while (cur.in(end)) {
        if (c.sflag[cur.slot()] & KC_FLAG_ZERO) {
            if (qstart == -1) qstart = cur.i;
            qend = cur.i;
        } else if (qstart != -1) {   // gap 0: the first unflagged slot closes the region; pcon >= 1 > con = 0
            int32_t a = qstart, b = qend;
            a = a >= start + c.ext_len_edge ? a - c.ext_len_edge : start;
            b = b <= end - c.ext_len_edge ? b + c.ext_len_edge : end;
            if (n + 2 > out_cap) return -1;
            out[n++] = a; out[n++] = b;
            if (b > cur.i) { cur.i = b; cur.j = 0; }
            qstart = qend = -1;
        }
        cur.next();
    }
    if (qstart != -1) {
        int32_t a = qstart, b = qend;
        a = a >= start + c.ext_len_edge ? a - c.ext_len_edge : start;
        b = b <= end - c.ext_len_edge ? b + c.ext_len_edge : end;
        if (n + 2 > out_cap) return -1;
        out[n++] = a; out[n++] = b;
    }
    return n;
}

// contig_score_correct(start, end, 0x12, rate) (contig.c:706-734): level-2 pileup, then level-1 on what is still uncovered
NP1_HD void kc_score_correct_level2(const KcCtx& c, uint32_t ctg, int32_t start, int32_t end) {
    const uint32_t g0 = c.ctg_off[ctg];
    kc_as_read(c, g0, start, end);
    kc_parse_region(c, ctg, start, end, 2);
    if (!kc_region_solve(c, g0, start, end, c.K, c.Rfix)) return;
    int32_t sub[128];
    int32_t ns = kc_zero_subregions(c, ctg, start, end, sub, 128);
    if (ns < 0) { np1_atomic_or(c.err, ERR_KC_REGIONS); return; }
    ns = kc_merge_regions(sub, ns);
    for (int32_t i = 0; i < ns; i += 2) {
        kc_parse_region(c, ctg, sub[i], sub[i + 1], 1);
        if (!kc_region_solve(c, g0, sub[i], sub[i + 1], c.K, c.Rfix)) return;
    }
}

// ---- ss_spilt_region for one merged k-mer region (kmercount.c:128-173): cut points at the midpoints of the
// unflagged runs that follow the first flagged slot; returns the number of values (pairs) written
NP1_HD int32_t kc_split_region(const KcCtx& c, uint32_t ctg, int32_t rs, int32_t re, int32_t* out, int32_t out_cap) {
    const uint32_t g0 = c.ctg_off[ctg];
    int32_t n = 0;
    if (n + 1 > out_cap) return -1;
    out[n++] = rs;
    if (re - rs > c.max_len_kmer) {
        KcCursor cur{&c, g0, (int32_t)(c.ctg_off[ctg + 1] - g0), rs, 0};
        while (cur.in(re)) {   // skip to the first flagged slot
            if (c.sflag[cur.slot()] & KC_FLAG_ZERO) break;
            cur.next();
        }
        int32_t qstart = -1, qend = -1;
        while (cur.in(re)) {
            if (!(c.sflag[cur.slot()] & KC_FLAG_ZERO)) {
                if (qstart == -1) qstart = cur.i;
                qend = cur.i;
            } else if (qstart != -1) {
                const int32_t k = (qstart + qend) >> 1;
                if (n + 2 > out_cap) return -1;
                out[n++] = k; out[n++] = k;
                qstart = qend = -1;
            }
            cur.next();
        }
    }
    if (n + 1 > out_cap) return -1;
    out[n++] = re;
    return n;
}

// ---- fts_spilt_region (snpvalid.c:38-66) for one region nothing spanned in snp_valid's first round: appends to out[] (which is
// NOT started with the region's start); returns the new count or -1 when out_cap is too small
NP1_HD int32_t kc_fts_split(const KcCtx& c, uint32_t ctg, int32_t start, int32_t end, int32_t* out, int32_t n, int32_t out_cap) {
    const uint32_t g0 = c.ctg_off[ctg];
    KcCursor cur{&c, g0, (int32_t)(c.ctg_off[ctg + 1] - g0), start, 0};
    int32_t qstart = -1, qend = -1;
    while (cur.in(end)) {
        if (!(c.sflag[cur.slot()] & KC_FLAG_ZERO)) {
            if (qstart == -1) qstart = cur.i;
            qend = cur.i;
        } else if (qstart != -1) {
            int32_t count = 2;
            if (qstart == start) { qend = start; --count; }
            int32_t mid = (qstart + qend) / 2;
            for (int32_t k = 0; k < count; ++k) {
                if (n >= out_cap) return -1;
                out[n++] = mid;
                if (qstart != qend) ++mid;
            }
            qstart = qend = -1;
        }
        cur.next();
    }
    if (n >= out_cap) return -1;
    out[n++] = end;
    return n;
}

// ---- ss_kmer_correct for one part [start,end] (kmercount.c:175-261, 332-465) -----------------------------------
struct KcHapSink {
    const KcCtx* c;
    uint32_t g0;
    uint8_t* buf;
    int32_t length, cap, qual, del;
    const uint8_t* q;
    NP1_HD void vote(int32_t pos, uint32_t col, uint32_t sym, int32_t qpos, bool pad) {
        if (length < cap) buf[length] = (uint8_t)sym;
        ++length;
        if (qpos >= 0) qual += q[qpos];
        if (pad) ++del;
        // snp_valid's second round can pair a region start with a leftover end far away: an insertion column that was never created
        // there is a null list in the reference (undefined upstream), reported instead of written through
        if (col >= c->soff[g0 + (uint32_t)pos + 1] - c->soff[g0 + (uint32_t)pos]) { np1_atomic_or(c->err, ERR_KC_UNDEFINED); return; }
        const uint32_t s = c->soff[g0 + (uint32_t)pos] + col;
        if (!c->keep_zero_marks) c->sflag[s] = (uint8_t)(c->sflag[s] & ~KC_FLAG_ZERO);   // flagzero == 0: a covering read clears the mark (accepted or not)
    }
};

// candidate table in the haplotype pool: [n_cand] entries of {num, mapqual, qual} + length bytes each
struct KcCand { int32_t num, mapqual, qual; };

// Computes the winner haplotype of one part into winner[0..length) and returns 1, or returns 0 when no spanning
// record yields a full-length haplotype.  `scratch` holds up to max_cand candidates of `length` bytes + one work row.
// rp (optional): what the replay of the reference's region iterator says this part gets (np1_replay.h): the records of the first loop
// in order, the record left in the buffer, the passes of the second loop (n2 < 0: not known yet -- the call then returns 2 when
// the first loop leaves no candidate, and is repeated with n2 set).  Without it: records in file order.
// brk (optional): receives how many records of the list the first loop consumed when it left through the max_count_kmer break, 0 = it did not.
struct KcReplay { const uint32_t* list; uint32_t n; int64_t stale; int32_t n2; uint32_t* brk; };
NP1_HD int32_t kc_part_winner(const KcCtx& c, uint32_t ctg, int32_t start, int32_t end, bool has_next_record,
                              uint8_t* winner, int32_t length, const KcReplay* rp = nullptr) {
    const uint32_t g0 = c.ctg_off[ctg];
    const int64_t rb = (int64_t)c.read_begin[ctg], re = (int64_t)c.read_begin[ctg + 1];
    const int64_t r0 = kc_lower_bound_pos(c.R, rb, re, start - c.max_span);
    const int64_t rstop = kc_lower_bound_pos(c.R, rb, re, start);   // first record with pos >= start ends the swapped-interval scan
    // spanning records: pos < start and endpos > end + 1 (contig.c:1130-1135)
    int64_t n_span = 0;
    if (rp) n_span = rp->n;
    else for (int64_t r = r0; r < rstop; ++r) n_span += c.endpos[r] > end + 1;
    // scratch: candidates (distinct haplotypes, first-seen order) + one work row
    const uint32_t max_cand = (uint32_t)(n_span > 0 ? n_span : 1) + 1;
    const uint32_t stride = ((uint32_t)length + 12u + 3u) & ~3u;   // 12 bytes KcCand header + haplotype, 4-byte aligned
    const uint32_t bytes = (max_cand + 1) * stride;
    const uint32_t off = kc_bump(c.hcount, (bytes + 15u) & ~15u);
    if ((uint64_t)off + bytes > c.hcap) { np1_atomic_or(c.err, ERR_KC_POOL); return 0; }
    uint8_t* base = c.hpool + off;
    uint8_t* work = base + (uint64_t)max_cand * stride;
    uint32_t ncand = 0;
    int32_t count = 0, last_mapqual = 0;
    auto parse = [&](int64_t r, int32_t* out_mapqual) -> void {   // ss_kmer_get_region + ss_parse_read_kmer
        KcHapSink sink{&c, g0, work + 12, 0, length, 0, 0, c.qual + c.qual_off[r]};
        int32_t mq = 0;
        if (c.R.n_cigar[r]) {
            mq = c.mapq[r];
            kc_walk(c, r, g0, start, end, sink);
            if (sink.length > 0 && sink.length != sink.del) sink.qual /= sink.length - sink.del; else sink.qual = 0;
        }
        if (sink.length == length) {
            uint32_t hit = ncand;
            for (uint32_t k = 0; k < ncand; ++k) {
                const uint8_t* h = base + (uint64_t)k * stride + 12;
                bool same = true;
                for (int32_t t = 0; t < length; ++t) if (h[t] != work[12 + t]) { same = false; break; }
                if (same) { hit = k; break; }
            }
            if (hit == ncand) {
                if (ncand < max_cand) {
                    uint8_t* dst = base + (uint64_t)ncand * stride;
                    KcCand* cd = reinterpret_cast<KcCand*>(dst);
                    cd->num = 1; cd->mapqual = mq; cd->qual = sink.qual;
                    for (int32_t t = 0; t < length; ++t) dst[12 + t] = work[12 + t];
                    ++ncand;
                }
            } else {
                KcCand* cd = reinterpret_cast<KcCand*>(base + (uint64_t)hit * stride);
                cd->num++; cd->mapqual += mq; cd->qual += sink.qual;
            }
            *out_mapqual = mq;
        } else {
            *out_mapqual = 0;
        }
    };
    if (rp) {
        uint32_t consumed = 0;
        for (uint32_t t = 0; t < rp->n; ++t) {
            const int64_t r = (int64_t)rp->list[t];
            if (c.level[r] == 2) {
                parse(r, &last_mapqual);
                if (last_mapqual == 60) {
                    ++count;
                    if (count >= c.max_count_kmer) { consumed = t + 1; break; }
                }
            }
        }
        if (rp->brk) *rp->brk = consumed;
        if (ncand == 0) {
            if (rp->n2 < 0) return 2;
            if (rp->stale >= 0 && c.level[rp->stale] == 1)
                for (int32_t t = 0; t < rp->n2; ++t) { int32_t mq; parse(rp->stale, &mq); }
        }
    } else {
    for (int64_t r = r0; r < rstop; ++r) {
        if (!(c.endpos[r] > end + 1)) continue;
        if (c.level[r] == 2) {
            parse(r, &last_mapqual);
            if (last_mapqual == 60) {
                ++count;
                if (count >= c.max_count_kmer) break;
            }
        }
    }
    }
    if (!rp && ncand == 0) {
        // bug-compatible fallback (kmercount.c:212-217): one pass per spanning record, always on the record the first
        // loop stopped on: first record with pos >= start, else the last one read = the contig's last record (the chunk list of
        // the query ends with the contig's records, the reader never reaches another contig's)
        int64_t stale = -1;
        (void)has_next_record;
        if (rstop < re) stale = rstop;
        else if (re > rb) stale = re - 1;
        if (stale >= 0 && c.level[stale] == 1)
            for (int64_t t = 0; t < n_span; ++t) { int32_t mq; parse(stale, &mq); }
    }
    if (ncand == 0) return 0;
    uint32_t best = ncand;
    if (count == c.max_count_kmer) {
        const int32_t want = 60 * count;
        for (uint32_t k = 0; k < ncand; ++k)
            if (reinterpret_cast<KcCand*>(base + (uint64_t)k * stride)->mapqual == want) { best = k; break; }
    }
    if (best == ncand) {
        best = 0;
        for (uint32_t k = 0; k < ncand; ++k) {   // ks_compare (kmercount.c:63-88): strict improvement only
            const KcCand* a = reinterpret_cast<KcCand*>(base + (uint64_t)best * stride);
            const KcCand* b = reinterpret_cast<KcCand*>(base + (uint64_t)k * stride);
            int cmp = 0;
            if (a != b) {
                if (a->num != b->num) cmp = a->num > b->num ? 1 : -1;
                else if (a->mapqual != b->mapqual) cmp = a->mapqual > b->mapqual ? 1 : -1;
                else if (a->qual != b->qual) cmp = a->qual > b->qual ? 1 : -1;
            }
            if (cmp < 0) best = k;
        }
    }
    const uint8_t* h = base + (uint64_t)best * stride + 12;
    for (int32_t t = 0; t < length; ++t) winner[t] = h[t];
    return 1;
}

}  // namespace np1k
