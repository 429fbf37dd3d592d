// Split-read structural layer (host side).  See np2_sv.h.  Every routine restates the behaviour of the reference
// function named next to it, including its integer widths and its in-place reorderings.
#include "np2_sv.h"

#include <algorithm>
#include <cstdlib>
#include <cstddef>
#include <cstring>

namespace np2 {
namespace {

constexpr int INS_WIN_STEP = 10, INS_WIN_DIV = 20, INS_WIN_MIN_SIZE = 500;
constexpr double INS_MIN_DEPTH_RATIO = 0.1, CLUSTER_MIN_DEPTH_RATIO = 0.2;
constexpr uint32_t LQSEQ_MAX_CAN_COUNT = 60;

inline uint32_t mabs(uint32_t x, uint32_t y) { return x > y ? x - y : y - x; }

// k-th smallest value (0-based).  The reference selects it with an in-place partition loop (ctg_cns.c:3249-3268); only the value
// is used, so any selection gives the same answer.
template <class T> T kth_smallest(std::vector<T>& v, size_t k) {
    std::nth_element(v.begin(), v.begin() + (std::ptrdiff_t)k, v.end());
    return v[k];
}

// Mean depth of the bins that carry reads, sampled every tenth bin, with the high outliers cut off iteratively: the ceiling is
// three times the previous round's mean and the rounds stop when the mean no longer falls below a third of it (ctg_cns.c:3279-3296).
int trimmed_mean_depth(const std::vector<uint16_t>& depth, int32_t n_bins, int skip) {
    uint64_t sum = 150, cnt = 1, ceiling = 0;
    while (cnt && sum / cnt > ceiling / 3) {
        ceiling = sum / cnt * 3;
        sum = cnt = 0;
        for (int32_t i = skip; i < n_bins - skip; i += 10) {
            const uint16_t d = depth[(size_t)i];
            if (d && d < ceiling) { sum += d; ++cnt; }
        }
    }
    return cnt ? (int)(sum / cnt) : 0;
}

void cluster_median(SvWindow* w, SvCluster* clu) {   // cal_gap_cluster_median, ctg_cns.c:2509-2549
    auto G = [&](int32_t i) -> const SvGapRead& { return w->gaps[clu->gap[(size_t)i]]; };
    uint32_t offset = 10;
    while (offset <= 100) {
        clu->median = 0;
        int32_t count_m = 0;
        uint32_t count_mc = 0;
        uint64_t count_m_diff = 0;
        for (int32_t i = 0; i < (int32_t)clu->i_m; ++i) {
            const uint32_t median = (G(i).gap.s + G(i).gap.e) / 2;
            if (median == clu->median) continue;
            const uint32_t s = median > offset ? median - offset : 0, e = median + offset;
            int32_t count_t = 0, count_t_diff = 0;
            for (int32_t j = i - 1; j >= 0; --j) {
                const uint32_t mt = (G(j).gap.s + G(j).gap.e) / 2;
                if (mt >= s) { ++count_t; count_t_diff += (int32_t)mabs(mt, median); }
                else break;
            }
            for (int32_t j = i + 1; j < (int32_t)clu->i_m; ++j) {
                const uint32_t mt = (G(j).gap.s + G(j).gap.e) / 2;
                if (mt <= e) { ++count_t; count_t_diff += (int32_t)mabs(mt, median); }
                else break;
            }
            if (count_t > count_m || (count_t == count_m && count_m_diff > (uint64_t)(int64_t)count_t_diff)) {
                count_m = count_t;
                count_mc = median;
                count_m_diff = (uint64_t)(int64_t)count_t_diff;
            }
        }
        if (count_m >= std::max<int32_t>(3, (int32_t)(clu->i_m / 6))) {
            clu->median = count_mc;
            break;
        }
        offset += 10;
    }
    if (offset > 100) clu->median = (G((int32_t)(clu->i_m / 2)).gap.s + G((int32_t)(clu->i_m / 2)).gap.e) / 2;
}

}  // namespace

// Bin width of the depth statistics: a twentieth of the median read span (rounded like the reference), at least 500 (ctg_cns.c:3224-3246).
int sv_cal_rreads_w(std::vector<SvPos>& rs) {
    std::vector<uint32_t> span(rs.size());
    for (size_t i = 0; i < rs.size(); ++i) span[i] = rs[i].e - rs[i].s;
    const uint32_t w = (kth_smallest(span, span.size() / 2) + 1) / INS_WIN_DIV;
    return w > (uint32_t)INS_WIN_MIN_SIZE ? (int)w : INS_WIN_MIN_SIZE;
}

// Depth track: a read counts in the 10-bp bins of its span minus a margin of w at the start and 2 w at the end, and only if it is at
// least 3 w long (ctg_cns.c:3311-3319).  The reference increments the bins one by one; here the two ends go into a difference array
// and the bins are summed once when the window's records are all in (SvWindow::finish_depth).  Bins are 16-bit counters that wrap.
void sv_update_ref_d(SvWindow& win, int w, const SvPos& p, int32_t s) {
    uint32_t lo = p.s > (uint32_t)s ? p.s - (uint32_t)s : 0;
    uint32_t hi = p.e - (uint32_t)s;
    if (hi - lo + 1 < (uint32_t)(w * 3)) return;
    lo = (lo + (uint32_t)w) / INS_WIN_STEP;
    hi = (hi - 2 * (uint32_t)w) / INS_WIN_STEP;
    if (lo > hi) return;
    if (hi + 2 > win.depth_diff.size()) win.depth_diff.resize((size_t)hi + 1026, 0);
    ++win.depth_diff[lo];
    --win.depth_diff[(size_t)hi + 1];
}

void SvWindow::finish_depth() {
    if (ref_ds.size() < depth_diff.size()) ref_ds.resize(depth_diff.size(), 0);
    int32_t run = 0;
    for (size_t i = 0; i < depth_diff.size(); ++i) {
        run += depth_diff[i];
        ref_ds[i] = (uint16_t)(ref_ds[i] + (uint16_t)run);
    }
    std::fill(depth_diff.begin(), depth_diff.end(), 0);
}

// Typical depth of the window: median over the bins between the first and the last bin that carry reads (a margin at the start is
// never looked at); windows of more than 50 000 bins where a fifth of those bins is nearly empty take the trimmed mean instead
// (ctg_cns.c:3298-3313).
int sv_cal_ref_d(const std::vector<uint16_t>& depth, int32_t n_bins) {
    int head = n_bins > 20000 ? 10000 : n_bins > 200 ? 100 : 20, tail = 0;
    while (head < n_bins && !depth[(size_t)head++]) {}
    while (n_bins - 1 - tail >= 0 && !depth[(size_t)(n_bins - 1 - tail++)]) {}
    std::vector<int> body;
    uint32_t nearly_empty = 0;
    for (int32_t i = head; i < n_bins - tail; ++i) {
        body.push_back(depth[(size_t)i]);
        if (body.back() < 4) ++nearly_empty;
    }
    if (body.empty()) return 0;
    if (n_bins > 50000 && (double)nearly_empty / (double)body.size() > 0.2) return trimmed_mean_depth(depth, n_bins, head);
    return kth_smallest(body, body.size() / 2);
}

// Median identity of the assembler's QV track (ctg_cns.c:3270-3278).
int sv_cal_ref_ide(const ref_qv* qv, uint32_t n) {
    if (n == 0 || qv == nullptr) return 0;
    std::vector<int> ide(n);
    for (uint32_t i = 0; i < n; ++i) ide[i] = (int)qv[i].ide;
    return kth_smallest(ide, n / 2);
}

// Low-depth regions of the window (ctg_cns.c:2688-2742).  A bin whose depth is at most a tenth of the typical depth opens a
// region; the region reaches as far to both sides as the depth stays at most a fifth of it (the left walk stops at bin 1 and
// names the first bin above the shoulder); the right end gets one read window on top.  A region that starts within ten read
// windows of the previous one's end extends it.  Coordinates are window bases (bin * 10).
void sv_update_ld_regs(std::vector<SvPos>* out, const std::vector<uint16_t>& depth, int32_t n_bins, int w, int d) {
    const int32_t low = (int32_t)(d * INS_MIN_DEPTH_RATIO);
    const int32_t shoulder = (int32_t)(d * INS_MIN_DEPTH_RATIO * 2);
    auto walk_left = [&](int32_t i) { while (i > 1 && depth[(size_t)i] <= shoulder) --i; return i; };
    auto walk_right = [&](int32_t i) { while (i < n_bins && depth[(size_t)i] <= shoulder) ++i; return i; };
    std::vector<SvPos> regs;
    for (int32_t i = 0; i < n_bins; ++i) {
        if (depth[(size_t)i] > low) continue;
        const int32_t lo = walk_left(i), hi = walk_right(i);
        const uint32_t end = (uint32_t)((hi - 1) * INS_WIN_STEP + w);
        const uint32_t begin = (uint32_t)(lo * INS_WIN_STEP);
        if (regs.empty()) regs.push_back(SvPos{lo > 1 ? begin : 0u, end});                  // only the very first region snaps to 0
        else if (begin > regs.back().e + (uint32_t)(INS_WIN_DIV / 2 * w)) regs.push_back(SvPos{begin, end});
        else regs.back().e = end;
        if (regs.back().s > regs.back().e) std::swap(regs.back().s, regs.back().e);
        i = hi;                                                                              // the loop's own step skips the bin above the shoulder
    }
    out->swap(regs);
}

void sv_update_ld_regs_with_refqv(std::vector<SvPos>* regs, const std::vector<uint16_t>& r, const ref_* ref, int32_t w, int32_t s_t, int32_t e_t,
                                  int32_t d_t, uint32_t ide_t, uint32_t ort_t, uint32_t irt_t) {   // ctg_cns.c:2750-2794
    int32_t t = 0;
    for (uint32_t i = 0; i < ref->qv_l && ref->qv[i].p < (uint32_t)e_t; ++i) {
        const ref_qv& q = ref->qv[i];
        if (q.p < (uint32_t)s_t) continue;
        if (q.ide < ide_t && q.ort < ort_t && q.irt < irt_t) {
            const int32_t s = q.p > (uint32_t)(w * 2 + s_t) ? (int32_t)((q.p - (uint32_t)(w * 2) - (uint32_t)s_t) / INS_WIN_STEP) : 0;
            const int32_t e = q.p + (uint32_t)(w * 2) < (uint32_t)e_t ? (int32_t)((q.p + (uint32_t)(w * 2) - (uint32_t)s_t) / INS_WIN_STEP) : (e_t - s_t) / INS_WIN_STEP;
            int l = 0;
            for (int32_t p = s; p <= e && !l; ++p)
                if ((size_t)p < r.size() && r[(size_t)p] <= d_t) l = 1;
            if (l) {
                ++t;
                regs->push_back(SvPos{q.p - (uint32_t)s_t, q.p + 1 - (uint32_t)s_t});
            }
        }
    }
    if (t) {
        std::stable_sort(regs->begin(), regs->end(), [](const SvPos& a, const SvPos& b) {
            return (a.s == b.s ? (int)(a.e - b.e) : (int)(a.s - b.s)) < 0;
        });
        for (size_t i = 1; i < regs->size(); ++i) {
            if ((*regs)[i].s < (*regs)[i - 1].e + (uint32_t)(INS_WIN_DIV / 2 * w)) {
                (*regs)[i].s = (*regs)[i - 1].s;
                if ((*regs)[i].e < (*regs)[i - 1].e) (*regs)[i].e = (*regs)[i - 1].e;
                (*regs)[i - 1].s = (*regs)[i - 1].e = 0;
            }
        }
    }
}

int sv_update_gap_cluster(SvWindow* w, int rw, int d, int32_t ref_s) {   // ctg_cns.c:2551-2612
    w->clusters.clear();
    if (d < 10) return 0;
    const int md = (int)(d * CLUSTER_MIN_DEPTH_RATIO);
    std::stable_sort(w->gaps.begin(), w->gaps.end(), [](const SvGapRead& a, const SvGapRead& b) {
        if (a.gap.s != b.gap.s) return a.gap.s < b.gap.s;
        return a.gap.e < b.gap.e;
    });
    const int32_t n = (int32_t)w->gaps.size();
    auto ds = [&](int64_t idx) -> uint32_t { return idx >= 0 && (size_t)idx < w->ref_ds.size() ? w->ref_ds[(size_t)idx] : 0u; };
    for (int32_t i = 0; i < n - md; ++i) {
        const int32_t p = (int32_t)((w->gaps[(size_t)i].gap.s + w->gaps[(size_t)i].gap.e) / 2) - ref_s;
        if (p < rw || (int)ds(p / INS_WIN_STEP) >= d / 2) continue;
        int32_t e = (int32_t)w->gaps[(size_t)i].gap.e;
        SvCluster clu;
        int32_t t = 1, j;
        for (j = i + 1; j < n && (int32_t)w->gaps[(size_t)j].gap.s <= e; ++j) {
            if ((int)ds(((int32_t)((w->gaps[(size_t)j].gap.s + w->gaps[(size_t)j].gap.e) / 2) - ref_s) / INS_WIN_STEP) >= d / 2) continue;
            ++t;
            if ((int32_t)w->gaps[(size_t)j].gap.e > e) e = (int32_t)w->gaps[(size_t)j].gap.e;
            if (clu.i_m < (LQSEQ_MAX_CAN_COUNT << 1)) { clu.gap.push_back((uint32_t)j); ++clu.i_m; }
        }
        i = j - 1;
        if ((int)clu.i_m > md && (int)ds(p / INS_WIN_STEP) < t) w->clusters.push_back(std::move(clu));
    }
    int total = 0;
    for (SvCluster& clu : w->clusters) {
        total += (int)clu.i_m;
        std::stable_sort(clu.gap.begin(), clu.gap.end(), [&](uint32_t a, uint32_t b) {
            return w->gaps[a].gap.s + w->gaps[a].gap.e < w->gaps[b].gap.s + w->gaps[b].gap.e;
        });
        cluster_median(w, &clu);
    }
    return total;
}

uint32_t sv_update_align_tags(SvWindow* w, const std::vector<SpanOut>& sup_span, uint32_t seq_count, int32_t ref_s, std::vector<StreamRef>* streams) {
    // ctg_cns.c:2836-2888
    for (SvCluster& clu : w->clusters) {
        uint32_t lqseq_count = 0;
        for (uint32_t offset = 20; lqseq_count < LQSEQ_MAX_CAN_COUNT && lqseq_count < clu.i_m * 0.8 && offset < 300; offset += 20) {
            const uint32_t s = clu.median > offset ? clu.median - offset : 0, e = clu.median + offset;
            for (uint32_t j = 0; j < clu.i_m; ++j) {
                SvGapRead& gap = w->gaps[clu.gap[j]];
                if (gap.l) continue;
                const uint32_t median = (gap.gap.s + gap.gap.e) / 2;
                if (median < s || median > e) continue;
                const SpanOut& a = sup_span[clu.gap[j]];
                if (a.aln_t_s > a.aln_t_e - 500u) continue;
                StreamRef sr;
                sr.set = 1;
                sr.rec = clu.gap[j];
                sr.span = a;
                streams->push_back(sr);
                ++seq_count;
                gap.l = offset / 20;
                gap.s_id = seq_count - 1;
                gap.s_s = a.aln_q_s;
                ++lqseq_count;
            }
        }
    }
    (void)ref_s;
    return seq_count;
}

void sv_generate_gapseqs(SvWindow* w, const WindowOutput& wo, int32_t s_) {   // ctg_cns.c:2898-2996
    np2k::Tag tag{0, 0, 0};
    auto TS = [&](uint32_t id) -> uint32_t { return wo.aln_t_s[id]; };
    auto TE = [&](uint32_t id) -> uint32_t { return wo.aln_t_e[id] - 1; };   // inclusive, like align_tags_t.aln_t_e
    for (SvCluster& clu : w->clusters) {
        uint32_t offset = 10, lqseq_rmcount = 0, lqseq_count = 0, lqseq_pcount = 0;
        clu.r.s = clu.r.e = 0;
        for (;;) {
            for (lqseq_pcount = lqseq_count = 0; offset < 30000 && lqseq_pcount < clu.i_m - lqseq_rmcount &&
                                                 (lqseq_count >= lqseq_pcount || lqseq_pcount < clu.i_m / 2); offset += 10) {
                const uint32_t s = clu.median > offset ? clu.median - offset - (uint32_t)s_ : 0;
                const uint32_t e = clu.median + offset - (uint32_t)s_;
                lqseq_pcount = lqseq_count;
                uint32_t j;
                for (lqseq_rmcount = lqseq_count = j = 0; j < clu.i_m; ++j) {
                    SvGapRead& g = w->gaps[clu.gap[j]];
                    if (!g.l) { ++lqseq_rmcount; continue; }
                    uint32_t f = g.p_id, t = g.s_id;
                    if (TS(f) > TS(t)) {
                        std::swap(g.p_id, g.s_id);
                        std::swap(g.p_s, g.s_s);
                        std::swap(f, t);
                    }
                    if (TS(f) < s && TE(f) > s && TS(t) < e && TE(t) > e && s < TS(t) && e > TE(f)) ++lqseq_count;
                }
                if (lqseq_count > lqseq_pcount) { clu.r.s = s; clu.r.e = e; }
            }
            uint32_t offset_step = UINT32_MAX;
            lqseq_count = 0;
            for (uint32_t j = 0; j < clu.i_m; ++j) {
                SvGapRead& g = w->gaps[clu.gap[j]];
                if (!g.l) continue;
                const uint32_t f = g.p_id, t = g.s_id;
                if (TS(f) > clu.r.s || TE(f) < clu.r.s || TS(t) > clu.r.e || TE(t) < clu.r.e) { g.l = 1; continue; }
                uint32_t s = 0, e = g.p_s - 1;
                const uint8_t* tg = wo.tags.data() + wo.tag_off[f];
                while (np2k::next_tag(tg, TS(f), &s, &tag)) {
                    if (tag.q_base != 4) ++e;
                    if ((uint32_t)tag.t_pos == clu.r.s) break;
                }
                g.gap.s = e;
                s = 0;
                e = g.s_s;
                tg = wo.tags.data() + wo.tag_off[t];
                while (np2k::next_tag(tg, TS(t), &s, &tag)) {
                    if ((uint32_t)tag.t_pos == clu.r.e + 1) break;
                    if (tag.q_base != 4) ++e;
                }
                g.gap.e = e;
                if (g.gap.e > g.gap.s + 10) { ++lqseq_count; g.l = 2; }
                else g.l = 1;
                if (mabs(g.gap.s, g.gap.e) < offset_step) offset_step = mabs(g.gap.s, g.gap.e);
            }
            if (lqseq_count >= lqseq_pcount / 2 || lqseq_count >= 10) break;
            offset += offset_step / 2 + 20;
        }
    }
    auto valid = [&](const SvCluster& c) { int n = 0; for (uint32_t j = 0; j < c.i_m; ++j) n += w->gaps[c.gap[j]].l ? 1 : 0; return n; };
    for (size_t i = 0; i < w->clusters.size(); ++i) {
        SvCluster& clu = w->clusters[i];
        if (!clu.i_m) continue;
        if (i + 1 < w->clusters.size() && clu.r.e + 500 >= w->clusters[i + 1].r.s) {
            if (valid(w->clusters[i + 1]) > valid(clu)) { clu.i_m = 0; continue; }
            w->clusters[i + 1].i_m = 0;
        }
    }
}

void sv_update_split_p(std::vector<SvPos>* split_ps, const SvWindow& w, int32_t s, int32_t l, const ref_* ref) {   // ctg_cns.c:2999-3051
    const uint32_t ENDING_FLANK = 1000;
    int j = 0;
    for (size_t i = 0; i < w.ld_regs.size(); ++i) {
        const SvPos& reg = w.ld_regs[i];
        if (reg.s < ENDING_FLANK || reg.e + ENDING_FLANK > (uint32_t)l) continue;
        j = j > 1 ? j - 1 : 0;
        int split = 1;
        for (; j < (int)w.clusters.size() && split; ++j) {
            const SvCluster& clu = w.clusters[(size_t)j];
            if (clu.r.s > reg.e) break;
            if ((reg.s <= clu.r.s && clu.r.s <= reg.e) || (reg.s <= clu.r.e && clu.r.e <= reg.e) || (clu.r.s <= reg.s && reg.s <= clu.r.e) ||
                (clu.r.s <= reg.e && reg.e <= clu.r.e)) split = 0;
        }
        if (split) {
            if (split_ps->empty() || reg.s + (uint32_t)s > split_ps->back().e + 10000) split_ps->push_back(SvPos{reg.s + (uint32_t)s, reg.e + (uint32_t)s});
            else split_ps->back().e = reg.e + (uint32_t)s;
        }
    }
    for (SvPos& reg : *split_ps) {
        uint32_t sco = 0;
        int p = 0;
        for (uint32_t q = 0; q < ref->qv_l && ref->qv[q].p <= reg.e; ++q) {
            if (ref->qv[q].p >= reg.s) {
                const uint32_t v = ref->qv[q].ide + ref->qv[q].ort + ref->qv[q].irt;
                if (sco == 0 || v < sco) { sco = v; p = (int)q; }
            }
        }
        if (sco && sco < 2900) reg.s = reg.e = ref->qv[p].p;
    }
}

std::vector<LqCluster> sv_lq_clusters(const SvWindow& w) {
    std::vector<LqCluster> out;
    for (const SvCluster& clu : w.clusters) {
        LqCluster c;
        c.rs = clu.r.s;
        c.re = clu.r.e;
        c.i_m = clu.i_m;
        for (uint32_t i = 0; i < clu.i_m; ++i) {   // generate_lqseqs_from_cluster, ctg_cns.c:585-600
            const SvGapRead& g = w.gaps[clu.gap[i]];
            if (g.l != 2) continue;
            std::string s;
            for (uint32_t q = g.gap.s; q < g.gap.e; ++q) s.push_back(np2k::nt16_char((uint32_t)(g.dseq[q >> 1] >> ((~q & 1) << 2))));
            c.cands.push_back(std::move(s));
        }
        out.push_back(std::move(c));
    }
    return out;
}

}  // namespace np2
