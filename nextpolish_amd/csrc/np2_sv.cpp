// Split-read structural layer (host side).  See np2_sv.h.  Every routine produces what the reference function named next to it
// produces (integer widths and tie rules included); how it gets there is this file's own: difference arrays and selection for the
// depth statistics, a sorted-midpoint window with prefix sums for the breakpoint estimate, a sweep over the gaps for the clusters,
// rings around the estimate for the choice of supplementary alignments, an interval search object per cluster for the bridged region.
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

// Where a cluster's gaps agree on the breakpoint: the member midpoint with the most other midpoints within `radius` of it (ties: the
// tighter crowd, then the earlier member), tried with radius 10, 20 ... 100 until such a crowd holds at least max(3, members / 6);
// failing that, the middle member's midpoint.  The members are ordered by midpoint, so the crowd of member i is a window
// [lo, hi) that only moves right as i grows, and the spread (sum of distances to mid[i]) comes from prefix sums -- the reference walks
// left and right from every member (cal_gap_cluster_median, ctg_cns.c:2509-2549: same counts, same sums, same winner).
void cluster_median(SvWindow* w, SvCluster* clu) {
    const size_t n = clu->i_m;
    std::vector<uint32_t> mid(n);
    std::vector<int64_t> below(n + 1, 0);          // below[k] = mid[0] + ... + mid[k - 1]
    for (size_t k = 0; k < n; ++k) {
        const SvGapRead& g = w->gaps[clu->gap[k]];
        mid[k] = (g.gap.s + g.gap.e) / 2;
        below[k + 1] = below[k] + (int64_t)mid[k];
    }
    const int64_t quorum = std::max<int64_t>(3, (int64_t)(n / 6));
    for (uint32_t radius = 10; radius <= 100; radius += 10) {
        int64_t best_crowd = 0, best_spread = 0;
        uint32_t best_mid = 0;
        size_t lo = 0, hi = 0;
        for (size_t i = 0; i < n; ++i) {
            if (mid[i] == 0) continue;             // (a midpoint of zero never stands for the cluster)
            const uint32_t from = mid[i] > radius ? mid[i] - radius : 0, to = mid[i] + radius;
            while (mid[lo] < from) ++lo;
            if (hi <= i) hi = i + 1;
            while (hi < n && mid[hi] <= to) ++hi;
            const int64_t crowd = (int64_t)(hi - lo) - 1;
            const int64_t spread = (int64_t)(i - lo) * (int64_t)mid[i] - (below[i] - below[lo]) + (below[hi] - below[i + 1]) - (int64_t)(hi - i - 1) * (int64_t)mid[i];
            if (crowd > best_crowd || (crowd == best_crowd && best_spread > spread)) {
                best_crowd = crowd;
                best_spread = spread;
                best_mid = mid[i];
            }
        }
        if (best_crowd >= quorum) { clu->median = best_mid; return; }
    }
    clu->median = mid[n / 2];
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

// Suspicious spots the assembler's own quality track reports (identity and both read-support ratios below their thresholds) count as
// low-depth regions of one base when the read depth anywhere within two windows of them is at most d_t.  If there is any, all
// regions -- the depth-derived ones too -- are put in order and every region that starts within ten bin windows of its predecessor's
// end is folded into it (update_ld_regs_with_refqv, ctg_cns.c:2750-2794).  The reference folds in place and leaves the absorbed
// entries behind as (0, 0); the one reader of the list, sv_update_split_p, passes over those, so they are simply not kept here.
void sv_update_ld_regs_with_refqv(std::vector<SvPos>* regs, const std::vector<uint16_t>& depth, const ref_* ref, int32_t w, int32_t win_s, int32_t win_e,
                                  int32_t d_t, uint32_t ide_t, uint32_t ort_t, uint32_t irt_t) {
    const uint32_t reach = (uint32_t)(w * 2);
    const int32_t last_bin = (win_e - win_s) / INS_WIN_STEP;
    auto thin_nearby = [&](uint32_t p) {
        const int32_t b0 = p > reach + (uint32_t)win_s ? (int32_t)((p - reach - (uint32_t)win_s) / INS_WIN_STEP) : 0;
        const int32_t b1 = p + reach < (uint32_t)win_e ? (int32_t)((p + reach - (uint32_t)win_s) / INS_WIN_STEP) : last_bin;
        for (int32_t b = b0; b <= b1; ++b)
            if ((size_t)b < depth.size() && depth[(size_t)b] <= d_t) return true;
        return false;
    };
    size_t added = 0;
    for (uint32_t i = 0; i < ref->qv_l && ref->qv[i].p < (uint32_t)win_e; ++i) {
        const ref_qv& q = ref->qv[i];
        if (q.p < (uint32_t)win_s || q.ide >= ide_t || q.ort >= ort_t || q.irt >= irt_t || !thin_nearby(q.p)) continue;
        regs->push_back(SvPos{q.p - (uint32_t)win_s, q.p + 1 - (uint32_t)win_s});
        ++added;
    }
    if (!added) return;
    std::stable_sort(regs->begin(), regs->end(), [](const SvPos& a, const SvPos& b) {
        return (a.s == b.s ? (int)(a.e - b.e) : (int)(a.s - b.s)) < 0;
    });
    const uint32_t slack = (uint32_t)(INS_WIN_DIV / 2 * w);
    std::vector<SvPos> merged;
    for (const SvPos& r : *regs) {
        if (!merged.empty() && r.s < merged.back().e + slack) merged.back().e = std::max(merged.back().e, r.e);
        else merged.push_back(r);
    }
    regs->swap(merged);
}

// Clusters of split-read gaps = candidate structural differences (update_gap_cluster, ctg_cns.c:2551-2612).  The gaps are swept in
// order of their start.  A gap whose midpoint lies in thin depth (below half the typical depth; and not within the first read
// window) seeds a chain: every later gap that starts no later than the chain's reach belongs to the sweep of this chain, and those
// of them whose own midpoint lies in thin depth join it (their ends extend the reach; at most 120 are remembered).  Note the seed
// itself only counts, it is not a member.  A chain is kept when it has more members than a fifth of the typical depth and its seed
// sits where fewer reads span than gaps agree.  Kept clusters get their members ordered by midpoint and their breakpoint estimate.
int sv_update_gap_cluster(SvWindow* w, int rw, int d, int32_t ref_s) {
    w->clusters.clear();
    if (d < 10) return 0;
    const int min_members = (int)(d * CLUSTER_MIN_DEPTH_RATIO);
    std::stable_sort(w->gaps.begin(), w->gaps.end(), [](const SvGapRead& a, const SvGapRead& b) {
        return a.gap.s != b.gap.s ? a.gap.s < b.gap.s : a.gap.e < b.gap.e;
    });
    const int64_t n = (int64_t)w->gaps.size();
    auto midpoint = [&](int64_t k) { return (int32_t)((w->gaps[(size_t)k].gap.s + w->gaps[(size_t)k].gap.e) / 2) - ref_s; };
    auto spanning_reads = [&](int32_t p) -> int {      // depth track at window coordinate p (bins of ten bases)
        const int64_t bin = p / INS_WIN_STEP;
        return bin >= 0 && (size_t)bin < w->ref_ds.size() ? (int)w->ref_ds[(size_t)bin] : 0;
    };
    auto thin = [&](int64_t k) { return spanning_reads(midpoint(k)) < d / 2; };
    int64_t k = 0;
    while (k + min_members < n) {
        const int32_t seed_at = midpoint(k);
        if (seed_at < rw || !thin(k)) { ++k; continue; }
        SvCluster chain;
        uint32_t reach = w->gaps[(size_t)k].gap.e;
        int agreeing = 1;
        int64_t next = k + 1;
        for (; next < n && (int32_t)w->gaps[(size_t)next].gap.s <= (int32_t)reach; ++next) {
            if (!thin(next)) continue;
            ++agreeing;
            if ((int32_t)w->gaps[(size_t)next].gap.e > (int32_t)reach) reach = w->gaps[(size_t)next].gap.e;
            if (chain.i_m < 2 * LQSEQ_MAX_CAN_COUNT) { chain.gap.push_back((uint32_t)next); ++chain.i_m; }
        }
        if ((int)chain.i_m > min_members && spanning_reads(seed_at) < agreeing) w->clusters.push_back(std::move(chain));
        k = next;
    }
    int members = 0;
    for (SvCluster& c : w->clusters) {
        members += (int)c.i_m;
        std::stable_sort(c.gap.begin(), c.gap.end(), [&](uint32_t a, uint32_t b) {
            return w->gaps[a].gap.s + w->gaps[a].gap.e < w->gaps[b].gap.s + w->gaps[b].gap.e;
        });
        cluster_median(w, &c);
    }
    return members;
}

// Which supplementary alignments of a cluster's split reads become tag streams of their own (update_align_tags, ctg_cns.c:2836-2888):
// the members are taken ring by ring around the breakpoint estimate -- ring r = midpoints within 20 r bases of it, r = 1 .. 14 --
// and a new ring is opened only while fewer than 60 and fewer than 0.8 x members have been taken.  A member whose supplementary
// alignment keeps fewer than 500 window columns never qualifies.  Every taken member remembers its ring, its new stream and where
// that stream starts on the read.
uint32_t sv_update_align_tags(SvWindow* w, const std::vector<SpanOut>& sup_span, uint32_t seq_count, int32_t ref_s, std::vector<StreamRef>* streams) {
    (void)ref_s;
    constexpr uint32_t RING = 20, N_RINGS = 14;
    for (SvCluster& clu : w->clusters) {
        std::vector<uint32_t> by_ring[N_RINGS + 1];
        for (uint32_t j = 0; j < clu.i_m; ++j) {
            const SvGapRead& g = w->gaps[clu.gap[j]];
            const SpanOut& a = sup_span[clu.gap[j]];
            if (g.l || a.aln_t_s > a.aln_t_e - 500u) continue;
            const uint32_t away = mabs((g.gap.s + g.gap.e) / 2, clu.median);
            const uint32_t ring = away == 0 ? 1 : (away + RING - 1) / RING;
            if (ring <= N_RINGS) by_ring[ring].push_back(j);
        }
        uint32_t taken = 0;
        for (uint32_t ring = 1; ring <= N_RINGS && taken < LQSEQ_MAX_CAN_COUNT && taken < clu.i_m * 0.8; ++ring)
            for (uint32_t j : by_ring[ring]) {
                SvGapRead& g = w->gaps[clu.gap[j]];
                StreamRef sr;
                sr.set = 1;
                sr.rec = clu.gap[j];
                sr.span = sup_span[clu.gap[j]];
                streams->push_back(sr);
                g.l = ring;
                g.s_id = seq_count++;
                g.s_s = sr.span.aln_q_s;
                ++taken;
            }
    }
    return seq_count;
}

namespace {

// One cluster's search for the draft interval its split reads bridge, and the piece of every read that lies across it.  The search is
// a small state machine so that ALL clusters of a window advance together: what needs the tag streams -- "which base of its read has
// stream s reached at column c" -- is asked of the executor for every cluster at once (Exec::read_coords, a lane per question on the
// device; round 4 downloaded every tag stream of the window, ~40 MB, and walked them here).
struct GapBridge {
    SvWindow* w;
    SvCluster& clu;
    const WindowOutput& wo;
    uint32_t win_s;
    uint32_t radius = 10;
    bool done = false;
    std::vector<uint32_t> asked;      // members whose two coordinates are in flight, in request order
    uint32_t first_col(uint32_t stream) const { return wo.aln_t_s[stream]; }
    uint32_t last_col(uint32_t stream) const { return wo.aln_t_e[stream] - 1; }     // inclusive, like align_tags_t.aln_t_e

    // members still in play whose primary stream runs across `from` and whose supplementary stream runs across `to`, the two streams
    // meeting nowhere in between; *idle = members no longer in play.  (A member's two streams are kept in draft order.)
    uint32_t bridging(uint32_t from, uint32_t to, uint32_t* idle) {
        uint32_t n = 0;
        *idle = 0;
        for (uint32_t j = 0; j < clu.i_m; ++j) {
            SvGapRead& g = w->gaps[clu.gap[j]];
            if (!g.l) { ++*idle; continue; }
            if (first_col(g.p_id) > first_col(g.s_id)) { std::swap(g.p_id, g.s_id); std::swap(g.p_s, g.s_s); }
            const uint32_t a = g.p_id, b = g.s_id;
            n += first_col(a) < from && last_col(a) > from && first_col(b) < to && last_col(b) > to && from < first_col(b) && to > last_col(a);
        }
        return n;
    }
    // generate_gapseqs for this cluster (ctg_cns.c:2898-2971): widen the interval around the breakpoint estimate in steps of ten
    // while that wins bridging members (or until half of them bridge), keep the last interval that won some; then every member's
    // read substring across it is wanted: the questions go to `req`
    void widen_and_ask(std::vector<CoordReq>* req) {
        if (radius == 10) clu.r.s = clu.r.e = 0;
        uint32_t now = 0, before = 0, idle = 0;
        while (radius < 30000 && before < clu.i_m - idle && (now >= before || before < clu.i_m / 2)) {
            const uint32_t from = clu.median > radius ? clu.median - radius - win_s : 0, to = clu.median + radius - win_s;
            before = now;
            now = bridging(from, to, &idle);
            if (now > before) { clu.r.s = from; clu.r.e = to; }
            radius += 10;
        }
        last_before = before;
        asked.clear();
        for (uint32_t j = 0; j < clu.i_m; ++j) {
            SvGapRead& g = w->gaps[clu.gap[j]];
            if (!g.l) continue;
            const uint32_t a = g.p_id, b = g.s_id;
            if (first_col(a) > clu.r.s || last_col(a) < clu.r.s || first_col(b) > clu.r.e || last_col(b) < clu.r.e) { g.l = 1; continue; }
            asked.push_back(j);
            req->push_back(CoordReq{a, clu.r.s, 1u});             // bases of the primary's read used up through column r.s ...
            req->push_back(CoordReq{b, clu.r.e + 1, 0u});         // ... and of the supplementary's before column r.e + 1
        }
    }
    uint32_t last_before = 0;
    // with the answers: every member's substring; usable = longer than ten bases; if too few are, the search widens further by half
    // the shortest piece + 20 and asks again
    void take(const uint32_t* ans) {
        uint32_t usable = 0, shortest = UINT32_MAX;
        for (size_t k = 0; k < asked.size(); ++k) {
            SvGapRead& g = w->gaps[clu.gap[asked[k]]];
            g.gap.s = g.p_s - 1 + ans[2 * k];
            g.gap.e = g.s_s + ans[2 * k + 1];
            g.l = g.gap.e > g.gap.s + 10 ? 2 : 1;
            usable += g.l == 2;
            shortest = std::min(shortest, mabs(g.gap.s, g.gap.e));
        }
        if (usable >= last_before / 2 || usable >= 10) { done = true; return; }
        radius += shortest / 2 + 20;
    }
};

}  // namespace

bool sv_generate_gapseqs(SvWindow* w, const WindowOutput& wo, int32_t s_, Exec* exec, std::string* err) {
    std::vector<GapBridge> br;
    br.reserve(w->clusters.size());
    for (SvCluster& clu : w->clusters) br.push_back(GapBridge{w, clu, wo, (uint32_t)s_});
    std::vector<CoordReq> req;
    std::vector<uint32_t> ans;
    for (;;) {
        req.clear();
        std::vector<std::pair<size_t, size_t>> part;      // (cluster, first request)
        for (size_t i = 0; i < br.size(); ++i) {
            if (br[i].done) continue;
            part.emplace_back(i, req.size());
            br[i].widen_and_ask(&req);
        }
        if (part.empty()) break;
        if (!exec->read_coords(req, &ans, err)) return false;
        for (auto& pr : part) br[pr.first].take(ans.data() + pr.second);
    }
    // of two neighbouring clusters whose intervals come within 500 bases, the one with fewer members in play goes (ctg_cns.c:2973-2996)
    auto in_play = [&](const SvCluster& c) { int n = 0; for (uint32_t j = 0; j < c.i_m; ++j) n += w->gaps[c.gap[j]].l ? 1 : 0; return n; };
    for (size_t i = 0; i + 1 < w->clusters.size(); ++i) {
        SvCluster& here = w->clusters[i];
        SvCluster& next = w->clusters[i + 1];
        if (!here.i_m || here.r.e + 500 < next.r.s) continue;
        if (in_play(next) > in_play(here)) here.i_m = 0;
        else next.i_m = 0;
    }
    return true;
}

// Where the contig is cut (update_split_p, ctg_cns.c:2999-3051): a low-depth region away from the window's ends that no cluster
// interval touches is a split region (regions within 10 kb of the previous one extend it); inside every split region the
// assembler's quality record with the lowest identity + support sum, if below 2900, becomes the exact split point.
// The clusters are visited with the reference's moving cursor (it backs up one cluster per region and stops at the first cluster
// that starts behind the region), which decides the outcome when cluster intervals are not in order.
void sv_update_split_p(std::vector<SvPos>* split_ps, const SvWindow& w, int32_t s, int32_t l, const ref_* ref) {
    constexpr uint32_t FLANK = 1000;
    auto touches = [](const SvPos& a, const SvPos& b) {
        return (a.s <= b.s && b.s <= a.e) || (a.s <= b.e && b.e <= a.e) || (b.s <= a.s && a.s <= b.e) || (b.s <= a.e && a.e <= b.e);
    };
    size_t cursor = 0;
    for (const SvPos& reg : w.ld_regs) {
        if (reg.s < FLANK || reg.e + FLANK > (uint32_t)l) continue;
        cursor = cursor > 1 ? cursor - 1 : 0;
        bool bridged = false;
        while (cursor < w.clusters.size() && !bridged) {
            const SvPos& iv = w.clusters[cursor].r;
            if (iv.s > reg.e) break;
            bridged = touches(reg, iv);
            ++cursor;
        }
        if (bridged) continue;
        const SvPos abs_reg{reg.s + (uint32_t)s, reg.e + (uint32_t)s};
        if (split_ps->empty() || abs_reg.s > split_ps->back().e + 10000) split_ps->push_back(abs_reg);
        else split_ps->back().e = abs_reg.e;
    }
    for (SvPos& reg : *split_ps) {
        uint32_t worst = 0, at = 0;
        for (uint32_t q = 0; q < ref->qv_l && ref->qv[q].p <= reg.e; ++q) {
            if (ref->qv[q].p < reg.s) continue;
            const uint32_t sum = ref->qv[q].ide + ref->qv[q].ort + ref->qv[q].irt;
            if (worst == 0 || sum < worst) { worst = sum; at = q; }
        }
        if (worst && worst < 2900) reg.s = reg.e = ref->qv[at].p;
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
