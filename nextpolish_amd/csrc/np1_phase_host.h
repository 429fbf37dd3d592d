// snp_phase, host side of both the product pipeline (np1_device.hip) and the CPU model (tests/model): the two short sequential
// passes over a contig's sites -- which stretches are searched for read links, and the two-state chain over the sites.  Both give
// what the reference's ts_find_snp_region / ts_snps_score / ts_snps_correct give (snpphase.c:450-613); the formulations are this
// file's own (see the comments at the functions).
#pragma once
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

namespace np1p {

struct SpHostSite {
    int32_t pos, left, right, len;
    uint8_t flag;                     // marks of the site's main slot
    int32_t total;                    // Snps.total (an int16 in the reference)
    int32_t num[4], mapqual[4], qual[4];
    unsigned long long first[4];
};

// low-depth regions that touch (same contig, one starts on the base the previous one ends on) form a group: first[k] .. first[k + 1]
inline std::vector<uint32_t> sp_region_groups(const std::vector<uint32_t>& reg_ctg, const std::vector<int32_t>& reg_se) {
    std::vector<uint32_t> first;
    for (size_t k = 0; k < reg_ctg.size(); ++k)
        if (k == 0 || reg_ctg[k] != reg_ctg[k - 1] || reg_se[2 * k] > reg_se[2 * k - 1]) first.push_back((uint32_t)k);
    first.push_back((uint32_t)reg_ctg.size());
    return first;
}

// Link regions = stretches of the contig whose records are searched for pairs of sites (reference: ts_find_snp_region,
// snpphase.c:559-613).  Two kinds, both as pairs of local positions:
//   short-read links (flag != 0): the candidate sites are those carrying `flag` or a LEFT / RIGHT mark; consecutive candidates less
//     than `gap` bases apart share a region [first site, last site + 1) -- the last region of the contig ends ON its last site
//     (snpphase.c:603-606) -- and a site on its own makes none;
//   long-read links (flag == 0): only LEFT / RIGHT marked sites matter.  A region is opened by any marked site and grown by
//     RIGHT-marked ones while the stretch from the previous site's left anchor to the new site's right anchor stays below `gap`;
//     when it does not, the region [left anchor of its first site, right anchor of its last) is closed (if it has two sites) and
//     the RIGHT-marked site that did not fit opens the next one only if it is LEFT-marked too.
inline std::vector<int32_t> sp_link_regions(const std::vector<SpHostSite>& s, int32_t gap, uint32_t flag) {
    constexpr uint32_t LEFT = 64u, RIGHT = 128u;
    std::vector<int32_t> out;
    int head = -1, tail = -1;                       // the open region's first and last site
    auto close = [&](bool last_of_contig) {
        if (head < 0 || head == tail) return;
        if (flag) { out.push_back(s[(size_t)head].pos); out.push_back(s[(size_t)tail].pos + (last_of_contig ? 0 : 1)); }
        else { out.push_back(s[(size_t)head].left); out.push_back(s[(size_t)tail].right); }
    };
    for (int i = 0; i < (int)s.size(); ++i) {
        const uint32_t marks = s[(size_t)i].flag;
        const bool candidate = flag ? (marks & (flag | LEFT | RIGHT)) != 0 : (marks & (LEFT | RIGHT)) != 0;
        if (!candidate) continue;
        if (head < 0) { head = tail = i; continue; }
        if (!flag && !(marks & RIGHT)) continue;     // long-read regions grow at RIGHT-marked sites only
        const int32_t stretch = flag ? s[(size_t)i].pos - s[(size_t)tail].pos : s[(size_t)i].right - s[(size_t)tail].left;
        if (stretch < gap) { tail = i; continue; }
        close(false);
        head = tail = (flag || (marks & LEFT)) ? i : -1;
    }
    close(true);
    return out;
}

// the marks after the short-read links (snpphase.c:383-393): returns (local position, marks) pairs to OR into the main slots
inline std::vector<std::pair<int32_t, uint8_t>> sp_link_marks(const std::vector<SpHostSite>& s, int32_t min_count_snp_link) {
    std::vector<std::pair<int32_t, uint8_t>> m;
    for (size_t i = 1; i < s.size(); ++i)
        if ((int16_t)s[i].total <= min_count_snp_link) {
            m.push_back({s[i - 1].left, 64}); m.push_back({s[i - 1].pos, 64}); m.push_back({s[i - 1].right, 128});
            m.push_back({s[i].left, 64}); m.push_back({s[i].pos, 128}); m.push_back({s[i].right, 128});
        }
    return m;
}

// The chain over the sites of a contig (reference: ts_snps_score + ts_snps_correct, snpphase.c:450-557) is a two-state Viterbi pass
// with a matching constraint.  States of site i: its two alleles.  A link combination (allele a of site i - 1 with allele b of site i)
// seen by `num` records is an edge of weight  num * log10((mapq sum + quality sum) / num + 2) - total / ploidy;  the edges chosen
// between two neighbouring sites must pair the alleles one to one.  The reference builds that pairing greedily, edge by edge in the
// order the combinations first appeared in the records, and the result depends on that order and on its comparisons (an edge takes an
// allele only by STRICTLY beating the score the allele holds; an edge whose source is already paired gives way unless it STRICTLY
// beats the score its source's partner holds) -- so the pass below is that greedy, stated on explicit partner arrays:
//   offer(a -> b, score): refused unless it beats b's score; if a is paired with some b', refused unless it also beats b''s score,
//   else b' is orphaned (it keeps its stale score until the end of the site); b's former source is set free; a and b pair up.
//   afterwards every orphaned / unreached allele of site i takes the lowest free allele of site i - 1 at that allele's score minus
//   the site penalty.  A site without links restarts the chain (both alleles at 0, no source).
// Backwards, from the last linked site: the best allele (first strictly greatest, in the order the alleles were first scored), then
// the sources; a site without links ends a stretch and the next linked site starts again from its best allele.
// choice[i] = allele (0 / 1) to write at site i, -1 = leave it; false = the reference would read a score that does not exist.
// log10 is the host libm's, like the reference running on the same machine.
inline bool sp_chain(const std::vector<SpHostSite>& s, double ploidy, std::vector<int8_t>* choice) {
    const int n = (int)s.size();
    choice->assign((size_t)n, -1);
    if (n <= 1) return true;
    struct Column {
        double score[3] = {0, 0, 0};     // by allele 1 / 2
        uint8_t source[3] = {0, 0, 0};   // allele of the previous site it continues (0: none)
        uint8_t scored[2] = {0, 0};      // alleles in the order they were first scored
        uint8_t n_scored = 0;
        bool linked = false;             // the site has link combinations
        bool has(uint32_t allele) const { return (n_scored > 0 && scored[0] == allele) || (n_scored > 1 && scored[1] == allele); }
        void set(uint32_t allele, uint32_t from, double v) {
            if (!has(allele)) scored[n_scored++] = (uint8_t)allele;
            score[allele] = v;
            source[allele] = (uint8_t)from;
        }
        int best() const {               // first strictly greatest
            int q = 0;
            for (uint8_t k = 0; k < n_scored; ++k)
                if (!q || score[scored[k]] > score[q]) q = scored[k];
            return q;
        }
    };
    std::vector<Column> col((size_t)n);
    col[0].set(1, 0, 0.0);
    col[0].set(2, 0, 0.0);
    for (int i = 1; i < n; ++i) {
        const SpHostSite& site = s[(size_t)i];
        Column& cur = col[(size_t)i];
        const Column& prev = col[(size_t)i - 1];
        int combos[4], n_combos = 0;     // the combinations that occurred, in the order the records first showed them
        for (int c = 0; c < 4; ++c)
            if (site.num[c] > 0) combos[n_combos++] = c;
        std::sort(combos, combos + n_combos, [&](int x, int y) { return site.first[x] < site.first[y]; });
        cur.linked = n_combos > 0;
        if (!cur.linked) { cur.set(1, 0, 0.0); cur.set(2, 0, 0.0); continue; }
        const double penalty = (int16_t)site.total / ploidy;
        uint8_t partner_of_prev[3] = {0, 0, 0}, partner_of_cur[3] = {0, 0, 0};
        for (int t = 0; t < n_combos; ++t) {
            const int c = combos[t];
            const uint32_t a = (uint32_t)(c / 2 + 1), b = (uint32_t)(c % 2 + 1);
            if (!prev.has(a)) return false;
            double offer = prev.score[a];
            offer += site.num[c] * log10((site.mapqual[c] + site.qual[c]) / (double)site.num[c] + 2) - penalty;
            if (cur.has(b) && !(cur.score[b] < offer)) continue;                       // b keeps what it has
            if (partner_of_prev[a]) {
                if (cur.score[partner_of_prev[a]] >= offer) continue;                  // a stays with its partner
                partner_of_cur[partner_of_prev[a]] = 0;                                // the partner is orphaned
            }
            if (cur.has(b)) partner_of_prev[cur.source[b]] = 0;                        // b's former source is free again
            cur.set(b, a, offer);
            partner_of_prev[a] = (uint8_t)b;
            partner_of_cur[b] = (uint8_t)a;
        }
        uint32_t free_prev = 1;
        for (uint32_t b = 1; b <= 2; ++b) {
            if (partner_of_cur[b]) continue;
            while (free_prev <= 2 && partner_of_prev[free_prev]) ++free_prev;
            if (free_prev > 2) break;
            if (!prev.has(free_prev)) return false;
            cur.set(b, free_prev, prev.score[free_prev] - penalty);
            // (the reference does not mark the pair in its tables either: the next orphan starts looking at the same allele)
        }
    }
    int at = 0;                           // allele the walk holds at site i (0: about to start from the best one)
    for (int i = n - 1; i > 0; --i) {
        const Column& cur = col[(size_t)i];
        if (!cur.linked) continue;
        if (!at) {
            at = cur.best();
            if (!at) return false;
            (*choice)[(size_t)i] = (int8_t)(at - 1);
        }
        const int from = cur.source[at];
        if (from < 1 || from > 2) return false;
        (*choice)[(size_t)i - 1] = (int8_t)(from - 1);
        if (col[(size_t)i - 1].linked) {
            if (!col[(size_t)i - 1].has((uint32_t)from)) return false;
            at = from;
        } else {
            at = 0;
        }
    }
    return true;
}

}  // namespace np1p
