// snp_phase, host side of both the product pipeline (np1_device.hip) and the CPU model (tests/model): the two short sequential
// passes over a contig's sites -- link regions (snpphase.c:559-613) and the chain over the sites (snpphase.c:450-557).  The chain
// scores use log10 of the host's libm, like the reference running on the same machine.
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

// ts_find_snp_region: flag != 0 -> groups of sites closer than `gap` (short-read links); flag == 0 -> groups delimited by the
// LEFT / RIGHT marks, measured between anchors (long-read links).  Pairs of local positions.
inline std::vector<int32_t> sp_link_regions(const std::vector<SpHostSite>& s, int32_t gap, uint32_t flag) {
    std::vector<int32_t> out;
    const uint32_t lr_marks = 64u | 128u;
    int qs = -1, qe = -1;
    for (int i = 0; i < (int)s.size(); ++i) {
        const uint32_t f2 = s[i].flag;
        if (!((f2 & flag) || (f2 & lr_marks))) continue;
        if (qs < 0) { qs = qe = i; continue; }
        if (!(flag || (f2 & 128u))) continue;
        const int32_t d = flag ? s[i].pos - s[qe].pos : s[i].right - s[qe].left;
        if (d < gap) { qe = i; continue; }
        if (qs != qe) {
            if (flag) { out.push_back(s[qs].pos); out.push_back(s[qe].pos + 1); }
            else { out.push_back(s[qs].left); out.push_back(s[qe].right); }
        }
        if (flag || (f2 & 64u)) qs = qe = i; else qs = qe = -1;
    }
    if (qs >= 0 && qs != qe) {
        if (flag) { out.push_back(s[qs].pos); out.push_back(s[qe].pos); }   // (no + 1 on the last group, snpphase.c:603-606)
        else { out.push_back(s[qs].left); out.push_back(s[qe].right); }
    }
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

// ts_snps_score + ts_snps_correct: choice[i] = allele (0 / 1) to write at site i, -1 = leave it; false = the reference would
// read a score that does not exist
inline bool sp_chain(const std::vector<SpHostSite>& s, double ploidy, std::vector<int8_t>* choice) {
    const int n = (int)s.size();
    choice->assign((size_t)n, -1);
    if (n <= 1) return true;
    struct St { uint8_t base; uint16_t kmer; double score; };
    std::vector<std::vector<St>> sc((size_t)n);
    auto get = [](std::vector<St>& v, uint32_t base) -> St* { for (auto& x : v) if (x.base == base) return &x; return nullptr; };
    auto add = [&](std::vector<St>& v, uint16_t kmer, double score) {
        St* x = get(v, kmer & 0xfu);
        if (x) { x->kmer = kmer; x->score = score; } else v.push_back(St{(uint8_t)(kmer & 0xf), kmer, score});
    };
    auto best = [](std::vector<St>& v) -> St* {   // first strictly greatest
        St* q = nullptr;
        for (auto& x : v) if (!q || x.score > q->score) q = &x;
        return q;
    };
    std::vector<std::vector<int>> order((size_t)n);   // the link list of a site: combinations in the order they first appeared
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < 4; ++c) if (s[i].num[c] > 0) order[i].push_back(c);
        std::sort(order[i].begin(), order[i].end(), [&](int a, int b) { return s[i].first[a] < s[i].first[b]; });
    }
    add(sc[0], 1, 0); add(sc[0], 2, 0);
    for (int i = 1; i < n; ++i) {
        std::vector<St>&prev = sc[i - 1], &cur = sc[i];
        const double pen = (int16_t)s[i].total / ploidy;
        if (order[i].empty()) { add(cur, 1, 0); add(cur, 2, 0); continue; }
        uint16_t link0[3] = {0, 0, 0}, link1[3] = {0, 0, 0};
        for (int c : order[i]) {
            const uint16_t a = (uint16_t)(c / 2 + 1), b = (uint16_t)(c % 2 + 1), code = (uint16_t)(a << 4 | b);
            St* p0 = get(prev, a);
            if (!p0) return false;
            double score = p0->score;
            score += s[i].num[c] * log10((s[i].mapqual[c] + s[i].qual[c]) / (double)s[i].num[c] + 2) - pen;
            St* ps = get(cur, b);
            if (ps == nullptr || ps->score < score) {
                if (link0[a]) {
                    if (get(cur, link0[a])->score >= score) continue;
                    link1[link0[a]] = 0;
                }
                if (ps != nullptr) link0[ps->kmer >> 4] = 0;
                add(cur, code, score);
                link0[a] = b;
                link1[b] = a;
            }
        }
        int k = 1;
        for (int j = 1; j <= 2; ++j)
            if (link1[j] == 0)
                for (; k <= 2; ++k)
                    if (link0[k] == 0) {
                        St* p0 = get(prev, (uint32_t)k);
                        if (!p0) return false;
                        double score = p0->score;
                        score -= pen;
                        add(cur, (uint16_t)((k << 4) + j), score);
                        break;
                    }
    }
    St* score = nullptr;
    for (int i = n - 1; i > 0; --i) {
        if (order[i].empty()) continue;
        if (score == nullptr) {
            score = best(sc[i]);
            if (!score) return false;
            (*choice)[i] = (int8_t)(score->base - 1);
        }
        const int index = (score->kmer >> 4) - 1;
        if (index < 0 || index > 1) return false;
        (*choice)[i - 1] = (int8_t)index;
        if (!order[i - 1].empty()) {
            score = get(sc[i - 1], (uint32_t)index + 1);
            if (!score) return false;
        } else {
            score = nullptr;
        }
    }
    return true;
}

}  // namespace np1p
