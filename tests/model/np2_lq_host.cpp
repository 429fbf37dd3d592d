// TEST INFRASTRUCTURE: see np2_lq_host.h.
#include "np2_lq_host.h"

#include <cstring>

namespace np2 {

// ---- banded O(ND) alignment (align.c:39-177).  Returns false when no alignment was produced (aln untouched).
bool ond_align(const char* query_seq, int q_len, const char* target_seq, int t_len, OndAln* aln) {
    int max_d = (int)(0.4 * (q_len + t_len));
    const float band_factor = q_len + t_len > 5000 ? 0.1f : 1.0f;
    const int band_size = (int)(band_factor * (float)(q_len + t_len));
    const int k_offset = max_d;
    std::vector<int> V((size_t)2 * (size_t)(max_d + 2) + 4, 0);
    std::vector<std::vector<uint8_t>> D;
    int x = 0, y = 0, kk = 0, min_k = 0, max_k = 0, best_m = -1, k = 0, d;
    bool aligned = false;
    aln->aln_len = 0;
    for (d = 0; d < max_d && max_k - min_k <= band_size; ++d) {
        D.emplace_back((size_t)d + 2, (uint8_t)0);
        for (k = min_k; k <= max_k; k += 2) {
            kk = k < 0 ? -1 * k - 1 : k;
            if ((k == min_k) || ((k != max_k) && (V[(size_t)(k - 1 + k_offset)] < V[(size_t)(k + 1 + k_offset)]))) {
                x = V[(size_t)(k + 1 + k_offset)];
                D[(size_t)d][(size_t)kk] = 0;
            } else {
                x = V[(size_t)(k - 1 + k_offset)] + 1;
                D[(size_t)d][(size_t)kk] = 1;
            }
            y = x - k;
            while (x < q_len && y < t_len && query_seq[x] == target_seq[y]) { ++x; ++y; }
            V[(size_t)(k + k_offset)] = x;
            if (x + y > best_m) best_m = x + y;
            if (x >= q_len && y >= t_len) { aligned = true; break; }
        }
        int new_min_k = max_k, new_max_k = min_k;
        int k2 = min_k;
        while (k2 < new_min_k) {
            if (V[(size_t)(k2 + k_offset)] * 2 - k2 >= best_m - 150) new_min_k = k2;
            k2 += 2;
        }
        k2 = max_k;
        while (k2 > new_max_k) {
            if (V[(size_t)(k2 + k_offset)] * 2 - k2 >= best_m - 150) new_max_k = k2;
            k2 -= 2;
        }
        max_k = new_max_k + 1;
        min_k = new_min_k - 1;
        if (aligned) {
            --x;
            aln->aln_t_len = y;
            aln->aln_q_len = x + 1;
            int gap = 0;
            std::string ts, qs;
            for (;;) {
                while (x >= 0 && x >= k && query_seq[x] == target_seq[x - k]) {
                    ts.push_back(query_seq[x]);
                    qs.push_back(query_seq[x]);
                    --x;
                    gap = 0;
                }
                const int pre_d = d - 1;
                if (x < 0 && x - k < 0) break;
                int pre_k, pre_x;
                if (D[(size_t)d][(size_t)kk]) { pre_k = k - 1; pre_x = x - 1; }
                else { pre_k = k + 1; pre_x = x; }
                const int pre_y = pre_x - pre_k;
                const int pre_kk = pre_k < 0 ? -1 * pre_k - 1 : pre_k;
                if (pre_x == x && pre_y != x - k) {
                    if (x - k < 0) gap = 260;
                    else { qs.push_back('-'); ts.push_back(target_seq[x - k]); }
                } else {
                    if (x < 0) gap = 260;
                    else { qs.push_back(query_seq[x]); ts.push_back('-'); }
                }
                if (gap++ > 250) {   // a gap run longer than 250: give up (the reference leaves two columns of junk, caller tests aln_len > 2)
                    ts.resize(2, '-');
                    qs.resize(2, '-');
                    break;
                }
                d = pre_d;
                k = pre_k;
                kk = pre_kk;
                x = pre_x;
            }
            aln->aln_len = (int)ts.size();
            aln->t_aln_str.assign(ts.rbegin(), ts.rend());
            aln->q_aln_str.assign(qs.rbegin(), qs.rend());
            return true;
        }
    }
    return false;
}


namespace {
constexpr int LQSEQ_MAX_COUNT = LQ_ROUNDS;
// gapped string pair under construction: the reference writes with strcpy at a logical length that one of its fill
// helpers advances by less than it wrote (fill_aln_with_lqseq, ctg_cns.c:1268-1285), so keep position semantics
struct LinkAln {
    std::string t, q;
    size_t len = 0;
    void put(const std::string& ts, const std::string& qs) {   // strcpy both at `len` (does not advance)
        if (t.size() < len + ts.size()) { t.resize(len + ts.size(), '\0'); q.resize(len + ts.size(), '\0'); }
        if (q.size() < len + qs.size()) { t.resize(len + qs.size(), '\0'); q.resize(len + qs.size(), '\0'); }
        t.replace(len, ts.size(), ts);
        q.replace(len, qs.size(), qs);
    }
    void push(char tc, char qc) {
        if (t.size() <= len) { t.resize(len + 1, '\0'); q.resize(len + 1, '\0'); }
        t[len] = tc; q[len] = qc;
        ++len;
    }
};
void fill_with_seed(LinkAln& a, int seed_len) {
    const std::string m((size_t)seed_len, 'M');
    a.put(m, m);
    a.len += (size_t)seed_len;
}
void fill_with_lqseq(LinkAln& a, const std::string& seed, int seed_len, const std::string& lqseq, int lqseq_len) {
    if (lqseq_len > seed_len) a.put(seed.substr(0, (size_t)seed_len) + std::string((size_t)(lqseq_len - seed_len), '-'), lqseq.substr(0, (size_t)lqseq_len));
    else a.put(seed.substr(0, (size_t)seed_len), lqseq.substr(0, (size_t)lqseq_len) + std::string((size_t)(seed_len - lqseq_len), '-'));
    a.len += (size_t)lqseq_len;
}

}  // namespace

void lq_concatenate_host(const LqAlignInput& in, LqInput* out) {
    const size_t count = in.regions.size();
    struct Pieces { std::string t[LQSEQ_MAX_COUNT], q[LQSEQ_MAX_COUNT]; };
    std::vector<Pieces> pieces(count);
    uint32_t t_len = 1;
    for (size_t j = 0; j < count; ++j) {
        const LqAlignRegion& r = in.regions[j];
        const int seed_len = (int)r.seed_len;
        const std::string seed = in.chars.substr(r.seed_off, r.seed_len);
        auto cand = [&](uint32_t i) { return in.chars.substr(in.cand_off[r.first_cand + i], in.cand_len[r.first_cand + i]); };
        t_len += r.seed_len + 1;
        int lqcount = 0;
        for (int i = 0; i < LQSEQ_MAX_COUNT; ++i) {
            LinkAln a;
            const bool beyond = (uint32_t)i >= r.n_cand;
            const int query_len = beyond ? seed_len : (int)in.cand_len[r.first_cand + (uint32_t)i];
            if (beyond) lqcount = 0;
            bool fallback = false;
            if (beyond || (i && (query_len < seed_len * 0.5 || query_len > seed_len * 1.3))) {
                fallback = true;
            } else {
                const std::string cd = cand((uint32_t)i);
                OndAln al;
                ond_align(cd.c_str(), query_len, seed.c_str(), seed_len, &al);
                if (al.aln_len > 2) {
                    a.put(al.t_aln_str, al.q_aln_str);
                    a.len += (size_t)al.aln_len;
                    int tl = al.aln_t_len, ql = al.aln_q_len;
                    while (tl < seed_len) a.push(seed[(size_t)tl++], '-');
                    int delta = 0;
                    while (ql < (int)cd.size() && delta++ < 250) a.push('-', cd[(size_t)ql++]);
                } else {
                    fallback = true;
                }
            }
            if (fallback) {
                if (lqcount++ < (int)r.n_cand - 1) fill_with_seed(a, seed_len);
                else { const std::string c0 = cand(0); fill_with_lqseq(a, seed, seed_len, c0, (int)c0.size()); }
            }
            pieces[j].t[i].assign(a.t, 0, a.len);
            pieces[j].q[i].assign(a.q, 0, a.len);
        }
    }
    out->t.clear();
    out->q.clear();
    for (int i = 0; i < LQSEQ_MAX_COUNT; ++i) {
        std::string t, q;
        for (size_t j = 0; j < count; ++j) {      // the regions arrive in the order of concatenation
            t.push_back('N');
            q.push_back('N');
            t += pieces[j].t[i];
            q += pieces[j].q[i];
        }
        t.push_back('N');
        q.push_back('N');
        out->t.push_back(std::move(t));
        out->q.push_back(std::move(q));
    }
    out->t_len = t_len;
    out->gap_min_len = in.gap_min_len;
    out->hifi = in.hifi;
}

}  // namespace np2
