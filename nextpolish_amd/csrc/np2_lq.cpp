// Low-quality-region stage of a window (ONT / CLR): candidate strings of every region from the tag streams, pseudo
// seed by partial-order alignment, two rounds of "align all candidates to the seeds, concatenate the regions, rerun
// the link-graph consensus on the concatenation" (the graph + DP run in the window executor, i.e. on the GPU), and
// the splice of the accepted seeds back into the window consensus.
//
// Behaviour of: generate_lqseqs_from_tags (source/lib/ctg_cns.c:822-984), count_kmers / count_kscore (:405-449),
// remove_short_lqseq (:620-633), generate_consensus_trimed (:1287-1414), iterate_generate_consensus_trimed
// (:1425-1473), update_consensus_trimed (:1165-1211).  Candidate ordering uses a stable sort: the reference calls
// glibc qsort, which is a merge sort for these sizes.
#include "np2_lq.h"
#include "np_threads.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>

namespace np2 {
namespace {

using np2k::Tag;

constexpr int LQSEQ_MAX_CAN_COUNT = 60, LQSEQ_MAX_COUNT = 30, KMER_RANGE = 40, KMER_LEN = 8, KMER_MAX_SEQ = 10;
constexpr unsigned LQSEQ_MAX_REV_LEN = 2000;

struct Cand {        // struct seq_ (ctg_cns.h:83-89)
    uint16_t order = 0, kscore = 0;
    uint32_t len = 0;
    std::string seq;
};
struct Region {      // lqseq (ctg_cns.h:73-90)
    uint8_t l = 0, indexs = 0, indexe = 0;
    int len = 0;
    unsigned lqcount = 0, start = 0, end = 0, sudoseed_len = 0;
    std::string sudoseed;
    std::vector<Cand> seqs;
    int poa_first = -1, poa_n = 0;      // the pseudo-seed is still to be made: partial-order consensus of seqs[poa_first .. + poa_n)
    bool counts = false;                // the region takes part in the longest-alignment bound of the round
};

// ---- candidate ranking (what the reference does in ctg_cns.c:405-449,620-633,880-960, reformulated) -------------------------------
//
// A candidate is scored by how well the 8-mers of its first 40 bases (and, when the region is long, of its last 40) are
// shared with the other candidates: score = sum over its window's 8-mers of how often that 8-mer occurs in the windows of
// the region's candidates.  The reference fills a 65 536-bin table per region; here the occurrences are sorted once and a
// candidate's 8-mers are looked up by binary search (<= 60 candidates x 32 8-mers).  Scores are 16-bit and wrap like the
// reference's counters.  Order matters everywhere below (stable sorts, "last of equals" cuts), so the candidates are ranked
// through a permutation and moved once at the end.

// 2-bit codes of the 8-mers of a candidate's head (or tail) window, in window order.  The window holds min(len, 40) bases
// and contributes that many minus 8 codes (the loop bound of the reference: the last 8-mer of the window is not used).
void window_codes(const Cand& c, bool tail, std::vector<uint16_t>* out) {
    out->clear();
    if (c.len < (uint32_t)KMER_LEN) return;
    const uint32_t span = std::min<uint32_t>(c.len, KMER_RANGE);
    const uint32_t first = tail && c.len > (uint32_t)KMER_RANGE ? c.len - KMER_RANGE : 0;
    if (span <= (uint32_t)KMER_LEN) return;
    uint16_t code = 0;
    for (uint32_t p = 0; p < (uint32_t)KMER_LEN; ++p) code = (uint16_t)(code << 2 | np2k::base_to_int((unsigned char)c.seq[first + p]));
    out->push_back(code);
    for (uint32_t k = 1; k < span - KMER_LEN; ++k) {
        code = (uint16_t)(code << 2 | np2k::base_to_int((unsigned char)c.seq[first + k + KMER_LEN - 1]));
        out->push_back(code);
    }
}

struct Ranker {
    Region& r;
    std::vector<uint16_t> at;        // at[i] = index into r.seqs of the candidate ranked i; the first `live` are in play
    int live;
    explicit Ranker(Region& reg) : r(reg), live(reg.len) {
        at.resize((size_t)std::max(reg.len, 0));
        for (size_t i = 0; i < at.size(); ++i) at[i] = (uint16_t)i;
    }
    uint32_t len_at(int i) const { return r.seqs[at[(size_t)i]].len; }
    template <class Less> void stable_by(Less less) {
        std::stable_sort(at.begin(), at.begin() + live, [&](uint16_t a, uint16_t b) { return less(r.seqs[a], r.seqs[b]); });
    }
    void flip() { std::reverse(at.begin(), at.begin() + std::max(live, 0)); }
    // moves the candidates into ranked order (the rest of the stage indexes r.seqs by rank)
    void commit() {
        std::vector<Cand> tmp(r.seqs.size());
        for (size_t i = 0; i < at.size(); ++i) tmp[i] = std::move(r.seqs[at[i]]);
        for (size_t i = at.size(); i < r.seqs.size(); ++i) tmp[i] = std::move(r.seqs[i]);
        r.seqs.swap(tmp);
        r.len = live;
    }

    // k-mer support of every candidate in play; `tail` selects the window.  Occurrences are counted in a small open-addressing table
    // (<= 60 voters x 32 codes in 4096 slots; round 3 sorted the codes and searched them: ten times the work for the same counts).
    void add_support(bool tail, std::vector<uint32_t>* total) {
        constexpr uint32_t SLOTS = 4096, EMPTY = 0xffffffffu;
        static thread_local std::vector<uint32_t> key, cnt;          // key = the 16-bit code, or EMPTY
        static thread_local std::vector<uint16_t> touched, mine;
        if (key.empty()) { key.assign(SLOTS, EMPTY); cnt.assign(SLOTS, 0); }
        touched.clear();
        auto slot_of = [&](uint16_t c) {
            uint32_t h = ((uint32_t)c * 40503u >> 4) & (SLOTS - 1);
            while (key[h] != EMPTY && key[h] != c) h = (h + 1) & (SLOTS - 1);
            return h;
        };
        const int voters = std::min(live, LQSEQ_MAX_CAN_COUNT);
        for (int i = 0; i < voters; ++i) {
            window_codes(r.seqs[at[(size_t)i]], tail, &mine);
            for (uint16_t c : mine) {
                const uint32_t h = slot_of(c);
                if (key[h] == EMPTY) { key[h] = c; cnt[h] = 0; touched.push_back((uint16_t)h); }
                ++cnt[h];
            }
        }
        for (int i = 0; i < live; ++i) {
            window_codes(r.seqs[at[(size_t)i]], tail, &mine);
            uint32_t sum = 0;
            for (uint16_t c : mine) {
                const uint32_t h = slot_of(c);
                if (key[h] != EMPTY) sum += cnt[h];
            }
            (*total)[(size_t)i] += sum & 0xffffu;      // each pass is a 16-bit counter of its own
        }
        for (uint16_t h : touched) key[h] = EMPTY;
    }
};

// Regions found by deletions: the shortest candidates are cut while they are outliers against the upper quartile or against
// their neighbour, at most 30 stay, shortest first (ctg_cns.c:620-633).
void remove_short(Region& r) {
    Ranker k(r);
    k.stable_by([](const Cand& a, const Cand& b) { return a.len > b.len; });
    const int quart = k.live / 4;
    while (k.live > quart && (k.len_at(k.live - 1) < k.len_at(quart) / 2 || k.len_at(k.live - 1) * 1.4 < k.len_at(k.live - 2))) --k.live;
    if (quart == k.live) k.live = 0;
    if (k.live > LQSEQ_MAX_COUNT) k.live = LQSEQ_MAX_COUNT;
    k.flip();
    k.commit();
}

// Length outliers out (optional), support scores, best-supported candidates first, how many of them are aligned later
// (indexs .. indexe) and which ones seed the partial-order consensus.  min_span: indexe - indexs must exceed it (3, or 1 in
// the HiFi variant).  Returns false when the region is dropped (r.len = 0).
static std::atomic<long long> g_prof_rank{0}, g_prof_poa{0};
static inline long long prof_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1000000000ll + ts.tv_nsec; }

bool rank_and_seed(Region& r, bool trim, int min_span, bool defer_poa) {
    const long long t_begin = prof_now();
    Ranker k(r);
    if (trim) {
        k.stable_by([](const Cand& a, const Cand& b) { return a.len < b.len; });
        int mid = k.live / 2;
        while (k.live > mid && (k.len_at(k.live - 1) > 2 * k.len_at(mid) || k.len_at(k.live - 1) >= 1.4 * k.len_at(k.live - 2))) --k.live;   // too long
        if (mid == k.live) { r.len = 0; return false; }
        mid = k.live / 2;
        if (k.len_at(0) < k.len_at(mid) / 2) {        // too short: they are at the front, turn the list over and cut from the back
            k.flip();
            while (k.len_at(k.live - 1) < k.len_at(mid) / 2) --k.live;
            if (mid == k.live) { r.len = 0; return false; }
        }
    }
    std::vector<uint32_t> support((size_t)k.live, 0);
    k.add_support(false, &support);
    if (k.len_at(0) > 100) k.add_support(true, &support);
    for (int i = 0; i < k.live; ++i) r.seqs[k.at[(size_t)i]].kscore = (uint16_t)support[(size_t)i];
    k.stable_by([](const Cand& a, const Cand& b) { return a.kscore > b.kscore; });
    k.commit();
    // how far down the ranking the support holds up: stop at a tenth of the best score, at half of the previous one, or at 30
    const unsigned top_score = r.seqs[0].kscore;
    unsigned prev_score = top_score, longest = r.seqs[0].len;
    int n_good = 0;
    for (; n_good < r.len; ++n_good) {
        const Cand& c = r.seqs[(size_t)n_good];
        if ((unsigned)c.kscore * 10 < top_score || n_good >= LQSEQ_MAX_COUNT || (unsigned)c.kscore * 2 < prev_score) break;
        prev_score = c.kscore;
        if (n_good < KMER_MAX_SEQ && c.kscore > top_score * 0.8 && c.len > longest) longest = c.len;
    }
    r.indexs = 0;
    r.indexe = (uint8_t)(longest > LQSEQ_MAX_REV_LEN && n_good > 6 ? 5 : n_good - 1);
    if (r.indexe - r.indexs <= min_span || (r.seqs[0].len > 20000 && r.len < LQSEQ_MAX_CAN_COUNT / 3)) { r.len = 0; return false; }
    // pseudo-seed: partial-order consensus of the best six (two for long regions), the best candidate itself beyond 20 kb
    const int first = r.indexs;
    const int want = r.seqs[0].len < 3000 ? 6 : 2;
    const int n_poa = first + want < r.indexe ? want : r.indexe - first + 1;
    if (r.seqs[0].len < 20000) {
        if (defer_poa) {          // all regions of the window go to the executor in one batch (collect_candidates)
            r.poa_first = first;
            r.poa_n = n_poa;
            g_prof_rank += prof_now() - t_begin;
            return true;
        }
        std::vector<std::string> v;
        for (int q = 0; q < n_poa; ++q) v.push_back(r.seqs[(size_t)(first + q)].seq);
        const long long t_poa = prof_now();
        r.sudoseed = poa_consensus(v);
        g_prof_poa += prof_now() - t_poa;
    } else {
        r.sudoseed = r.seqs[0].seq;
    }
    r.sudoseed_len = (unsigned)r.sudoseed.size();
    g_prof_rank += prof_now() - t_begin;
    return true;
}

// generate_lqseqs_from_tags (kmer = false) / generate_lqseqs_from_tags_kmer (HiFi, ctg_cns.c:636-820); returns max_aln_length
int collect_candidates(Exec* exec, std::vector<Region>& lq, const WindowOutput& wo, bool kmer, const std::vector<LqCluster>& clusters, std::string* err) {
    const int count = (int)lq.size();
    if (getenv("NP2_TIMING")) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "[np2 lq]   collect start (t=%.2f)\n", ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6); }
    for (Region& r : lq) {
        r.sudoseed.clear();
        r.lqcount = 0; r.len = 0; r.sudoseed_len = 0;
        r.poa_first = -1; r.poa_n = 0; r.counts = false;
        r.seqs.clear();   // grown on acceptance (at most LQSEQ_MAX_CAN_COUNT)
    }
    // every (stream, region) pair with the region inside the stream's span, in the reference's visiting order (streams
    // ascending, regions from the lowest position up); the strings come from the executor (tag streams in HBM)
    struct Visit { uint32_t stream; int hi, lo; size_t req0; };   // regions lq[hi] .. lq[lo] (hi >= lo), requests req0 ..
    std::vector<Visit> visits;
    std::vector<SubReq> req;
    {
        int s = count - 1;
        for (uint32_t i = 1; i < wo.seq_count; ++i) {
            const uint32_t ts = wo.aln_t_s[i], te = wo.aln_t_e[i] - 1;   // the reference's aln_t_e of a stream is inclusive
            while (s >= 0 && lq[(size_t)s].start < ts) --s;
            int j = s;
            for (; j >= 0 && lq[(size_t)j].end <= te; --j) {}
            if (j == s) continue;
            visits.push_back(Visit{i, s, j + 1, req.size()});
            for (int k = s; k > j; --k) req.push_back(SubReq{i, lq[(size_t)k].start, lq[(size_t)k].end});
        }
    }
    std::vector<uint32_t> off;
    std::string bases;
    if (!exec->extract(req, &off, &bases, err)) return -1;
    // the acceptance rules and the 60-candidate cap depend on the visiting order: replayed here (a region that is full
    // is skipped whether it sits on top of the range or inside it, ctg_cns.c:838,852)
    for (const Visit& v : visits) {
        for (int k = v.hi; k >= v.lo; --k) {
            Region& r = lq[(size_t)k];
            if (r.len >= LQSEQ_MAX_CAN_COUNT) continue;
            const size_t rq = v.req0 + (size_t)(v.hi - k);
            const uint32_t index = off[rq + 1] - off[rq];
            if (kmer ? index != 0 : ((r.l && index) || index > r.end - r.start + 1)) {
                if (r.seqs.size() <= (size_t)r.len) r.seqs.emplace_back();
                Cand& cd = r.seqs[(size_t)r.len];
                cd.seq.assign(bases, off[rq], index);
                cd.len = index;
                cd.order = (uint16_t)r.len;
                if (index > r.lqcount) r.lqcount = index;
                ++r.len;
            } else {
                ++r.sudoseed_len;
            }
        }
    }
    if (getenv("NP2_TIMING")) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "[np2 lq]   tag walk done (t=%.2f)\n", ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6); }
    // a gap cluster's region: the split reads' substrings join the candidates (ctg_cns.c:585-600,687-695,871-879); the
    // cluster cursor is shared by the regions, so this part stays in region order
    int clusters_i = (int)clusters.size() - 1;
    for (int i = 0; i < count; ++i) {
        Region& r = lq[(size_t)i];
        if (r.l != 1) continue;
        while (clusters_i >= 0 && !clusters[(size_t)clusters_i].i_m) --clusters_i;
        if (clusters_i < 0) continue;
        const LqCluster& c = clusters[(size_t)clusters_i--];
        for (const std::string& s : c.cands) {
            if (r.len >= LQSEQ_MAX_CAN_COUNT) break;
            if (r.seqs.size() <= (size_t)r.len) r.seqs.emplace_back();
            Cand& cd = r.seqs[(size_t)r.len];
            cd.seq = s;
            cd.len = (uint32_t)s.size();
            cd.order = (uint16_t)r.len;
            if (cd.len > r.lqcount) r.lqcount = cd.len;
            ++r.len;
        }
    }
    // ranking: the regions are independent of each other; the partial-order pseudo-seeds are made afterwards, all at once, by the
    // executor (one wave per region on the device, np2_poa_dev.h)
    np::parallel_for((size_t)count, 16, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            Region& r = lq[i];
            if (kmer) {
                if (!r.len) continue;
                // identical candidates vote: a dominant (or the only short) string is taken as it is (ctg_cns.c:719-737)
                int8_t used[LQSEQ_MAX_CAN_COUNT] = {0};
                int s = 0;
                for (int j = 0; j < r.len; ++j) {
                    r.seqs[(size_t)j].kscore = 1;
                    if (used[j]) continue;
                    for (int k = j + 1; k < r.len; ++k)
                        if (r.seqs[(size_t)j].seq == r.seqs[(size_t)k].seq) { used[k] = 1; ++r.seqs[(size_t)j].kscore; }
                    if (r.seqs[(size_t)j].kscore > r.seqs[(size_t)s].kscore ||
                        (r.seqs[(size_t)j].kscore == r.seqs[(size_t)s].kscore && r.seqs[(size_t)j].len > r.seqs[(size_t)s].len)) s = j;
                }
                const Cand& top = r.seqs[(size_t)s];
                if ((top.kscore > r.len / 3 || top.len < 10 || r.len <= 4) && (top.kscore != 1 || (r.len != 3 && r.len != 4))) {
                    r.sudoseed = top.seq;
                    r.sudoseed_len = top.len;
                    r.len = -2;
                    r.l = 4;
                } else if (!rank_and_seed(r, r.len > 4, 1, true)) {
                    continue;
                }
                r.counts = true;
                continue;
            }
            if (r.l != 1 && r.l > 1 && r.len > 4) remove_short(r);
            if (r.len <= 4 || r.len < r.sudoseed_len * 0.5) { r.len = 0; continue; }
            if (!rank_and_seed(r, true, 3, true)) continue;
            r.counts = true;
        }
    });
    {
        PoaBatch pb;
        std::vector<int> owner;
        for (int i = 0; i < count; ++i) {
            Region& r = lq[(size_t)i];
            if (r.poa_first < 0) continue;
            owner.push_back(i);
            pb.job_first.push_back((uint32_t)pb.str_off.size());
            pb.job_n.push_back((uint32_t)r.poa_n);
            for (int q = 0; q < r.poa_n; ++q) {
                const std::string& sq = r.seqs[(size_t)(r.poa_first + q)].seq;
                pb.str_off.push_back((uint32_t)pb.chars.size());
                pb.str_len.push_back((uint32_t)sq.size());
                pb.chars.append(sq);
                pb.chars.push_back('\0');
            }
        }
        if (!owner.empty()) {
            const long long t_poa = prof_now();
            std::vector<std::string> seeds;
            if (!exec->run_poa(pb, &seeds, err)) return -1;
            g_prof_poa += prof_now() - t_poa;
            for (size_t k = 0; k < owner.size(); ++k) {
                Region& r = lq[(size_t)owner[k]];
                r.sudoseed.swap(seeds[k]);
                r.sudoseed_len = (unsigned)r.sudoseed.size();
                r.poa_first = -1;
            }
        }
    }
    std::atomic<int> max_aln{0};
    for (int i = 0; i < count; ++i) {
        const Region& r = lq[(size_t)i];
        if (r.counts && (int)(r.lqcount + r.sudoseed_len) > max_aln.load()) max_aln = (int)(r.lqcount + r.sudoseed_len);
    }
    const int max_aln_length = max_aln.load();
    if (getenv("NP2_TIMING")) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "[np2 lq]   rank + poa done (t=%.2f); ranking %.1f ms CPU, pseudo-seeds (executor) %.1f ms\n", ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6, g_prof_rank.exchange(0) * 1e-6, g_prof_poa.exchange(0) * 1e-6); }
    return max_aln_length;
}

// generate_consensus_trimed (ctg_cns.c:1287-1414): every valid region hands its current seed and its ranked candidates to the
// executor, which aligns the candidates to the seed, builds the 30 concatenated gapped string pairs and runs the graph
// consensus on them -- all on the device (np2_exec_hip.hip:run_lq_aligned).  Regions go in the order of concatenation.
bool consensus_of_regions(Exec* exec, std::vector<Region>& lq, uint32_t gap_min_len, bool hifi, std::string* cons_rev, std::string* err) {
    LqAlignInput in;
    in.gap_min_len = gap_min_len;
    in.hifi = hifi;
    size_t need = 0;
    for (const Region& r : lq)
        if (r.len > 0) { need += r.sudoseed_len; for (int k = r.indexs; k <= r.indexe && k < r.indexs + LQSEQ_MAX_COUNT; ++k) need += r.seqs[(size_t)k].len; }
    in.chars.reserve(need + 16);
    for (size_t j = lq.size(); j-- > 0;) {
        Region& r = lq[j];
        r.lqcount = 0;
        if (r.len <= 0) continue;
        LqAlignRegion a;
        a.seed_off = (uint32_t)in.chars.size();
        a.seed_len = (uint32_t)r.sudoseed_len;
        in.chars.append(r.sudoseed, 0, (size_t)r.sudoseed_len);
        in.chars.resize((size_t)a.seed_off + a.seed_len, '\0');
        a.first_cand = (uint32_t)in.cand_off.size();
        a.n_cand = (uint32_t)(r.indexe - r.indexs + 1);
        for (int k = r.indexs; k <= r.indexe && k < r.indexs + LQSEQ_MAX_COUNT; ++k) {
            const Cand& cd = r.seqs[(size_t)k];
            in.cand_off.push_back((uint32_t)in.chars.size());
            in.cand_len.push_back((uint32_t)cd.len);
            const size_t at = in.chars.size();
            in.chars.append(cd.seq, 0, (size_t)cd.len);
            in.chars.resize(at + (size_t)cd.len, '\0');
        }
        in.regions.push_back(a);
    }
    if (in.regions.empty()) { cons_rev->assign("N"); return true; }
    return exec->run_lq_aligned(in, cons_rev, err);
}

uint32_t min_cand_len(const Region& r) {
    uint32_t t = r.seqs[0].len;
    for (int i = 1; i < r.len; ++i) t = std::min(t, r.seqs[(size_t)i].len);
    return t;
}

}  // namespace

bool lq_stage(Exec* exec, uint32_t gap_min_len, bool hifi, const std::vector<LqRegionIn>& regions, const std::vector<LqCluster>& clusters,
              const WindowOutput& wo, std::vector<ConsBase>* cons, std::string* err) {
    const int count = (int)regions.size();
    std::vector<Region> lq((size_t)count);
    for (int i = 0; i < count; ++i) { lq[(size_t)i].start = regions[(size_t)i].start; lq[(size_t)i].end = regions[(size_t)i].end; lq[(size_t)i].l = regions[(size_t)i].l; }
    const bool timing = getenv("NP2_TIMING") != nullptr;
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    double t0 = now();
    if (collect_candidates(exec, lq, wo, hifi, clusters, err) < 0) return false;
    if (timing) { const double t = now(); fprintf(stderr, "[np2 lq] candidates+poa %.2f ms\n", t - t0); t0 = t; }
    // ---- iterate_generate_consensus_trimed (two rounds)
    for (int it = 1; it <= 2; ++it) {
        std::string cr;
        if (!consensus_of_regions(exec, lq, gap_min_len, hifi, &cr, err)) return false;
        if (timing) { const double t = now(); fprintf(stderr, "[np2 lq] round %d (align + graph consensus) %.2f ms\n", it, t - t0); t0 = t; }
        int j = count;
        for (size_t k = cr.size(); k; --k) {
            const char c = cr[k - 1];
            if (c != 'N') {
                if (j < 0 || j >= count) continue;   // (the reference would write through a stale index here; cannot happen: the string starts with 'N')
                Region& r = lq[(size_t)j];
                if (c < 'a') r.sudoseed.push_back(c);
                else { r.sudoseed.push_back((char)toupper(c)); ++r.lqcount; }
                ++r.sudoseed_len;
            } else {
                if (j != count && j >= 0) {
                    Region& r = lq[(size_t)j];
                    if (((r.sudoseed_len <= r.end - r.start + 1 || r.lqcount > r.sudoseed_len * 4 / 5) && !r.l) ||
                        (r.l && r.sudoseed_len * 1.3 < min_cand_len(r))) r.len = -1;
                }
                --j;
                while (j >= 0 && lq[(size_t)j].len <= 0) --j;
                if (j < 0) continue;
                lq[(size_t)j].sudoseed.clear();
                lq[(size_t)j].sudoseed_len = 0;
                lq[(size_t)j].lqcount = 0;
            }
        }
    }
    // ---- update_consensus_trimed (ctg_cns.c:1165-1211): regions are in descending position order
    // the spliced consensus is built in a buffer this thread keeps and then swapped with the window's: the two 40 MB
    // vectors of a 5 Mb window change places from call to call instead of being allocated and paged in every time
    static thread_local std::vector<ConsBase> spare;
    std::vector<ConsBase>& out = spare;
    out.clear();
    out.reserve(cons->size() + 1024);
    int lqi = count - 1;
    int update = 1;
    for (size_t i = 0; i < cons->size(); ++i) {
        const uint32_t p = (*cons)[i].pos;
        if (lqi >= 0 && ((lq[(size_t)lqi].len <= 0 && lq[(size_t)lqi].len != -2) || p > lq[(size_t)lqi].end)) {
            --lqi;
            update = 1;
        }
        if (lqi >= 0 && (lq[(size_t)lqi].len > 0 || lq[(size_t)lqi].len == -2) && p >= lq[(size_t)lqi].start && p <= lq[(size_t)lqi].end) {
            if (update) {
                const Region& r = lq[(size_t)lqi];
                for (unsigned q = 0; q < r.sudoseed_len; ++q) out.push_back(ConsBase{r.start, 0, r.sudoseed[q]});
                update = 0;
            }
        } else {
            out.push_back((*cons)[i]);
            update = 1;
        }
    }
    cons->swap(out);
    return true;
}

}  // namespace np2
