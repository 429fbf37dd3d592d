// Low-quality-region stage of a window (ONT / CLR): candidate strings of every region from the tag streams, pseudo
// seed by partial-order alignment, two rounds of "align all candidates to the seeds, concatenate the regions, rerun
// the link-graph consensus on the concatenation" (the graph + DP run in the window executor, i.e. on the GPU), and
// the splice of the accepted seeds back into the window consensus.
//
// Restated from: generate_lqseqs_from_tags (source/lib/ctg_cns.c:822-984), count_kmers / count_kscore (:405-449),
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
};

// the 65 536-bin table is cleared through the list of bins the previous call touched (the reference memsets 128 KB
// per call; the counts are the same)
thread_local std::vector<uint16_t> g_touched;
thread_local std::vector<uint16_t> g_kmers;   // the table itself, one per host thread (paired with g_touched)
void count_kmers(const Region& lq, std::vector<uint16_t>& kmers, int c, int l) {
    for (uint16_t k : g_touched) kmers[k] = 0;
    g_touched.clear();
    for (int j = 0; j < std::min(lq.len, c); ++j) {
        const Cand& cd = lq.seqs[(size_t)j];
        if (cd.len < (uint32_t)KMER_LEN) continue;
        const int s = l && cd.len > (uint32_t)KMER_RANGE ? (int)cd.len - KMER_RANGE : 0;
        uint16_t kmer = 0;
        for (int k = 0; k < (int)std::min<uint32_t>(cd.len, KMER_RANGE) - KMER_LEN; ++k) {
            if (k) kmer = (uint16_t)(kmer << 2 | np2k::base_to_int((unsigned char)cd.seq[(size_t)(s + k + KMER_LEN - 1)]));
            else
                for (int index = 0; index < KMER_LEN; ++index) kmer = (uint16_t)(kmer << 2 | np2k::base_to_int((unsigned char)cd.seq[(size_t)(s + k + index)]));
            if (kmers[kmer]++ == 0) g_touched.push_back(kmer);
        }
    }
}
void count_kscore(Region& lq, const std::vector<uint16_t>& kmers, int l) {
    for (int j = 0; j < lq.len; ++j) {
        Cand& cd = lq.seqs[(size_t)j];
        cd.kscore = 0;
        if (cd.len < (uint32_t)KMER_LEN) continue;
        const int s = l && cd.len > (uint32_t)KMER_RANGE ? (int)cd.len - KMER_RANGE : 0;
        uint16_t kmer = 0;
        for (int k = 0; k < (int)std::min<uint32_t>(cd.len, KMER_RANGE) - KMER_LEN; ++k) {
            if (k) kmer = (uint16_t)(kmer << 2 | np2k::base_to_int((unsigned char)cd.seq[(size_t)(s + k + KMER_LEN - 1)]));
            else
                for (int index = 0; index < KMER_LEN; ++index) kmer = (uint16_t)(kmer << 2 | np2k::base_to_int((unsigned char)cd.seq[(size_t)(s + k + index)]));
            cd.kscore = (uint16_t)(cd.kscore + kmers[kmer]);
        }
    }
}

void reverse_cands(Region& r) { std::reverse(r.seqs.begin(), r.seqs.begin() + std::max(r.len, 0)); }
void sort_by_len_asc(Region& r) {
    std::stable_sort(r.seqs.begin(), r.seqs.begin() + r.len, [](const Cand& a, const Cand& b) { return a.len < b.len; });
}
void sort_by_len_desc(Region& r) {
    std::stable_sort(r.seqs.begin(), r.seqs.begin() + r.len, [](const Cand& a, const Cand& b) { return a.len > b.len; });
}
void sort_by_kscore_desc(Region& r) {
    std::stable_sort(r.seqs.begin(), r.seqs.begin() + r.len, [](const Cand& a, const Cand& b) { return a.kscore > b.kscore; });
}

void remove_short(Region& r) {   // ctg_cns.c:620-633
    sort_by_len_desc(r);
    const int k = r.len / 4;
    while (r.len > k && (r.seqs[(size_t)r.len - 1].len < r.seqs[(size_t)k].len / 2 ||
                         r.seqs[(size_t)r.len - 1].len * 1.4 < r.seqs[(size_t)r.len - 2].len)) --r.len;
    if (k == r.len) r.len = 0;
    if (r.len > LQSEQ_MAX_COUNT) r.len = LQSEQ_MAX_COUNT;
    reverse_cands(r);
}

// shared tail of the two candidate routines: length-outlier trimming, k-mer ranking, choice of the POA inputs and the
// pseudo-seed.  min_span: indexe - indexs must exceed it (3, or 1 in the HiFi variant).  Returns false when the region
// is dropped (r.len = 0).
bool rank_and_seed(Region& r, std::vector<uint16_t>& kmers, bool trim, int min_span) {
    int k;
    if (trim) {
        sort_by_len_asc(r);
        k = r.len / 2;
        while (r.len > k && (r.seqs[(size_t)r.len - 1].len > 2 * r.seqs[(size_t)k].len ||
                             r.seqs[(size_t)r.len - 1].len >= 1.4 * r.seqs[(size_t)r.len - 2].len)) --r.len;
        if (k == r.len) { r.len = 0; return false; }
        k = r.len / 2;
        if (r.seqs[0].len < r.seqs[(size_t)k].len / 2) {
            reverse_cands(r);
            while (r.seqs[(size_t)r.len - 1].len < r.seqs[(size_t)k].len / 2) --r.len;
            if (k == r.len) { r.len = 0; return false; }
        }
    }
    count_kmers(r, kmers, LQSEQ_MAX_CAN_COUNT, 0);
    count_kscore(r, kmers, 0);
    unsigned kmaxlen = r.seqs[0].len;
    if (kmaxlen > 100) {
        uint16_t score[LQSEQ_MAX_CAN_COUNT];
        for (int j = 0; j < r.len; ++j) score[r.seqs[(size_t)j].order] = r.seqs[(size_t)j].kscore;
        count_kmers(r, kmers, LQSEQ_MAX_CAN_COUNT, 1);
        count_kscore(r, kmers, 1);
        for (int j = 0; j < r.len; ++j) r.seqs[(size_t)j].kscore = (uint16_t)(r.seqs[(size_t)j].kscore + score[r.seqs[(size_t)j].order]);
    }
    sort_by_kscore_desc(r);
    kmaxlen = r.seqs[0].len;
    unsigned klastscore, kmaxscore;
    klastscore = kmaxscore = r.seqs[0].kscore;
    int j;
    for (k = j = 0; j < r.len; ++j) {
        const Cand& cd = r.seqs[(size_t)j];
        if ((unsigned)cd.kscore * 10 < kmaxscore || j >= LQSEQ_MAX_COUNT || (unsigned)cd.kscore * 2 < klastscore) break;
        klastscore = cd.kscore;
        if (j < KMER_MAX_SEQ && cd.kscore > kmaxscore * 0.8 && cd.len > kmaxlen) { kmaxlen = cd.len; k = j; }
    }
    r.indexs = 0;
    r.indexe = (uint8_t)(kmaxlen > LQSEQ_MAX_REV_LEN && j > 6 ? 5 : j - 1);
    if (r.indexe - r.indexs <= min_span || (r.seqs[0].len > 20000 && r.len < LQSEQ_MAX_CAN_COUNT / 3)) { r.len = 0; return false; }
    j = r.indexs;
    if (r.seqs[0].len < 3000) k = j + 6 < r.indexe ? 6 : r.indexe - j + 1;
    else k = j + 2 < r.indexe ? 2 : r.indexe - j + 1;
    if (r.seqs[0].len < 20000) {
        std::vector<std::string> v;
        for (int q = 0; q < k; ++q) v.push_back(r.seqs[(size_t)(j + q)].seq);
        r.sudoseed = poa_consensus(v);
    } else {
        r.sudoseed = r.seqs[0].seq;
    }
    r.sudoseed_len = (unsigned)r.sudoseed.size();
    return true;
}

// generate_lqseqs_from_tags (kmer = false) / generate_lqseqs_from_tags_kmer (HiFi, ctg_cns.c:636-820); returns max_aln_length
int collect_candidates(Exec* exec, std::vector<Region>& lq, const WindowOutput& wo, bool kmer, const std::vector<LqCluster>& clusters, std::string* err) {
    const int count = (int)lq.size();
    if (getenv("NP2_TIMING")) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "[np2 lq]   collect start (t=%.2f)\n", ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6); }
    for (Region& r : lq) {
        r.sudoseed.clear();
        r.lqcount = 0; r.len = 0; r.sudoseed_len = 0;
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
    // ranking + pseudo-seed: the regions are independent of each other
    std::atomic<int> max_aln{0};
    np::parallel_for((size_t)count, 16, [&](size_t lo, size_t hi) {
        if (g_kmers.size() != 65536) { g_kmers.assign(65536, 0); g_touched.clear(); }
        std::vector<uint16_t>& kmers = g_kmers;
        int max_aln_length = 0;
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
                } else if (!rank_and_seed(r, kmers, r.len > 4, 1)) {
                    continue;
                }
                if ((int)(r.lqcount + r.sudoseed_len) > max_aln_length) max_aln_length = (int)(r.lqcount + r.sudoseed_len);
                continue;
            }
            if (r.l != 1 && r.l > 1 && r.len > 4) remove_short(r);
            if (r.len <= 4 || r.len < r.sudoseed_len * 0.5) { r.len = 0; continue; }
            if (!rank_and_seed(r, kmers, true, 3)) continue;
            if ((int)(r.lqcount + r.sudoseed_len) > max_aln_length) max_aln_length = (int)(r.lqcount + r.sudoseed_len);
        }
        int cur = max_aln.load();
        while (max_aln_length > cur && !max_aln.compare_exchange_weak(cur, max_aln_length)) {}
    });
    const int max_aln_length = max_aln.load();
    if (getenv("NP2_TIMING")) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "[np2 lq]   rank + poa done (t=%.2f)\n", ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6); }
    return max_aln_length;
}

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

// generate_consensus_trimed: builds the 30 concatenated alignments and runs the graph consensus on them.
// The reference appends, per round i and region j, one gapped piece (a helper writes more than it advances, the excess
// is overwritten by what follows or cut at the end: only the first `advance` characters of a piece survive).  A
// region's pieces depend on that region's own counter only, so the regions are aligned in parallel and the 30
// strings are concatenated afterwards.
bool consensus_of_regions(Exec* exec, std::vector<Region>& lq, uint32_t gap_min_len, bool hifi, std::string* cons_rev, std::string* err) {
    const int count = (int)lq.size();
    LqInput in;
    in.gap_min_len = gap_min_len;
    in.hifi = hifi;
    for (Region& r : lq) r.lqcount = 0;
    struct Pieces { std::string t[LQSEQ_MAX_COUNT], q[LQSEQ_MAX_COUNT]; };
    std::vector<Pieces> pieces((size_t)count);
    np::parallel_for((size_t)count, 8, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
            Region& r = lq[j];
            if (r.len <= 0) continue;
            const int seed_len = (int)r.sudoseed_len;
            for (int i = 0; i < LQSEQ_MAX_COUNT; ++i) {
                LinkAln a;
                const bool beyond = (i + r.indexs) > r.indexe;
                const int query_len = beyond ? seed_len : (int)r.seqs[(size_t)(i + r.indexs)].len;
                if (beyond) r.lqcount = 0;
                bool fallback = false;
                if (beyond || (i && (query_len < seed_len * 0.5 || query_len > seed_len * 1.3))) {
                    fallback = true;
                } else {
                    const Cand& cd = r.seqs[(size_t)(i + r.indexs)];
                    OndAln al;
                    ond_align(cd.seq.c_str(), query_len, r.sudoseed.c_str(), seed_len, &al);
                    if (al.aln_len > 2) {
                        a.put(al.t_aln_str, al.q_aln_str);
                        a.len += (size_t)al.aln_len;
                        int tl = al.aln_t_len, ql = al.aln_q_len;
                        while (tl < seed_len) a.push(r.sudoseed[(size_t)tl++], '-');
                        int delta = 0;
                        while (ql < (int)cd.len && delta++ < 250) a.push('-', cd.seq[(size_t)ql++]);
                    } else {
                        fallback = true;
                    }
                }
                if (fallback) {
                    if ((int)(r.lqcount++) < r.indexe - r.indexs) fill_with_seed(a, seed_len);
                    else fill_with_lqseq(a, r.sudoseed, seed_len, r.seqs[r.indexs].seq, (int)r.seqs[r.indexs].len);
                }
                pieces[j].t[i].assign(a.t, 0, a.len);
                pieces[j].q[i].assign(a.q, 0, a.len);
            }
        }
    });
    int aligned_linkseq_len = 0;
    for (int i = 0; i < LQSEQ_MAX_COUNT; ++i) {
        aligned_linkseq_len = 0;
        std::string t, q;
        for (int j = count - 1; j >= 0; --j) {
            const Region& r = lq[(size_t)j];
            if (r.len <= 0) continue;
            aligned_linkseq_len += (int)r.sudoseed_len + 1;
            t.push_back('N');
            q.push_back('N');
            t += pieces[(size_t)j].t[i];
            q += pieces[(size_t)j].q[i];
        }
        ++aligned_linkseq_len;
        t.push_back('N');
        q.push_back('N');
        in.t.push_back(std::move(t));
        in.q.push_back(std::move(q));
    }
    in.t_len = (uint32_t)aligned_linkseq_len;
    if (getenv("NP2_TIMING")) {
        timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        static double last = 0; const double t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
        fprintf(stderr, "[np2 lq]   alignments built, %u target columns (t=%.2f)\n", in.t_len, t - last); last = t;
    }
    return exec->run_lq(in, cons_rev, err);
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
