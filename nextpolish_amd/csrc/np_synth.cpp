// Synthetic draft + short-read workload generator.  See np_synth.h.
#include "np_synth.h"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <numeric>

#include "np_bam.h"
#include "np_threads.h"

namespace np {

namespace {

struct Rng {
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        uint64_t z = seed;
        for (int i = 0; i < 4; ++i) {   // splitmix64 seeding
            z += 0x9e3779b97f4a7c15ULL;
            uint64_t x = z;
            x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
            x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
            s[i] = x ^ (x >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {   // xoshiro256**
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    bool chance(double p) { return p > 0 && uni() < p; }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

const char kBases[4] = {'A', 'C', 'G', 'T'};
inline uint8_t nt16(char c) {
    switch (c) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 4;
        case 'T': case 't': return 8;
        default: return 15;
    }
}

struct TmpRead {
    int32_t pos;
    uint16_t flag;
    uint8_t mapq;
    int32_t isize;
    uint64_t cig_beg;           // into the CIGAR pool (a 250 Mb contig at 30x holds 4.3 G bases: 32 bits are not enough)
    uint32_t cig_n;
    uint64_t seq_beg;           // into the unpacked base / qual pools
    uint32_t l_qseq;
};

struct Col { char op; int32_t dcoord; char base; };   // one alignment column of a read against the draft

// One read pair of a fragment truth[f, f + frag): alignment columns of both mates against the draft (edit script of the draft + read
// errors), CIGAR, flags; appended to the pools.  Shared by the serial generator (one RNG over the whole stream: its call order is what
// the committed golden vectors depend on) and the segmented one (one RNG per 1 Mb segment).
struct PairGen {
    const np_synth_params& p;
    const std::string& T;
    const std::vector<int32_t>& dpos;
    const std::vector<uint8_t>& dins;
    int32_t Lt, RL;
    std::vector<TmpRead>& reads;
    std::vector<uint32_t>& cig_pool;
    std::vector<char>& base_pool;
    std::vector<uint8_t>& qual_pool;
    std::vector<Col>& cols;
    void pair(Rng& rng, int32_t frag, int32_t f) {
        bool swap = rng.chance(0.5);   // which mate is on the forward strand
        int32_t starts[2] = {f, f + frag - RL};
        int32_t mate_pos[2] = {-1, -1};
        size_t first_idx = reads.size();
        for (int m = 0; m < 2; ++m) {
            // alignment columns of truth[a..) against the draft, plus read errors
            cols.clear();
            int32_t t = starts[m], q = 0;
            int32_t dins_done = -1;   // truth position whose preceding draft-only bases were already emitted
            while (q < RL && t < Lt) {
                if (!cols.empty() && dins[t] && dins_done != t)
                    for (int k = 0; k < dins[t]; ++k) cols.push_back(Col{'D', -1, 0});   // bases only the draft has
                dins_done = t;
                if (!cols.empty() && rng.chance(p.read_indel * 0.5)) {   // read deletion
                    if (dpos[t] >= 0) cols.push_back(Col{'D', dpos[t], 0});
                    ++t;
                    continue;
                }
                if (!cols.empty() && rng.chance(p.read_indel * 0.5)) {   // read insertion
                    cols.push_back(Col{'I', -1, kBases[rng.below(4)]});
                    ++q;
                    continue;
                }
                char b = T[t];
                if (rng.chance(p.read_sub)) b = kBases[(std::find(kBases, kBases + 4, b) - kBases + 1 + rng.below(3)) & 3];
                if (dpos[t] >= 0) cols.push_back(Col{'M', dpos[t], b});
                else cols.push_back(Col{'I', -1, b});
                ++q; ++t;
            }
            // optional soft clips
            int clipL = 0, clipR = 0;
            if (rng.chance(p.softclip_rate)) {
                int k = 3 + (int)rng.below(40);
                if (rng.chance(0.5)) clipL = k; else clipR = k;
            }
            // left end: turn the first clipL query bases into S, then make the alignment start with M
            size_t lo = 0, hi = cols.size();
            int sL = 0, sR = 0;
            while (lo < hi && (sL < clipL || cols[lo].op != 'M')) {
                if (cols[lo].op != 'D') ++sL;
                ++lo;
            }
            while (hi > lo && (sR < clipR || cols[hi - 1].op != 'M')) {
                if (cols[hi - 1].op != 'D') ++sR;
                --hi;
            }
            if (lo >= hi) continue;   // nothing aligned (can only happen for absurd parameters)
            TmpRead r;
            r.pos = cols[lo].dcoord;
            r.cig_beg = (uint64_t)cig_pool.size();
            r.seq_beg = (uint64_t)base_pool.size();
            // sequence = every non-D column in order (clipped bases stay in SEQ)
            for (const Col& cl : cols)
                if (cl.op != 'D') base_pool.push_back(cl.base);
            r.l_qseq = (uint32_t)(base_pool.size() - r.seq_beg);
            if (p.with_qual)
                for (uint32_t k = 0; k < r.l_qseq; ++k) qual_pool.push_back((uint8_t)(25 + rng.below(16)));
            auto push_op = [&](uint32_t op, uint32_t len) {
                if (!len) return;
                if (cig_pool.size() > r.cig_beg && (cig_pool.back() & 0xf) == op) cig_pool.back() += len << 4;
                else cig_pool.push_back(len << 4 | op);
            };
            bool weird = rng.chance(p.weird_rate);
            int wkind = weird ? (int)rng.below(5) : -1;
            if (wkind == 0 && sL == 0) {
                // hard clip on a primary record: SEQ does not hold the clipped bases, the walk still
                // advances the query cursor (reference: source/lib/contig.c:321-324)
                push_op(5, 5 + rng.below(20));
            }
            push_op(4, (uint32_t)sL);
            bool used_eqx = false;
            for (size_t k = lo; k < hi; ++k) {
                uint32_t op = cols[k].op == 'M' ? 0u : (cols[k].op == 'I' ? 1u : 2u);
                if (wkind == 1 && op == 0 && !used_eqx && k > lo + 20 && k + 20 < hi) {
                    // a short '='/'X' stretch (ops the reference walk ignores entirely)
                    uint32_t n = 0;
                    while (n < 6 && k + n < hi && cols[k + n].op == 'M') ++n;
                    // keep an M on both sides: two insertions with nothing but ignored ops between them would sit
                    // at one reference position, a CIGAR shape no aligner emits and the GPU path rejects
                    if (cols[k - 1].op != 'M' || k + n >= hi || cols[k + n].op != 'M') { push_op(op, 1); continue; }
                    push_op(rng.chance(0.5) ? 7u : 8u, n);
                    k += n - 1;
                    used_eqx = true;
                    continue;
                }
                if (wkind == 2 && op == 2 && !used_eqx && k > lo && k + 1 < hi && cols[k - 1].op == 'M' && cols[k + 1].op == 'M') {
                    push_op(3, 1); used_eqx = true; continue;   // N (ignored by the reference walk: no pos advance)
                }
                push_op(op, 1);
            }
            push_op(4, (uint32_t)sR);
            if (wkind == 3) push_op(5, 3 + rng.below(9));   // trailing hard clip
            r.cig_n = (uint32_t)(cig_pool.size() - r.cig_beg);
            if (wkind == 4 && r.pos == 0 && r.cig_n >= 1 && sL >= 2) {
                // leading insertion at contig position 0 (reference: source/lib/contig.c:315-319)
                cig_pool[r.cig_beg] = (uint32_t)sL << 4 | 1u;
            }
            bool fwd = (m == 0) != swap;
            uint16_t flag = (uint16_t)(0x1 | 0x2 | (fwd ? 0x20 : 0x10) | (m == 0 ? 0x40 : 0x80));
            if (rng.chance(p.dup_rate)) flag |= 0x400;
            if (rng.chance(p.supp_rate)) flag |= 0x800;
            if (rng.chance(p.sec_rate)) flag |= 0x100;
            r.flag = flag;
            r.mapq = rng.chance(p.lowmapq_rate) ? (uint8_t)rng.below(31) : 60;
            r.isize = (m == 0) ? frag : -frag;
            if (rng.chance(0.002)) r.isize = (m == 0 ? 1 : -1) * (int32_t)(20000 + rng.below(100000));   // chimeric pair
            mate_pos[m] = r.pos;
            reads.push_back(r);
            if (rng.chance(p.unmapped_rate)) {   // an unmapped mate placed at this position, no CIGAR
                TmpRead u = r;
                u.flag = (uint16_t)(0x1 | 0x4 | (m == 0 ? 0x80 : 0x40));
                u.cig_beg = (uint64_t)cig_pool.size();
                u.cig_n = 0;
                u.mapq = 0;
                u.isize = 0;
                reads.push_back(u);
            }
        }
        (void)first_idx; (void)mate_pos;
    }
};

}  // namespace

void synth_default_params(np_synth_params* p) {
    memset(p, 0, sizeof(*p));
    p->seed = 20250117;
    p->depth = 30;
    p->read_len = 150;
    p->frag_mean = 300;
    p->frag_sd = 30;
    p->draft_sub = 0.001;
    p->draft_indel = 0.005;
    p->draft_lower = 0.0;
    p->read_sub = 0.001;
    p->read_indel = 0.0001;
    p->softclip_rate = 0.005;
    p->dup_rate = 0.002;
    p->supp_rate = 0.001;
    p->sec_rate = 0.001;
    p->unmapped_rate = 0.001;
    p->lowmapq_rate = 0.01;
    p->weird_rate = 0.0;
    p->with_qual = 0;
}

namespace {

// Contigs of chromosome size (>= kSegmentedMin truth bases): the same workload model generated segment by segment (1 Mb of truth
// each, its own RNG stream, all host threads) instead of by one RNG over the whole contig -- a 250 Mb contig at 30x is 50 M
// records and took 100 s that way.  Draft edits do not cross a segment boundary; a fragment belongs to the segment its first base
// lies in; reads are sorted inside buckets of draft coordinates (a segment's own reads + the few of its left neighbour that start
// behind the boundary).  Contigs below the threshold keep the serial generator bit for bit (the committed golden vectors use those).
constexpr int32_t kSegmentedMin = 16 << 20, kSeg = 1 << 20;

void synth_contig_segmented(const np_synth_params& p, int c, const std::string& name, ReadStream* out) {
    const int32_t Lt = p.contig_len[c];
    const int nseg = (int)((Lt + kSeg - 1) / kSeg);
    const int RL = p.read_len;
    static const bool timing = getenv("NP_SYNTH_TIMING") != nullptr;
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
    double tp[6] = {now(), 0, 0, 0, 0, 0};
    auto seed_of = [&](int seg, uint64_t what) { return (p.seed + 0x51ed270b1ULL * (uint64_t)(c + 1)) * 0x9e3779b97f4a7c15ULL + ((uint64_t)seg << 20) + what; };
    std::string T((size_t)Lt, 'A');
    std::vector<int32_t> dpos((size_t)Lt + 1, -1);
    std::vector<uint8_t> dins((size_t)Lt + 1, 0);
    std::vector<std::string> Dseg((size_t)nseg);
    // ---- truth + draft pieces (dpos relative to the piece for now)
    parallel_for((size_t)nseg, 1, [&](size_t lo, size_t hi) {
        for (size_t sg = lo; sg < hi; ++sg) {
            Rng rng(seed_of((int)sg, 1));
            const int32_t t0 = (int32_t)sg * kSeg, t1 = std::min<int64_t>(Lt, (int64_t)t0 + kSeg);
            for (int32_t i = t0; i < t1; ++i) T[(size_t)i] = kBases[rng.below(4)];
            std::string& D = Dseg[sg];
            D.reserve((size_t)(t1 - t0) + (size_t)(t1 - t0) / 100);
            for (int32_t t = t0; t < t1;) {
                if (t > t0 && rng.chance(p.draft_indel * 0.5)) {
                    const int n = 1 + (int)rng.below(3);
                    const bool hp = rng.chance(0.5);
                    for (int k = 0; k < n; ++k) D.push_back(hp ? T[(size_t)t - 1] : kBases[rng.below(4)]);
                    dins[(size_t)t] = (uint8_t)n;
                }
                if (t > t0 && rng.chance(p.draft_indel * 0.5)) {
                    const int n = 1 + (int)rng.below(3);
                    for (int k = 0; k < n && t < t1 - 1; ++k) dpos[(size_t)t++] = -1;
                    continue;
                }
                char b = T[(size_t)t];
                if (rng.chance(p.draft_sub)) b = kBases[(std::find(kBases, kBases + 4, b) - kBases + 1 + rng.below(3)) & 3];
                dpos[(size_t)t] = (int32_t)D.size();
                D.push_back(b);
                ++t;
            }
            if (p.draft_lower > 0)
                for (size_t i = 0; i < D.size(); ++i)
                    if (rng.chance(p.draft_lower)) {
                        const int n = 1 + (int)rng.below(4);
                        for (int k = 0; k < n && i + (size_t)k < D.size(); ++k) D[i + (size_t)k] = (char)(D[i + (size_t)k] + 32);
                        i += (size_t)n;
                    }
        }
    });
    tp[1] = now();
    std::vector<int64_t> dbase((size_t)nseg + 1, 0);
    for (int sg = 0; sg < nseg; ++sg) dbase[(size_t)sg + 1] = dbase[(size_t)sg] + (int64_t)Dseg[(size_t)sg].size();
    const int32_t Ld = (int32_t)dbase[(size_t)nseg];
    const size_t draft0 = out->draft.size();
    out->draft.resize(draft0 + (size_t)Ld);
    parallel_for((size_t)nseg, 1, [&](size_t lo, size_t hi) {
        for (size_t sg = lo; sg < hi; ++sg) {
            memcpy(&out->draft[draft0 + (size_t)dbase[sg]], Dseg[sg].data(), Dseg[sg].size());
            const int32_t t0 = (int32_t)sg * kSeg, t1 = std::min<int64_t>(Lt, (int64_t)t0 + kSeg);
            for (int32_t t = t0; t < t1; ++t)
                if (dpos[(size_t)t] >= 0) dpos[(size_t)t] += (int32_t)dbase[sg];
            std::string().swap(Dseg[sg]);
        }
    });
    tp[2] = now();
    // ---- reads, segment by segment
    struct Seg {
        std::vector<TmpRead> reads;
        std::vector<uint32_t> cig_pool;
        std::vector<char> base_pool;
        std::vector<uint8_t> qual_pool;
        std::vector<uint32_t> own, spill;      // read indices sorted by position: below / at or behind the next segment's first draft base
        std::vector<uint64_t> order;           // final order of bucket sg: source (1 = left neighbour's spill) << 32 | read index
        uint64_t n_cig = 0, n_seq = 0, n_qual = 0;
    };
    std::vector<Seg> segs((size_t)nseg);
    const uint64_t n_pairs = Lt < RL + 2 ? 0 : (uint64_t)((double)Lt * p.depth / (2.0 * RL) + 0.5);
    parallel_for((size_t)nseg, 1, [&](size_t lo, size_t hi) {
        std::vector<Col> cols;
        for (size_t sg = lo; sg < hi; ++sg) {
            Seg& S = segs[sg];
            Rng rng(seed_of((int)sg, 2));
            const int32_t t0 = (int32_t)sg * kSeg, t1 = std::min<int64_t>(Lt, (int64_t)t0 + kSeg);
            const uint64_t mine = n_pairs * (uint64_t)t1 / (uint64_t)Lt - n_pairs * (uint64_t)t0 / (uint64_t)Lt;
            PairGen gen{p, T, dpos, dins, Lt, RL, S.reads, S.cig_pool, S.base_pool, S.qual_pool, cols};
            S.reads.reserve((size_t)mine * 2 + 16);
            S.base_pool.reserve((size_t)mine * 2 * (size_t)RL + 1024);
            for (uint64_t pi = 0; pi < mine; ++pi) {
                int32_t frag = (int32_t)std::lround(p.frag_mean + p.frag_sd * rng.normal());
                if (frag < RL) frag = RL;
                if (frag > Lt) frag = Lt;
                const int64_t f_hi = std::min<int64_t>(t1, (int64_t)Lt - frag + 1);
                if (f_hi <= t0) continue;
                gen.pair(rng, frag, t0 + (int32_t)rng.below((uint32_t)(f_hi - t0)));
            }
            // first draft coordinate of the next segment: reads that start there or behind it are sorted with that segment's
            int32_t next_d = Ld;
            for (int64_t t = t1; t < Lt; ++t)
                if (dpos[(size_t)t] >= 0) { next_d = dpos[(size_t)t]; break; }
            std::vector<uint32_t> idx(S.reads.size());
            std::iota(idx.begin(), idx.end(), 0u);
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return S.reads[a].pos < S.reads[b].pos; });
            for (uint32_t i : idx) (S.reads[i].pos >= next_d && sg + 1 < (size_t)nseg ? S.spill : S.own).push_back(i);
        }
    });
    tp[3] = now();
    // ---- buckets: own reads + the left neighbour's spill, sorted by position (spill first among equals); sizes
    parallel_for((size_t)nseg, 1, [&](size_t lo, size_t hi) {
        for (size_t sg = lo; sg < hi; ++sg) {
            Seg& S = segs[sg];
            const Seg* Lf = sg ? &segs[sg - 1] : nullptr;
            S.order.reserve(S.own.size() + (Lf ? Lf->spill.size() : 0));
            size_t a = 0, b = 0;
            const size_t na = Lf ? Lf->spill.size() : 0, nb = S.own.size();
            while (a < na || b < nb) {
                const bool take_spill = a < na && (b >= nb || Lf->reads[Lf->spill[a]].pos <= S.reads[S.own[b]].pos);
                if (take_spill) S.order.push_back(1ull << 32 | Lf->spill[a++]);
                else S.order.push_back((uint64_t)S.own[b++]);
            }
            for (uint64_t o : S.order) {
                const TmpRead& r = (o >> 32) ? Lf->reads[(uint32_t)o] : S.reads[(uint32_t)o];
                S.n_cig += r.cig_n;
                S.n_seq += (r.l_qseq + 1) / 2;
                S.n_qual += p.with_qual ? r.l_qseq : 0;
            }
        }
    });
    std::vector<uint64_t> r_at((size_t)nseg + 1, 0), c_at((size_t)nseg + 1, 0), s_at((size_t)nseg + 1, 0), q_at((size_t)nseg + 1, 0);
    for (int sg = 0; sg < nseg; ++sg) {
        r_at[(size_t)sg + 1] = r_at[(size_t)sg] + segs[(size_t)sg].order.size();
        c_at[(size_t)sg + 1] = c_at[(size_t)sg] + segs[(size_t)sg].n_cig;
        s_at[(size_t)sg + 1] = s_at[(size_t)sg] + segs[(size_t)sg].n_seq;
        q_at[(size_t)sg + 1] = q_at[(size_t)sg] + segs[(size_t)sg].n_qual;
    }
    tp[4] = now();
    const size_t r0 = out->n_reads(), c0 = out->cigar.size(), s0 = out->seq.size(), q0 = out->qual.size();
    const size_t nr = (size_t)r_at[(size_t)nseg];
    out->names.push_back(name);
    out->ctg_len.push_back(Ld);
    out->ctg_off.push_back((uint32_t)out->draft.size());
    out->read_begin.push_back(r0);
    out->pos.resize(r0 + nr); out->ctg.resize(r0 + nr); out->flag.resize(r0 + nr); out->n_cigar.resize(r0 + nr); out->l_qseq.resize(r0 + nr);
    out->mapq.resize(r0 + nr); out->isize.resize(r0 + nr); out->cigar_off.resize(r0 + nr); out->seq_off.resize(r0 + nr); out->qual_off.resize(r0 + nr);
    out->cigar.resize(c0 + (size_t)c_at[(size_t)nseg]);
    out->seq.resize(s0 + (size_t)s_at[(size_t)nseg]);
    out->qual.resize(q0 + (size_t)q_at[(size_t)nseg]);
    parallel_for((size_t)nseg, 1, [&](size_t lo, size_t hi) {
        for (size_t sg = lo; sg < hi; ++sg) {
            const Seg& S = segs[sg];
            const Seg* Lf = sg ? &segs[sg - 1] : nullptr;
            size_t ri = r0 + (size_t)r_at[sg], ci = c0 + (size_t)c_at[sg], si = s0 + (size_t)s_at[sg], qi = q0 + (size_t)q_at[sg];
            for (uint64_t o : S.order) {
                const Seg& src = (o >> 32) ? *Lf : S;
                const TmpRead& r = src.reads[(uint32_t)o];
                out->pos[ri] = r.pos;
                out->ctg[ri] = (uint32_t)c;
                out->flag[ri] = r.flag;
                out->n_cigar[ri] = (uint32_t)r.cig_n;
                out->l_qseq[ri] = (int32_t)r.l_qseq;
                out->mapq[ri] = r.mapq;
                out->isize[ri] = r.isize;
                out->cigar_off[ri] = ci;
                out->seq_off[ri] = si;
                out->qual_off[ri] = qi;
                for (uint32_t k = 0; k < r.cig_n; ++k) out->cigar[ci++] = src.cig_pool[r.cig_beg + k];
                for (uint32_t k = 0; k < r.l_qseq; k += 2) {
                    const uint8_t hi4 = nt16(src.base_pool[r.seq_beg + k]);
                    const uint8_t lo4 = (k + 1 < r.l_qseq) ? nt16(src.base_pool[r.seq_beg + k + 1]) : 0;
                    out->seq[si++] = (uint8_t)(hi4 << 4 | lo4);
                }
                if (p.with_qual)
                    for (uint32_t k = 0; k < r.l_qseq; ++k) out->qual[qi++] = src.qual_pool[r.seq_beg + k];
                ++ri;
            }
        }
    });
    tp[5] = now();
    if (timing)
        fprintf(stderr, "[np synth] contig %d, %d segments, %u threads: truth+draft %.2f s, draft concat %.2f, reads %.2f, buckets %.2f, output %.2f\n", c, nseg, host_threads(),
                tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4]);
}

}  // namespace

namespace {
// the reads of one contig (truth T, its map onto the draft D) appended to the stream, coordinate-sorted; shared by the generator that
// invents truth and draft and the one that is handed the draft (synth_stream_on)
struct ContigScratch {
    std::vector<TmpRead> reads;
    std::vector<uint32_t> cig_pool;
    std::vector<char> base_pool;
    std::vector<uint8_t> qual_pool;
    std::vector<Col> cols;
};
void synth_contig_reads(const np_synth_params& p, Rng& rng, int c, const std::string& name, const std::string& T, const std::vector<int32_t>& dpos,
                        const std::vector<uint8_t>& dins, const std::string& D, ContigScratch& W, ReadStream* out);
}  // namespace

bool synth_stream(const np_synth_params& p, const std::string& prefix, ReadStream* out) {
    out->clear();
    out->ctg_off.push_back(0);
    Rng rng(p.seed);
    ContigScratch W;
    for (int c = 0; c < p.n_contigs; ++c) {
        const int32_t Lt = p.contig_len[c];
        if (Lt >= kSegmentedMin) {     // chromosome-sized: generated in parallel, segment by segment (own RNG streams; `rng` is not touched)
            char nm[64];
            snprintf(nm, sizeof(nm), "%s%04d", prefix.c_str(), c + 1);
            synth_contig_segmented(p, c, nm, out);
            continue;
        }
        // ---- truth
        std::string T(Lt, 'A');
        for (int32_t i = 0; i < Lt; ++i) T[i] = kBases[rng.below(4)];
        // ---- draft = truth + edits.  dpos[t]: draft coordinate of truth base t (or -1 when the draft
        // lacks it); dins[t]: number of extra draft bases inserted right before truth base t.
        std::vector<int32_t> dpos(Lt + 1, -1);
        std::vector<uint8_t> dins(Lt + 1, 0);
        std::string D;
        D.reserve((size_t)Lt + Lt / 100);
        for (int32_t t = 0; t < Lt;) {
            if (t > 0 && rng.chance(p.draft_indel * 0.5)) {   // draft insertion, 1-3 bp, sometimes homopolymer
                int n = 1 + (int)rng.below(3);
                bool hp = rng.chance(0.5);
                for (int k = 0; k < n; ++k) D.push_back(hp ? T[t - 1] : kBases[rng.below(4)]);
                dins[t] = (uint8_t)n;
            }
            if (t > 0 && rng.chance(p.draft_indel * 0.5)) {   // draft deletion, 1-3 bp
                int n = 1 + (int)rng.below(3);
                for (int k = 0; k < n && t < Lt - 1; ++k) dpos[t++] = -1;
                continue;
            }
            char b = T[t];
            if (rng.chance(p.draft_sub)) b = kBases[(std::find(kBases, kBases + 4, b) - kBases + 1 + rng.below(3)) & 3];
            dpos[t] = (int32_t)D.size();
            D.push_back(b);
            ++t;
        }
        const int32_t Ld = (int32_t)D.size();
        if (p.draft_lower > 0)
            for (int32_t i = 0; i < Ld; ++i)
                if (rng.chance(p.draft_lower)) {   // short lowercase runs, like a previous round's output
                    int n = 1 + (int)rng.below(4);
                    for (int k = 0; k < n && i + k < Ld; ++k) D[i + k] = (char)(D[i + k] + 32);
                    i += n;
                }
        char nm[64];
        snprintf(nm, sizeof(nm), "%s%04d", prefix.c_str(), c + 1);
        synth_contig_reads(p, rng, c, nm, T, dpos, dins, D, W, out);
    }
    out->read_begin.push_back(out->n_reads());
    return true;
}

namespace {
void synth_contig_reads(const np_synth_params& p, Rng& rng, int c, const std::string& name, const std::string& T, const std::vector<int32_t>& dpos,
                        const std::vector<uint8_t>& dins, const std::string& D, ContigScratch& W, ReadStream* out) {
    std::vector<TmpRead>& reads = W.reads;
    std::vector<uint32_t>& cig_pool = W.cig_pool;
    std::vector<char>& base_pool = W.base_pool;
    std::vector<uint8_t>& qual_pool = W.qual_pool;
    std::vector<Col>& cols = W.cols;
    reads.clear(); cig_pool.clear(); base_pool.clear(); qual_pool.clear();
    const int32_t Lt = (int32_t)T.size(), Ld = (int32_t)D.size();
    {
        // ---- reads
        const int RL = p.read_len;
        PairGen gen{p, T, dpos, dins, Lt, RL, reads, cig_pool, base_pool, qual_pool, cols};
        uint64_t n_pairs = (uint64_t)((double)Lt * p.depth / (2.0 * RL) + 0.5);
        if (Lt < RL + 2) n_pairs = 0;
        for (uint64_t pi = 0; pi < n_pairs; ++pi) {
            int32_t frag = (int32_t)std::lround(p.frag_mean + p.frag_sd * rng.normal());
            if (frag < RL) frag = RL;
            if (frag > Lt) frag = Lt;
            int32_t f = (int32_t)rng.below((uint32_t)(Lt - frag + 1));
            gen.pair(rng, frag, f);
        }
        // ---- coordinate sort (stable) and append to the stream
        std::vector<uint32_t> order(reads.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return reads[a].pos < reads[b].pos; });
        out->names.push_back(name);
        out->ctg_len.push_back(Ld);
        out->draft += D;
        out->ctg_off.push_back((uint32_t)out->draft.size());
        out->read_begin.push_back(out->n_reads());
        for (uint32_t oi : order) {
            const TmpRead& r = reads[oi];
            out->pos.push_back(r.pos);
            out->ctg.push_back((uint32_t)c);
            out->flag.push_back(r.flag);
            out->n_cigar.push_back((uint32_t)r.cig_n);
            out->l_qseq.push_back((int32_t)r.l_qseq);
            out->mapq.push_back(r.mapq);
            out->isize.push_back(r.isize);
            out->cigar_off.push_back(out->cigar.size());
            out->seq_off.push_back(out->seq.size());
            out->cigar.insert(out->cigar.end(), cig_pool.begin() + r.cig_beg, cig_pool.begin() + r.cig_beg + r.cig_n);
            for (uint32_t k = 0; k < r.l_qseq; k += 2) {
                uint8_t hi4 = nt16(base_pool[r.seq_beg + k]);
                uint8_t lo4 = (k + 1 < r.l_qseq) ? nt16(base_pool[r.seq_beg + k + 1]) : 0;
                out->seq.push_back((uint8_t)(hi4 << 4 | lo4));
            }
            out->qual_off.push_back(out->qual.size());
            if (p.with_qual)
                out->qual.insert(out->qual.end(), qual_pool.begin() + r.seq_beg, qual_pool.begin() + r.seq_beg + r.l_qseq);
        }
        }
}
}  // namespace

// The short-read workload over contigs that are HANDED IN (the assembly a polishing step left): a truth is derived from each draft by the
// same edit process run the other way (substitutions; runs of 1-3 bases the truth has and the draft lacks; runs the draft has and the
// truth lacks), then reads are sampled from the truth and aligned through the known edit script -- the "re-mapped reads" of the next
// step of a multi-step run, without a mapper (tests/tools/check_config5_chain.py).  p.n_contigs / p.contig_len are not used.
bool synth_stream_on(const np_synth_params& p, const std::vector<std::string>& names, const std::vector<std::string>& drafts, ReadStream* out) {
    out->clear();
    out->ctg_off.push_back(0);
    Rng rng(p.seed);
    ContigScratch W;
    for (size_t c = 0; c < drafts.size(); ++c) {
        const std::string& D = drafts[c];
        const int32_t Ld = (int32_t)D.size();
        std::string T;
        std::vector<int32_t> dpos;
        std::vector<uint8_t> dins;
        T.reserve((size_t)Ld + (size_t)Ld / 100);
        uint32_t pending = 0;      // draft-only bases since the last truth base that has a place in the draft
        for (int32_t d = 0; d < Ld;) {
            if (d > 0 && rng.chance(p.draft_indel * 0.5)) {      // bases the draft lacks
                const int n = 1 + (int)rng.below(3);
                const bool hp = rng.chance(0.5);
                for (int k = 0; k < n; ++k) { T.push_back(hp ? T.back() : kBases[rng.below(4)]); dpos.push_back(-1); dins.push_back(0); }
            }
            if (d > 0 && rng.chance(p.draft_indel * 0.5)) {      // bases only the draft has (never its last one)
                const int n = 1 + (int)rng.below(3);
                int k = 0;
                for (; k < n && d < Ld - 1 && pending < 200; ++k) { ++d; ++pending; }
                if (k) continue;
            }
            char b = (char)(D[(size_t)d] & ~32);
            if (b != 'A' && b != 'C' && b != 'G' && b != 'T') b = 'A';
            if (rng.chance(p.draft_sub)) b = kBases[(std::find(kBases, kBases + 4, b) - kBases + 1 + rng.below(3)) & 3];
            // (a truth base without a place in the draft cannot carry the draft-only bases: they go to the next one that has a place)
            T.push_back(b);
            dpos.push_back(d);
            dins.push_back((uint8_t)pending);
            pending = 0;
            ++d;
        }
        dpos.push_back(-1);
        dins.push_back(0);
        synth_contig_reads(p, rng, (int)c, names[c], T, dpos, dins, D, W, out);
    }
    out->read_begin.push_back(out->n_reads());
    return true;
}

// Long-read workload of the nextpolish2 path (BASELINE configs[3]): a random draft per contig and noisy reads with
// log-normal lengths whose CIGAR against the draft follows from the error process (substitution / insertion /
// deletion per draft base, indel lengths uniform in 1..max_indel, first and last column always a match, optional
// soft clips).  Reads come out sorted by position.
namespace {
bool synth_long_impl(const np_synth_long_params& p_in, const std::string& prefix, const std::vector<std::string>* given_names, const std::vector<std::string>* given, ReadStream* out);
}
bool synth_long_stream(const np_synth_long_params& p, const std::string& prefix, ReadStream* out) { return synth_long_impl(p, prefix, nullptr, nullptr, out); }
// the same reads over contigs that are handed in (the assembly a polishing step left: "re-mapped" reads of the next step)
bool synth_long_stream_on(const np_synth_long_params& p, const std::vector<std::string>& names, const std::vector<std::string>& drafts, ReadStream* out) {
    return synth_long_impl(p, "", &names, &drafts, out);
}
namespace {
bool synth_long_impl(const np_synth_long_params& p_in, const std::string& prefix, const std::vector<std::string>* given_names, const std::vector<std::string>* given, ReadStream* out) {
    np_synth_long_params p = p_in;
    if (given) p.n_contigs = (int32_t)given->size();
    out->clear();
    out->ctg_off.push_back(0);
    Rng rng(p.seed);
    static const char B[4] = {'A', 'C', 'G', 'T'};
    std::vector<char> seq;
    std::vector<uint32_t> cig;
    for (int c = 0; c < p.n_contigs; ++c) {
        const int32_t L = given ? (int32_t)(*given)[(size_t)c].size() : p.contig_len[c];
        std::string D((size_t)L, 'A');
        if (given) {      // the reads are copies of the draft with errors: the draft's own letters, upper case (a polished FASTA marks changes in lower case)
            for (int32_t i = 0; i < L; ++i) {
                const char u = (char)((*given)[(size_t)c][(size_t)i] & ~32);
                D[(size_t)i] = u == 'A' || u == 'C' || u == 'G' || u == 'T' ? u : 'A';
            }
        } else {
            for (int32_t i = 0; i < L; ++i) D[(size_t)i] = B[rng.below(4)];
        }
        char nm[64];
        snprintf(nm, sizeof(nm), "%s%d", prefix.c_str(), c);
        out->names.push_back(given_names ? (*given_names)[(size_t)c] : std::string(nm));
        out->ctg_len.push_back(L);
        out->draft += given ? (*given)[(size_t)c] : D;
        out->ctg_off.push_back((uint32_t)out->draft.size());
        out->read_begin.push_back(out->n_reads());
        const uint32_t n_reads = std::max<uint32_t>(1, (uint32_t)(p.depth * L / p.mean_len));
        std::vector<int32_t> starts(n_reads);
        for (uint32_t k = 0; k < n_reads; ++k) starts[k] = (int32_t)rng.below((uint32_t)std::max(1, L - 600));
        std::sort(starts.begin(), starts.end());
        for (int32_t st : starts) {
            const int32_t rl = std::max(700, (int32_t)(std::exp(0.5 * rng.normal()) * p.mean_len));
            const int32_t en = std::min(L, st + rl);
            seq.clear();
            cig.clear();
            auto push = [&](uint32_t op, uint32_t n) {
                if (!cig.empty() && (cig.back() & 15u) == op) cig.back() += n << 4;
                else cig.push_back(n << 4 | op);
            };
            if (rng.chance(p.clip_rate)) {
                const uint32_t n = 1 + rng.below(300);
                push(4, n);
                for (uint32_t k = 0; k < n; ++k) seq.push_back(B[rng.below(4)]);
            }
            int32_t pos = st;
            while (pos < en) {
                const bool edge = pos == st || pos >= en - 1;
                const double x = rng.uni();
                if (!edge && x < p.dele) {
                    const int32_t n = std::min<int32_t>(1 + (int32_t)rng.below((uint32_t)p.max_indel), en - 1 - pos);
                    if (n > 0) { push(2, (uint32_t)n); pos += n; continue; }
                }
                if (!edge && x < p.dele + p.ins) {
                    const uint32_t n = 1 + rng.below((uint32_t)p.max_indel);
                    push(1, n);
                    for (uint32_t k = 0; k < n; ++k) seq.push_back(B[rng.below(4)]);
                }
                char b = D[(size_t)pos];
                if (rng.chance(p.sub)) { char o; do { o = B[rng.below(4)]; } while (o == b); b = o; }
                push(0, 1);
                seq.push_back(b);
                ++pos;
            }
            if (rng.chance(p.clip_rate)) {
                const uint32_t n = 1 + rng.below(300);
                push(4, n);
                for (uint32_t k = 0; k < n; ++k) seq.push_back(B[rng.below(4)]);
            }
            if (cig.size() > 65535) continue;   // the stream keeps n_cigar in 16 bits
            out->pos.push_back(st);
            out->ctg.push_back((uint32_t)c);
            out->flag.push_back((uint16_t)(rng.chance(0.5) ? 16 : 0));
            out->n_cigar.push_back((uint32_t)cig.size());
            out->l_qseq.push_back((int32_t)seq.size());
            out->mapq.push_back(60);
            out->isize.push_back(0);
            out->cigar_off.push_back(out->cigar.size());
            out->seq_off.push_back(out->seq.size());
            out->qual_off.push_back(0);
            out->cigar.insert(out->cigar.end(), cig.begin(), cig.end());
            for (size_t k = 0; k < seq.size(); k += 2) {
                const uint8_t hi4 = nt16(seq[k]), lo4 = k + 1 < seq.size() ? nt16(seq[k + 1]) : 0;
                out->seq.push_back((uint8_t)(hi4 << 4 | lo4));
            }
        }
    }
    out->read_begin.push_back(out->n_reads());
    return true;
}
}  // namespace

// Base qualities as a current Illumina instrument reports them (four bins: 2, 12, 23, 37), for BAM files written from streams that
// carry none: mostly 37, short runs of lower bins that get more frequent towards the 3' end of the read, a few reads bad from some
// cycle on.  A pure function of the record's serial number, so files are reproducible and nothing has to be kept in memory.
// (The qualities do not enter score_chain; they decide how well the BAM compresses, i.e. what the BGZF inflate has to do.)
void synth_binned_qualities(uint64_t serial, bool reverse, int32_t l, uint8_t* out) {
    if (l <= 0) return;
    memset(out, 37, (size_t)l);
    Rng rng(0x5eedf00dULL ^ (serial * 0x9e3779b97f4a7c15ULL));
    static const uint8_t kBin[3] = {23, 12, 2};
    // runs: gaps between low-quality runs shrink along the read (cycle c of l: rate 0.012 -> 0.1 per base); ~91 % of the bases end
    // up in the top bin, like the >= Q30 share of a good run
    double c = 0;
    for (;;) {
        const double rate = 0.012 + 0.088 * (c / (double)l) * (c / (double)l);
        double u = rng.uni();
        if (u < 1e-12) u = 1e-12;
        c += 1.0 + (-std::log(u) / rate);
        if (c >= (double)l) break;
        const uint32_t pick = rng.below(100);
        const uint8_t q = kBin[pick < 55 ? 0 : pick < 90 ? 1 : 2];
        int run = 1 + (int)rng.below(4);
        for (int32_t i = (int32_t)c; run > 0 && i < l; ++i, --run) out[i] = q;
        c += 3;
    }
    if (rng.chance(0.04)) {                       // a read that goes bad: everything behind a cycle in the low bins
        const int32_t from = l / 3 + (int32_t)rng.below((uint32_t)(l - l / 3));
        for (int32_t i = from; i < l; ++i) out[i] = kBin[1 + (rng.below(4) == 0)];
    }
    out[0] = out[0] == 37 && rng.chance(0.5) ? 23 : out[0];   // the first cycles are a little worse
    if (reverse) std::reverse(out, out + l);      // BAM stores the qualities in reference orientation
}

bool write_streams_files(const std::vector<const ReadStream*>& ss, const std::string& fasta, const std::string& bam, int level,
                         std::string* err, const uint8_t* aux_pool, const uint64_t* aux_off, int qual_model) {
    FILE* fp = fopen(fasta.c_str(), "w");
    if (!fp) { *err = "cannot write " + fasta; return false; }
    FILE* fi = fopen((fasta + ".fai").c_str(), "w");
    if (!fi) { fclose(fp); *err = "cannot write " + fasta + ".fai"; return false; }
    const int W = 60;
    int64_t off = 0;
    std::string lines;
    BamHeader h;
    h.text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (const ReadStream* sp : ss) {
        const ReadStream& s = *sp;
        for (size_t c = 0; c < s.n_contigs(); ++c) {
            off += fprintf(fp, ">%s\n", s.names[c].c_str());
            int64_t L = s.ctg_len[c];
            fprintf(fi, "%s\t%lld\t%lld\t%d\t%d\n", s.names[c].c_str(), (long long)L, (long long)off, W, W + 1);
            const char* d = s.draft.data() + s.ctg_off[c];
            lines.clear();
            lines.reserve((size_t)L + (size_t)L / W + 2);
            for (int64_t i = 0; i < L; i += W) {
                int n = (int)std::min<int64_t>(W, L - i);
                lines.append(d + i, (size_t)n);
                lines.push_back('\n');
            }
            fwrite(lines.data(), 1, lines.size(), fp);
            off += (int64_t)lines.size();
            h.names.push_back(s.names[c]);
            h.lens.push_back((uint32_t)s.ctg_len[c]);
            h.text += "@SQ\tSN:" + s.names[c] + "\tLN:" + std::to_string(s.ctg_len[c]) + "\n";
        }
    }
    fclose(fp);
    fclose(fi);
    std::vector<size_t> serial0(ss.size() + 1, 0);
    std::vector<int32_t> tid0(ss.size() + 1, 0);
    for (size_t k = 0; k < ss.size(); ++k) {
        serial0[k + 1] = serial0[k] + ss[k]->n_reads();
        tid0[k + 1] = tid0[k] + (int32_t)ss[k]->n_contigs();
    }
    std::vector<uint8_t> qtmp;
    auto write_records = [&](BamWriter& w, size_t k, std::vector<uint8_t>& qbuf) -> bool {
        const ReadStream& s = *ss[k];
        const bool have_q = !s.qual.empty();
        char qn[32];
        size_t serial = serial0[k];
        for (size_t i = 0; i < s.n_reads(); ++i, ++serial) {
            snprintf(qn, sizeof(qn), "r%zu", serial);
            const uint8_t* q = have_q ? s.qual.data() + s.qual_off[i] : nullptr;
            if (!have_q && qual_model == 1) {
                qbuf.resize((size_t)s.l_qseq[i] + 1);
                synth_binned_qualities(serial, (s.flag[i] & 0x10) != 0, s.l_qseq[i], qbuf.data());
                q = qbuf.data();
            } else if (!have_q && qual_model == 2) {      // uniformly random in [25, 40]: the worst case for a DEFLATE decoder
                qbuf.resize((size_t)s.l_qseq[i] + 1);
                Rng qr(0xabcdef12ULL ^ (serial * 0x9e3779b97f4a7c15ULL));
                for (int32_t t = 0; t < s.l_qseq[i]; ++t) qbuf[(size_t)t] = (uint8_t)(25 + qr.below(16));
                q = qbuf.data();
            }
            const int32_t tid = tid0[k] + (int32_t)s.ctg[i];
            if (!w.write(tid, s.pos[i], s.mapq[i], s.flag[i], tid, s.pos[i], s.isize[i], qn,
                         s.cigar.data() + s.cigar_off[i], s.n_cigar[i], s.seq.data() + s.seq_off[i], q, s.l_qseq[i],
                         aux_pool ? aux_pool + aux_off[serial] : nullptr, aux_pool ? (size_t)(aux_off[serial + 1] - aux_off[serial]) : 0))
                return false;
        }
        return true;
    };
    const unsigned nt = std::min<unsigned>(host_threads(), (unsigned)ss.size());
    if (nt <= 1) {       // one stream (or one thread): one writer, its deflate batches on the I/O threads
        BamWriter w;
        if (!w.open(bam, h, level)) { *err = "cannot write " + bam; return false; }
        for (size_t k = 0; k < ss.size(); ++k)
            if (!write_records(w, k, qtmp)) { *err = "BAM write failed"; return false; }
        if (!w.close()) { *err = "BAM/BAI close failed"; return false; }
        return true;
    }
    // Several streams: every stream is a part of the file with its own reference sequences, so the parts are serialised and
    // deflated side by side (one thread each, largest first, compressed bytes in memory), placed behind the header with parallel
    // positioned writes, and their index entries shifted by where the part landed.
    std::vector<std::vector<uint8_t>> part(ss.size());
    std::vector<BamWriter::PartIndex> pix(ss.size());
    std::vector<size_t> by_size(ss.size());
    std::iota(by_size.begin(), by_size.end(), (size_t)0);
    std::sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) { return ss[a]->n_reads() > ss[b]->n_reads(); });
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    {
        std::vector<std::thread> th;
        auto work = [&]() {
            std::vector<uint8_t> qbuf;
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= by_size.size() || !ok) return;
                const size_t k = by_size[j];
                BamWriter w;
                if (!w.open_part((size_t)tid0[ss.size()], level) || !write_records(w, k, qbuf) || !w.finish_part()) { ok = false; return; }
                part[k].swap(w.part_bytes());
                pix[k] = w.take_part_index();
            }
        };
        if (nt > 1) np::spawn_helpers(th, nt - 1, work);
        work();
        for (std::thread& t : th) t.join();
    }
    if (!ok) { *err = "BAM write failed"; return false; }
    std::vector<uint8_t> head;
    {
        BamWriter hw;       // header only: BAM magic, text, reference list, on blocks of their own like samtools writes them
        std::string tmp = bam + ".hdr.tmp";
        if (!hw.open(tmp, h, level) || !hw.close()) { *err = "cannot write " + bam; return false; }
        FILE* f = fopen(tmp.c_str(), "rb");
        if (!f) { *err = "cannot write " + bam; return false; }
        fseek(f, 0, SEEK_END);
        head.resize((size_t)ftell(f));
        fseek(f, 0, SEEK_SET);
        const bool rd = fread(head.data(), 1, head.size(), f) == head.size();
        fclose(f);
        remove(tmp.c_str());
        remove((tmp + ".bai").c_str());
        if (!rd || head.size() < 28) { *err = "cannot write " + bam; return false; }
        head.resize(head.size() - 28);      // without its EOF marker
    }
    std::vector<uint64_t> at(ss.size() + 1, head.size());
    for (size_t k = 0; k < ss.size(); ++k) at[k + 1] = at[k] + part[k].size();
    static const uint8_t eof_marker[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43,
                                           0x02, 0x00, 0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0};
    const int fd = ::open(bam.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { *err = "cannot write " + bam; return false; }
    auto put = [&](const uint8_t* p, size_t n, uint64_t off) {
        while (n) {
            const ssize_t g = pwrite(fd, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n, (off_t)off);
            if (g <= 0) return false;
            p += g; n -= (size_t)g; off += (uint64_t)g;
        }
        return true;
    };
    bool wrote = put(head.data(), head.size(), 0) && put(eof_marker, 28, at[ss.size()]);
    next = 0;
    {
        std::vector<std::thread> th;
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= ss.size()) return;
                if (!put(part[k].data(), part[k].size(), at[k])) ok = false;
                std::vector<uint8_t>().swap(part[k]);
            }
        };
        if (nt > 1) np::spawn_helpers(th, nt - 1, work);
        work();
        for (std::thread& t : th) t.join();
    }
    wrote = (::close(fd) == 0) && wrote && ok;
    if (!wrote) { *err = "BAM write failed"; return false; }
    // one index: the reference sequences of part k are its own; shift its offsets by the part's place in the file
    BamWriter::PartIndex all;
    const size_t n_ref = (size_t)tid0[ss.size()];
    all.refs.resize(n_ref); all.n_mapped.assign(n_ref, 0); all.n_unmapped.assign(n_ref, 0); all.ref_beg.assign(n_ref, 0); all.ref_end.assign(n_ref, 0);
    for (size_t k = 0; k < ss.size(); ++k) {
        const uint64_t sh = at[k] << 16;
        all.n_no_coor += pix[k].n_no_coor;
        for (int32_t t = tid0[k]; t < tid0[k + 1]; ++t) {
            BaiRef& r = pix[k].refs[(size_t)t];
            for (auto& kv : r.bins)
                for (BaiChunk& c : kv.second) { c.beg += sh; c.end += sh; }
            for (voff_t& v : r.linear)
                if (v != (voff_t)-1) v += sh;
            all.refs[(size_t)t] = std::move(r);
            all.n_mapped[(size_t)t] = pix[k].n_mapped[(size_t)t];
            all.n_unmapped[(size_t)t] = pix[k].n_unmapped[(size_t)t];
            const bool has = all.n_mapped[(size_t)t] + all.n_unmapped[(size_t)t] > 0;
            all.ref_beg[(size_t)t] = has ? pix[k].ref_beg[(size_t)t] + sh : 0;
            all.ref_end[(size_t)t] = has ? pix[k].ref_end[(size_t)t] + sh : 0;
        }
    }
    if (!BamWriter::write_bai(bam + ".bai", all)) { *err = "BAM/BAI close failed"; return false; }
    return true;
}

bool write_stream_files(const ReadStream& s, const std::string& fasta, const std::string& bam, int level,
                        std::string* err, const uint8_t* aux_pool, const uint64_t* aux_off) {
    return write_streams_files({&s}, fasta, bam, level, err, aux_pool, aux_off, 0);
}

}  // namespace np


// Diploid workload of task 3 (snp_phase): one draft per contig, short read pairs and long reads drawn from two haplotypes that
// differ by substitutions and small indels, the draft carrying its own errors.  Every read's CIGAR follows from the column-wise
// relation of its haplotype to the draft.  Two streams over the same contigs: `sr` and `lr`, both with qualities.
namespace np {
namespace {
struct DCol { int32_t d; char b; };   // draft coordinate (-1: the base is absent from the draft) and the base

void mutate_cols(Rng& rng, const std::string& seq, double sub, double ins, double dele, int max_indel, std::vector<std::pair<int32_t, char>>* out) {
    const int32_t n = (int32_t)seq.size();
    for (int32_t i = 0; i < n;) {
        const double x = rng.uni();
        if (x < sub) {
            char b;
            do b = kBases[rng.below(4)]; while (b == seq[(size_t)i]);
            out->push_back({i, b}); ++i;
        } else if (x < sub + ins) {
            const uint32_t k = 1 + rng.below((uint32_t)max_indel);
            for (uint32_t t = 0; t < k; ++t) out->push_back({-1, kBases[rng.below(4)]});
            out->push_back({i, seq[(size_t)i]}); ++i;
        } else if (x < sub + ins + dele) {
            i += 1 + (int32_t)rng.below((uint32_t)max_indel);
        } else {
            out->push_back({i, seq[(size_t)i]}); ++i;
        }
    }
}

struct DRead { int32_t pos; std::vector<uint32_t> cig; std::string seq; };

// columns [lo, hi) of a haplotype -> position, CIGAR and bases against the draft; false when no column is aligned
bool read_from_cols(Rng& rng, const DCol* seg, int32_t n, double err, DRead* out) {
    int32_t a = -1, z = -1;
    for (int32_t k = 0; k < n; ++k) if (seg[k].d >= 0) { if (a < 0) a = k; z = k; }
    if (a < 0) return false;
    out->cig.clear();
    out->seq.clear();
    auto push_op = [&](uint32_t op, uint32_t len) {
        if (!len) return;
        if (!out->cig.empty() && (out->cig.back() & 0xf) == op) out->cig.back() += len << 4;
        else out->cig.push_back(len << 4 | op);
    };
    push_op(4, (uint32_t)a);
    for (int32_t k = 0; k < a; ++k) out->seq.push_back(seg[k].b);
    out->pos = seg[a].d;
    int32_t cur = out->pos;
    bool any_m = false;
    for (int32_t k = a; k <= z; ++k) {
        char b = seg[k].b;
        if (rng.chance(err)) { char c; do c = kBases[rng.below(4)]; while (c == b); b = c; }
        if (seg[k].d < 0) {
            push_op(1, 1);
        } else {
            if (seg[k].d > cur) push_op(2, (uint32_t)(seg[k].d - cur));
            push_op(0, 1);
            cur = seg[k].d + 1;
            any_m = true;
        }
        out->seq.push_back(b);
    }
    push_op(4, (uint32_t)(n - z - 1));
    for (int32_t k = z + 1; k < n; ++k) out->seq.push_back(seg[k].b);
    return any_m && out->cig.size() <= 0xffffu;
}

void append_read(np::ReadStream* out, uint32_t c, const DRead& r, uint16_t flag, uint8_t mapq, int32_t isize, Rng& rng, uint32_t qlo, uint32_t qspan) {
    out->pos.push_back(r.pos);
    out->ctg.push_back(c);
    out->flag.push_back(flag);
    out->n_cigar.push_back((uint32_t)r.cig.size());
    out->l_qseq.push_back((int32_t)r.seq.size());
    out->mapq.push_back(mapq);
    out->isize.push_back(isize);
    out->cigar_off.push_back(out->cigar.size());
    out->seq_off.push_back(out->seq.size());
    out->cigar.insert(out->cigar.end(), r.cig.begin(), r.cig.end());
    for (size_t k = 0; k < r.seq.size(); k += 2)
        out->seq.push_back((uint8_t)(nt16(r.seq[k]) << 4 | (k + 1 < r.seq.size() ? nt16(r.seq[k + 1]) : 0)));
    out->qual_off.push_back(out->qual.size());
    for (size_t k = 0; k < r.seq.size(); ++k) out->qual.push_back((uint8_t)(qlo + rng.below(qspan)));
}
}  // namespace

bool synth_diploid_streams(const np1_diploid_params& p, const std::string& prefix, ReadStream* sr, ReadStream* lr) {
    sr->clear();
    lr->clear();
    Rng rng(p.seed);
    for (ReadStream* o : {sr, lr}) { o->ctg_off.push_back(0); }
    for (int32_t c = 0; c < p.n_contigs; ++c) {
        const int32_t Lt = p.contig_len[c];
        if (Lt < 400) return false;
        std::string truth((size_t)Lt, 'A');
        for (int32_t i = 0; i < Lt; ++i) truth[(size_t)i] = kBases[rng.below(4)];
        std::vector<std::pair<int32_t, char>> hap[2], dr;
        hap[0].reserve((size_t)Lt);
        for (int32_t i = 0; i < Lt; ++i) hap[0].push_back({i, truth[(size_t)i]});
        mutate_cols(rng, truth, p.het_sub, p.het_indel, p.het_indel, 3, &hap[1]);
        mutate_cols(rng, truth, p.draft_err * 0.5, p.draft_err * 0.25, p.draft_err * 0.25, 3, &dr);
        std::string draft;
        std::vector<int32_t> d_of_t((size_t)Lt, -1);
        for (size_t k = 0; k < dr.size(); ++k) {
            draft.push_back(dr[k].second);
            if (dr[k].first >= 0) d_of_t[(size_t)dr[k].first] = (int32_t)k;
        }
        std::vector<DCol> cols[2];
        for (int h = 0; h < 2; ++h) {
            cols[h].reserve(hap[h].size());
            for (auto& e : hap[h]) cols[h].push_back(DCol{e.first >= 0 ? d_of_t[(size_t)e.first] : -1, e.second});
        }
        std::vector<std::pair<int32_t, int32_t>> holes;
        for (int32_t k = 0; k < p.sr_holes; ++k) {
            const int32_t a = (int32_t)rng.below((uint32_t)std::max(1, Lt - 200));
            holes.push_back({a, a + 40 + (int32_t)rng.below(161)});
        }
        struct Tmp { DRead r; uint16_t flag; uint8_t mapq; int32_t isize; };
        std::vector<Tmp> srs, lrs;
        const int32_t RL = p.read_len;
        const int64_t n_pairs = (int64_t)(p.sr_depth * Lt / (2.0 * RL));
        static const uint8_t kMapq[6] = {60, 60, 60, 40, 20, 3};
        for (int64_t t = 0; t < n_pairs; ++t) {
            const int h = (int)rng.below(2);
            const int32_t n = (int32_t)cols[h].size();
            const int32_t f = std::max(RL + 1, (int32_t)std::lround(p.frag_mean + 30.0 * rng.normal()));
            if (f >= n) continue;
            const int32_t s = (int32_t)rng.below((uint32_t)(n - f));
            bool in_hole = false;
            for (auto& hl : holes) in_hole = in_hole || (hl.first <= s && s <= hl.second) || (hl.first <= s + f && s + f <= hl.second);
            if (in_hole) continue;
            Tmp m[2];
            if (!read_from_cols(rng, cols[h].data() + s, RL, p.sr_err, &m[0].r) || !read_from_cols(rng, cols[h].data() + s + f - RL, RL, p.sr_err, &m[1].r)) continue;
            const int32_t isz = m[1].r.pos + RL - m[0].r.pos;
            for (int k = 0; k < 2; ++k) {
                m[k].flag = (uint16_t)(0x1 | 0x2 | (k == 0 ? 0x40 | 0x20 : 0x80 | 0x10));
                m[k].mapq = kMapq[rng.below(6)];
                m[k].isize = k == 0 ? isz : -isz;
                srs.push_back(m[k]);
            }
        }
        const int64_t n_lr = (int64_t)(p.lr_depth * Lt / p.lr_len);
        std::vector<DCol> seg;
        static const uint8_t kMapqL[4] = {60, 60, 30, 10};
        for (int64_t t = 0; t < n_lr; ++t) {
            const int h = (int)rng.below(2);
            const int32_t n = (int32_t)cols[h].size();
            const int32_t ln = std::min(n - 1, std::max(200, (int32_t)std::lround(p.lr_len + p.lr_len / 3.0 * rng.normal())));
            const int32_t s = (int32_t)rng.below((uint32_t)(n - ln));
            seg.clear();
            for (int32_t k = s; k < s + ln; ++k) {   // long-read indel noise: dropped / extra columns
                const double x = rng.uni();
                if (x < p.lr_err * 0.3) continue;
                if (x < p.lr_err * 0.6) seg.push_back(DCol{-1, kBases[rng.below(4)]});
                seg.push_back(cols[h][(size_t)k]);
            }
            Tmp m;
            if (!read_from_cols(rng, seg.data(), (int32_t)seg.size(), p.lr_err * 0.4, &m.r)) continue;
            m.flag = rng.chance(0.5) ? 16 : 0;
            m.mapq = kMapqL[rng.below(4)];
            m.isize = 0;
            lrs.push_back(std::move(m));
        }
        char nm[64];
        snprintf(nm, sizeof(nm), "%s%04d", prefix.c_str(), c + 1);
        for (int w = 0; w < 2; ++w) {
            ReadStream* o = w == 0 ? sr : lr;
            std::vector<Tmp>& v = w == 0 ? srs : lrs;
            std::vector<uint32_t> order(v.size());
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return v[a].r.pos < v[b].r.pos; });
            o->names.push_back(nm);
            o->ctg_len.push_back((int32_t)draft.size());
            o->draft += draft;
            o->ctg_off.push_back((uint32_t)o->draft.size());
            o->read_begin.push_back(o->n_reads());
            for (uint32_t oi : order) append_read(o, (uint32_t)c, v[oi].r, v[oi].flag, v[oi].mapq, v[oi].isize, rng, w == 0 ? 20u : 5u, 21u);
        }
    }
    sr->read_begin.push_back(sr->n_reads());
    lr->read_begin.push_back(lr->n_reads());
    return true;
}
}  // namespace np
