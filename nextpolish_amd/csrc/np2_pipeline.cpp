// Long-read consensus of one contig: the host side of ctg_cns_core (reference: source/lib/ctg_cns.c:3399-3623).
//
// Host work (order dependent, tiny per record): k-way merge of the region iterators over the listed BAMs
// (bsort.c:174-199,1202-1484: order by position, strand, file index), per-record filters and the SA-tag gap test
// (ctg_cns.c:3475-3526), window bookkeeping (cal_win_len :2800-2807, overlap 1 Mb), low-quality region detection
// (:1562-1725) and the stitching of neighbouring windows (link_consensus :3121-3223).
// Everything per alignment column -- spans, tags, link graph, chain DP, backtrace -- runs in the window executor
// (np2_exec.h): HIP kernels in the product.
//
// Low-quality regions go through np2_lq.cpp (candidates, POA pseudo-seed, O(ND) alignment on the host; the graph
// consensus of the concatenated regions in the executor).
// The split-read structural layer (depth track, gap clusters, supplementary streams, split points) is in np2_sv.cpp.  See DESIGN.md section "path B".
#include <algorithm>
#include <cassert>
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <time.h>
#include <unistd.h>
#include <string>
#include <vector>

#include "../../include/nextpolish2.h"
#include "np2_exec.h"
#include "np2_lq.h"
#include "np2_sv.h"
#include "np_bam.h"

// storage hooks of the big record arrays (np2_exec.h): malloc unless an executor installs its own
void* (*np2::BigMem::make)(size_t) = nullptr;
void (*np2::BigMem::drop)(void*) = nullptr;

namespace {

thread_local std::string g_err;

using np2::ConsBase;
using np2k::ColStat;

struct Pos { uint32_t s, e; };
struct Gap { Pos gap; uint32_t fs, ds, score; };

// ---- per-record helpers (ctg_cns.c:2297-2401)
uint32_t cigar_clip(const uint32_t* cigar, uint32_t n, int end) {   // cigarint2ul
    const uint32_t c = cigar[end ? n - 1 : 0];
    const uint32_t op = c & 0xf;
    return (op == 4 || op == 5) ? c >> 4 : 0;
}
int32_t full_query_len(const np::BamRec& r) {   // cal_l_qseq / cal_l_qseq_from_cigar
    const uint32_t* cg = r.cigar();
    if (!r.l_qseq) {
        uint32_t rlen = 0;
        for (uint32_t i = 0; i < r.n_cigar; ++i) {
            const uint32_t op = cg[i] & 0xf;
            if (op == 4 || op == 5 || op == 0 || op == 7 || op == 8 || op == 1) rlen += cg[i] >> 4;
        }
        return (int32_t)rlen;
    }
    if ((cg[0] & 0xf) == 4) return r.l_qseq;
    int32_t rlen = r.l_qseq;
    if ((cg[0] & 0xf) == 5) rlen += (int32_t)(cg[0] >> 4);
    const uint32_t last = cg[r.n_cigar - 1];
    if ((last & 0xf) == 5) rlen += (int32_t)(last >> 4);
    return rlen;
}
// What the split-read logic needs from the CIGAR text of an SA:Z entry, found in one pass over its <length><op> pairs: the clip (H or S)
// the text opens with, the clip it ends in, and the contig bases it covers.  Counted like the reference counts them (cigarstr2ul /
// cigarstr2rlen, ctg_cns.c:2368-2401): only M and D advance on the contig (N, = and X do not), and a text that ends in digits without an
// operation has no closing clip.
struct SaCigar { uint32_t lead_clip = 0, tail_clip = 0, ref_span = 0; };
SaCigar read_sa_cigar(const char* text) {
    SaCigar c;
    uint32_t len = 0, last_len = 0;
    char last_op = 0;
    bool first = true;
    for (const char* p = text; *p; ++p) {
        if (*p >= '0' && *p <= '9') { len = len * 10 + (uint32_t)(*p - '0'); last_op = 0; continue; }
        if (*p == 'M' || *p == 'D') c.ref_span += len;
        if (first && (*p == 'H' || *p == 'S')) c.lead_clip = len;
        first = false;
        last_op = *p;
        last_len = len;
        len = 0;
    }
    if (last_op == 'H' || last_op == 'S') c.tail_clip = last_len;
    return c;
}

// One aligned piece of a read: where it lies on the contig and which part of the (unclipped) read it covers.
struct Piece { Pos ref, read; };
inline uint32_t span_between(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }

// Does the pair (this record, one same-strand SA entry on the same contig) look like ONE read torn apart by an insertion or deletion
// (ctg_cns.c:2463-2492)?  With `a` the piece that starts first on the contig and `b` the other: b must end later than a on the contig and
// on the read, the two together must reach both ends of the read to within a tenth of its length, and the jump between them must stay
// under 30 kb on both axes.  The cost of the pairing is what is left uncovered plus both jumps; the cheapest pairing of a record wins, and
// the gap it brackets on the contig lies between a's end and b's start (whichever comes first).
void consider_split(Gap* best, int32_t read_len, const Piece& self, const Piece& other) {
    const bool other_first = self.ref.s > other.ref.s;
    const Piece& a = other_first ? other : self;
    const Piece& b = other_first ? self : other;
    if (a.ref.s == b.ref.s) return;
    if (!(b.ref.e > a.ref.e && b.read.e > a.read.e)) return;
    const int32_t edge = (int32_t)(read_len * 0.1);
    if (!(a.read.s < (uint32_t)edge && b.read.e > (uint32_t)(read_len - edge))) return;
    const uint32_t ref_jump = span_between(b.ref.s, a.ref.e), read_jump = span_between(b.read.s, a.read.e);
    if (ref_jump >= 30000 || read_jump >= 30000) return;
    const uint32_t cost = a.read.s + (uint32_t)read_len - b.read.e + ref_jump + read_jump;
    if (best->score && cost >= best->score) return;
    best->score = cost;
    best->ds = other.read.s;      // (the reference keeps the starts of the piece the SA entry describes)
    best->fs = other.ref.s;
    best->gap.s = std::min(a.ref.e, b.ref.s);
    best->gap.e = std::max(a.ref.e, b.ref.s);
}

// SA:Z value of a record (SAMv1 4.2.4 aux layout), nullptr when absent
const char* find_sa(const np::BamRec& r) {
    const uint8_t* p = r.qual() + r.l_qseq;
    const uint8_t* end = r.end();
    while (p + 3 <= end) {
        const char t0 = (char)p[0], t1 = (char)p[1], ty = (char)p[2];
        p += 3;
        if (t0 == 'S' && t1 == 'A' && ty == 'Z') return reinterpret_cast<const char*>(p);
        switch (ty) {
            case 'A': case 'c': case 'C': p += 1; break;
            case 's': case 'S': p += 2; break;
            case 'i': case 'I': case 'f': p += 4; break;
            case 'Z': case 'H': while (p < end && *p) ++p; ++p; break;
            case 'B': {
                if (p + 5 > end) return nullptr;
                const char sub = (char)p[0];
                uint32_t cnt;
                memcpy(&cnt, p + 1, 4);
                const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                p += 5 + w * cnt;
                break;
            }
            default: return nullptr;
        }
    }
    return nullptr;
}

// ---- merged region iteration over the BAM list (bsort.c:1202-1484)
struct FileIter {
    np::BamReader rd;
    int tid = -1;
    bool have = false, finished = false;
    np::BamRec rec;
};
class MergeIter {
  public:
    bool open(const std::string& list_path, const std::string& ctg, int32_t beg, int32_t end, std::string* err) {
        std::ifstream in(list_path);
        if (!in) { *err = "cannot read BAM list " + list_path; return false; }
        std::string line;
        std::vector<std::string> paths;
        while (std::getline(in, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
            if (!line.empty()) paths.push_back(line);
        }
        beg_ = beg; end_ = end;
        files_.resize(paths.size());
        for (size_t i = 0; i < paths.size(); ++i) {
            files_[i].reset(new FileIter());
            FileIter& f = *files_[i];
            if (!f.rd.open(paths[i])) { *err = "fail to open \"" + paths[i] + "\""; return false; }
            // records are consumed before the next one of their file is asked for (next() advances the file of the record handed out
            // last), so they may point into the reader's window: one copy of every record less (NP2_ZERO_COPY=0: owned copies)
            static const bool zero_copy = !(getenv("NP2_ZERO_COPY") && getenv("NP2_ZERO_COPY")[0] == '0');
            f.rd.set_zero_copy(zero_copy);
            f.tid = f.rd.header().name2id(ctg);
            np::BaiIndex bai;
            if (!bai.load(paths[i] + ".bai")) { *err = "failed to load index for " + paths[i]; return false; }
            if (f.tid < 0) { f.finished = true; continue; }
            np::voff_t v;
            if (!bai.region_start(f.tid, beg, end, &v)) { f.finished = true; continue; }
            if (!f.rd.seek(v)) { *err = "seek failed in " + paths[i]; return false; }
            if (!advance(f, err)) return false;
        }
        return true;
    }
    // next record in (position, strand, file) order; nullptr at the end
    const np::BamRec* next(std::string* err) {
        if (last_ >= 0) {
            if (!advance(*files_[(size_t)last_], err)) { failed_ = true; return nullptr; }
        }
        int best = -1;
        for (size_t i = 0; i < files_.size(); ++i) {
            FileIter& f = *files_[i];
            if (!f.have) continue;
            if (best < 0) { best = (int)i; continue; }
            const FileIter& b = *files_[(size_t)best];
            const uint32_t pa = (uint32_t)(f.rec.pos + 1), pb = (uint32_t)(b.rec.pos + 1);
            if (pa != pb) { if (pa < pb) best = (int)i; continue; }
            const int ra = (f.rec.flag & 16) ? 1 : 0, rb = (b.rec.flag & 16) ? 1 : 0;
            if (ra != rb) { if (ra < rb) best = (int)i; continue; }
        }
        last_ = best;
        return best < 0 ? nullptr : &files_[(size_t)best]->rec;
    }
    bool failed() const { return failed_; }

  private:
    bool advance(FileIter& f, std::string* err) {   // next record of this file overlapping [beg, end) (hts_itr_next semantics)
        f.have = false;
        while (!f.finished) {
            const int rc = f.rd.next(f.rec);
            if (rc == 0) { f.finished = true; break; }
            if (rc < 0) { *err = "truncated BAM"; return false; }
            if (f.rec.tid != f.tid || f.rec.pos >= end_) { f.finished = true; break; }
            if (f.rec.endpos() > beg_) { f.have = true; break; }
        }
        return true;
    }
    std::vector<std::unique_ptr<FileIter>> files_;
    int32_t beg_ = 0, end_ = 0;
    int last_ = -1;
    bool failed_ = false;
};

int cal_win_len(int w, int s, uint64_t l) {   // ctg_cns.c:2800-2807 (float arithmetic as written there)
    int b = (int)l;
    if (l > (uint64_t)w) {
        const int n = (int)((float)(l - (uint64_t)s) / (float)(w - s) + 0.999);
        b = (int)((float)(l + (uint64_t)((n - 1) * s)) / (float)n + 0.999);
    }
    return b;
}

// ---- low-quality regions of a window consensus (get_l_del_regions / get_lqseqs_from_gap, ctg_cns.c:1562-1725)
struct Del { Pos gap; int l; };
struct LqReg { uint32_t start, end; uint8_t l; };

struct LqCtx {
    const std::vector<ColStat>* st;
    const std::vector<ConsBase>* c;
    int reads_type;
    float gap_min_ratio1;
    const std::vector<np2::LqCluster>* clusters;
    // where the two scans have anything to look at, marked by the executor (np2_exec.h: WindowOutput::trig_*); null: test every position
    const std::vector<uint64_t>* trig_del = nullptr;
    const std::vector<uint64_t>* trig_ins = nullptr;
};

// next set bit at or above i (ascending scans) / at or below i (descending scans) of a bitmap over [0, len); len / -1 when there is none
inline int next_bit_up(const std::vector<uint64_t>& m, int i, int len) {
    while (i < len) {
        const uint64_t w = m[(size_t)i >> 6] >> (i & 63);
        if (w) { i += __builtin_ctzll(w); return i < len ? i : len; }
        i = (i | 63) + 1;
    }
    return len;
}
inline int next_bit_down(const std::vector<uint64_t>& m, int i) {
    while (i >= 0) {
        const uint64_t w = m[(size_t)i >> 6] << (63 - (i & 63));
        if (w) return i - __builtin_clzll(w);
        i = (i & ~63) - 1;
    }
    return -1;
}

int cal_del_pos(const std::vector<ColStat>& m, int s, int e) {
    int validy = 0;
    for (int i = s; i <= e; ++i)
        if (m[(size_t)i].l_del > m[(size_t)i].coverage * 0.6) ++validy;
    return validy;
}

std::vector<Del> l_del_regions(const LqCtx& x) {
    const std::vector<ColStat>& msa = *x.st;
    const std::vector<ConsBase>& cb = *x.c;
    const int len = (int)cb.size();
    std::vector<Del> dels;
    int ps = 0, pe = 0;
    auto st = [&](int i) -> const ColStat& { return msa[(size_t)cb[(size_t)i].pos]; };
    const bool marked = x.trig_del && x.trig_del->size() * 64 >= (size_t)len;
    for (int i = 1; i < len; ++i) {
        if (marked) {      // (the executor evaluated the loop head for every position; only the marked ones are visited)
            i = next_bit_up(*x.trig_del, i, len);
            if (i >= len) break;
        } else if (st(i).l_del < st(i).coverage * 0.3 && cb[(size_t)i].pos < cb[(size_t)i - 1].pos + 20) continue;
        if (i >= ps && i <= pe) continue;
        int s = i - 1;
        while (s > 0 && st(s).l_del > st(s).coverage * 0.3) --s;
        int e = i + 1;
        while (e < len - 1 && st(e).l_del > st(e).coverage * 0.3) ++e;
        if (cb[(size_t)e].pos - cb[(size_t)s].pos < 10) continue;
        int p = cal_del_pos(msa, (int)cb[(size_t)s].pos, (int)cb[(size_t)e].pos);
        int l = (int)(cb[(size_t)e].pos - cb[(size_t)s].pos + 1);
        if ((x.reads_type == np2k::READS_CLR || x.reads_type == np2k::READS_RS) && p < l * 0.05) continue;
        l = p > l / 3 ? 2 : 3;
        ps = s;
        pe = e;
        for (p = 0, s = i - 0; s > 0; --s) {   // LQSEQ_MIN_LEN / 2 == 0
            if (cb[(size_t)s].qv >= 60 && st(s).l_del < st(s).coverage * 0.3) ++p; else p = 0;
            if (p >= 4 && np2k::base_to_int((unsigned char)cb[(size_t)s].base) != np2k::base_to_int((unsigned char)cb[(size_t)s - 1].base) &&
                st(s).l_ins <= 0) break;
        }
        for (p = 0, e = i + 0; e < len - 1; ++e) {
            if (cb[(size_t)e].qv >= 60 && st(e).l_del < st(e).coverage * 0.3) ++p; else p = 0;
            if (p >= 4 && np2k::base_to_int((unsigned char)cb[(size_t)e].base) != np2k::base_to_int((unsigned char)cb[(size_t)e + 1].base) &&
                st(e).l_ins <= 0) break;
        }
        s = s >= 0 ? (int)cb[(size_t)s].pos : (int)cb[0].pos;
        e = e < len - 1 ? (int)cb[(size_t)e].pos : (int)cb[(size_t)len - 1].pos;
        if (e - s < 20) continue;
        if (dels.empty() || s > (int)dels.back().gap.e) dels.push_back(Del{Pos{(uint32_t)s, (uint32_t)e}, l});
        else dels.back().gap.e = (uint32_t)e;
    }
    return dels;
}

int lq_from_dels(const Del& d, std::vector<LqReg>& lq, int index) {   // get_lqseqs_from_dels
    if (index >= 0) {
        const uint32_t s = d.gap.s < lq[(size_t)index].start ? d.gap.s : lq[(size_t)index].start;
        while (index > 0 && lq[(size_t)index].start <= d.gap.e && !lq[(size_t)index].l) --index;
        if (lq[(size_t)index].start > d.gap.e) { ++index; lq[(size_t)index].end = 0; }
        else if (lq[(size_t)index].l) return index;
        lq[(size_t)index].start = s;
        lq[(size_t)index].end = d.gap.e > lq[(size_t)index].end ? d.gap.e : lq[(size_t)index].end;
        lq[(size_t)index].l = (uint8_t)d.l;
    } else {
        ++index;
        lq[(size_t)index].start = d.gap.s;
        lq[(size_t)index].end = d.gap.e;
        lq[(size_t)index].l = (uint8_t)d.l;
    }
    return index;
}

int lq_from_cluster(const np2::LqCluster& clu, std::vector<LqReg>& lq, int index) {   // get_lqseqs_from_cluster, ctg_cns.c:1512-1530
    if (clu.i_m) {
        if (index >= 0) {
            while (index > 0 && lq[(size_t)index].start <= clu.re) --index;
            if (lq[(size_t)index].start > clu.re) ++index;
            if ((size_t)index >= lq.size()) lq.resize((size_t)index + 100);
            lq[(size_t)index].start = clu.rs;
            lq[(size_t)index].end = clu.re;
            lq[(size_t)index].l = 1;
        } else {
            ++index;
            lq[(size_t)index].start = clu.rs;
            lq[(size_t)index].end = clu.re;
            lq[(size_t)index].l = 1;
        }
    }
    return index;
}

// regions in DEscending order of position, like the reference builds them
std::vector<LqReg> lq_regions(const LqCtx& x) {
    const std::vector<ColStat>& msa = *x.st;
    const std::vector<ConsBase>& cb = *x.c;
    const int len = (int)cb.size();
    std::vector<LqReg> lq(200);
    lq[0].start = lq[0].end = 0;
    int index = 0;
    std::vector<Del> dels = l_del_regions(x);
    int dels_i = (int)dels.size();
    const std::vector<np2::LqCluster>& clusters = *x.clusters;
    int clusters_i = (int)clusters.size();
    auto st = [&](int i) -> const ColStat& { return msa[(size_t)cb[(size_t)i].pos]; };
    const bool marked = x.trig_ins && x.trig_ins->size() * 64 >= (size_t)len;
    for (int i = len - 1; i >= 0; --i) {
        if (marked) {
            i = next_bit_down(*x.trig_ins, i);
            if (i < 0) break;
        } else if ((float)st(i).l_ins < (float)st(i).coverage * x.gap_min_ratio1) continue;
        if (st(i).l_ins < st(i).coverage * 0.1) {
            const int s0 = (int)cb[(size_t)i].pos - 10;
            const int e0 = (int)cb[(size_t)i].pos + 10;
            int l_ins = st(i).l_ins;
            for (int p = i - 1; p >= 0 && cb[(size_t)p].pos >= (uint32_t)s0; --p)   // unsigned compare, as in the reference
                if (cb[(size_t)p].pos != cb[(size_t)p + 1].pos) l_ins += st(p).l_ins;
            for (int p = i + 1; p < len && cb[(size_t)p].pos <= (uint32_t)e0; ++p)
                if (cb[(size_t)p].pos != cb[(size_t)p - 1].pos) l_ins += st(p).l_ins;
            if (l_ins < st(i).coverage * 0.6) continue;
        }
        int p, s, e;
        for (p = 0, s = i; s > 0; --s) {
            if (cb[(size_t)s].qv >= 60) ++p; else p = 0;
            if (p >= 4 && np2k::base_to_int((unsigned char)cb[(size_t)s].base) != np2k::base_to_int((unsigned char)cb[(size_t)s - 1].base) &&
                st(s).l_ins <= 0) break;
        }
        for (p = 0, e = i; e < len - 1; ++e) {
            if (cb[(size_t)e].qv >= 60) ++p; else p = 0;
            if (p >= 4 && np2k::base_to_int((unsigned char)cb[(size_t)e].base) != np2k::base_to_int((unsigned char)cb[(size_t)e + 1].base) &&
                st(e).l_ins <= 0) break;
        }
        s = s >= 0 ? (int)cb[(size_t)s].pos : (int)cb[0].pos;
        e = e < len - 1 ? (int)cb[(size_t)e].pos : (int)cb[(size_t)len - 1].pos;
        if (index == 0 || e + 30 < (int64_t)lq[(size_t)index - 1].start) {
            while (dels_i && e < (int64_t)dels[(size_t)dels_i - 1].gap.s) {
                index = lq_from_dels(dels[(size_t)dels_i - 1], lq, index - 1);
                --dels_i;
                if (++index >= (int)lq.size()) lq.resize(lq.size() + 100);
            }
            while (clusters_i > 0 && (uint32_t)e < clusters[(size_t)clusters_i - 1].rs) {
                index = lq_from_cluster(clusters[(size_t)clusters_i - 1], lq, index - 1);
                --clusters_i;
                while (clusters_i > 0 && !clusters[(size_t)clusters_i - 1].i_m) --clusters_i;
                if (++index >= (int)lq.size()) lq.resize(lq.size() + 100);
            }
            lq[(size_t)index].start = (uint32_t)s;
            lq[(size_t)index].end = (uint32_t)e;
            lq[(size_t)index].l = 0;
            if (++index >= (int)lq.size()) lq.resize(lq.size() + 100);
        } else {
            lq[(size_t)index - 1].start = (uint32_t)s;
        }
    }
    lq.resize((size_t)index);
    return lq;
}

// HiFi: low-quality runs are found while walking the best path backwards (generate_cns_from_best_score_lq,
// ctg_cns.c:1727-1826): a run of bases with qv < 80 closed by more than 4 good bases becomes a region (l = 4), padded
// by 2 bases and merged with the previous one when they touch.  Upper case needs coverage > 4 and qv > 80.
std::vector<np2::LqRegionIn> hifi_regions(std::vector<ConsBase>* cons, const std::vector<ColStat>& st, const std::vector<np2::LqCluster>& clusters) {
    std::vector<LqReg> regs(200);
    int idx = 0;
    int clusters_i = (int)clusters.size();
    const int len = (int)cons->size();
    auto R = [&](int p) -> ConsBase& { return (*cons)[(size_t)(len - 1 - p)]; };   // backtrace order
    const int lq_min_length = 2;
    int lq = 0, lq_s = -1, lq_e = -1;
    for (int p = 0; p < len; ++p) {
        const uint32_t cov = st[R(p).pos].coverage;
        const int qv = (int)R(p).qv;
        if (cov < 4) {
            lq = 0;
            lq_s = -1;
        } else if (qv < 80) {
            if (lq_s == -1) lq_s = p;
            lq_e = p;
            lq = 1;
        } else if (lq && p - lq_e > 2 * lq_min_length && R(p).pos != R(p - 1).pos) {
            lq_e = p - lq_min_length - 1;
            lq_s = lq_s > lq_min_length ? lq_s - lq_min_length : 1;
            if (idx >= 1 && R(lq_s).pos >= regs[(size_t)idx - 1].start) {
                regs[(size_t)idx - 1].start = R(lq_e).pos;
            } else {
                while (clusters_i > 0 && R(lq_s).pos < clusters[(size_t)clusters_i - 1].rs) {
                    idx = lq_from_cluster(clusters[(size_t)clusters_i - 1], regs, idx - 1);
                    --clusters_i;
                    while (clusters_i > 0 && !clusters[(size_t)clusters_i - 1].i_m) --clusters_i;
                    if (++idx >= (int)regs.size()) regs.resize(regs.size() + 100);
                }
                regs[(size_t)idx].end = R(lq_s).pos;
                regs[(size_t)idx].start = R(lq_e).pos;
                regs[(size_t)idx].l = 4;
                if (++idx >= (int)regs.size()) regs.resize(regs.size() + 100);
            }
            lq = 0;
            lq_s = -1;
        }
        const char up = (char)toupper(R(p).base);
        R(p).base = (cov > 4 && qv > 80) ? up : (char)tolower(up);
    }
    std::vector<np2::LqRegionIn> out;
    for (int i = 0; i < idx; ++i) out.push_back(np2::LqRegionIn{regs[(size_t)i].start, regs[(size_t)i].end, regs[(size_t)i].l});
    return out;
}

struct WindowCons {
    std::vector<ConsBase> b;
    uint32_t lstrip = 0, rstrip = 0;
    int32_t uncorrected_len = 0;
};

// link_consensus (ctg_cns.c:3121-3223): windows stitched at an 8-mer shared in the overlap, cut at the split points
consensus_trimed_data* link_windows(std::vector<WindowCons>& w, const std::vector<np2::SvPos>& split_ps, int len, int k, int split, int overlap_s) {
    const int s = overlap_s / 2;
    WindowCons *consensus = nullptr, *consensusnext = nullptr;
    int l = 0;
    for (size_t i = 0; i + 1 < w.size(); ++i) {
        consensus = &w[i];
        consensusnext = &w[i + 1];
        consensus->rstrip = consensusnext->lstrip = (uint32_t)s;
        auto cpos = [&](void) -> uint32_t { return consensus->b[consensus->b.size() - consensus->rstrip].pos; };
        auto npos = [&](void) -> uint32_t { return consensusnext->b[consensusnext->lstrip].pos; };
        const uint32_t clast = consensus->b.back().pos, nfirst = consensusnext->b[0].pos;
        while (cpos() < clast - (uint32_t)s) --consensus->rstrip;
        while (cpos() > clast - (uint32_t)s) ++consensus->rstrip;
        while (npos() < nfirst + (uint32_t)s) ++consensusnext->lstrip;
        while (npos() > nfirst + (uint32_t)s) --consensusnext->lstrip;
        l = 0;
        const int p = consensusnext->uncorrected_len - consensus->uncorrected_len;
        while (l < k) {
            const int j = (int)(cpos() - npos());
            if (j == p && consensus->b[consensus->b.size() - consensus->rstrip].base == consensusnext->b[consensusnext->lstrip].base) {
                ++l;
                --consensusnext->lstrip;
                ++consensus->rstrip;
            } else {
                l = 0;
                if (j > p) ++consensusnext->lstrip;
                else if (j < p) --consensusnext->lstrip;
                else {
                    const int d = (int)(cpos() + (uint32_t)consensus->uncorrected_len - 1);
                    while ((int)(cpos() + (uint32_t)consensus->uncorrected_len) > d) ++consensus->rstrip;
                    while ((int)(npos() + (uint32_t)consensusnext->uncorrected_len) > d) --consensusnext->lstrip;
                }
            }
        }
    }
    if (w.size() > 1) {
        assert(l == k);
        consensus->rstrip -= (uint32_t)k;
        consensusnext->lstrip += (uint32_t)k;
    }
    consensus_trimed_data* out = (consensus_trimed_data*)malloc(sizeof(consensus_trimed_data));
    out->i_m = split ? (int)split_ps.size() + 1 : 1;
    out->data = (consensus_trimed*)calloc((size_t)out->i_m, sizeof(consensus_trimed));
    (void)len;
    size_t total = 0;
    for (auto& c : w) total += c.b.size();
    int index = 0;
    consensus_trimed* ct = &out->data[index++];
    ct->seq = (char*)malloc(total + 2);
    size_t li = 0;
    int sp = li < split_ps.size() ? (int)((split_ps[li].s + split_ps[li].e) / 2) : -1;
    ++li;
    for (auto& c : w) {
        const int p = c.uncorrected_len;
        const size_t end = c.b.size() - c.rstrip;
        for (size_t j = c.lstrip; j < end; ++j) {
            if (split && c.b[j].pos + (uint32_t)p >= (uint32_t)sp && j >= 1 && c.b[j - 1].pos + (uint32_t)p < (uint32_t)sp) {
                if (split == 1 && ct->len) {
                    ct->seq[ct->len] = '\0';
                    ct = &out->data[index++];
                    ct->seq = (char*)malloc(total + 2);
                } else if (split == 2) {
                    ct->seq[ct->len++] = 'N';
                }
                while (j < c.b.size() && c.b[j].pos + (uint32_t)p == (uint32_t)sp) ++j;
            } else {
                ct->seq[ct->len++] = c.b[j].base;
            }
            if (j < c.b.size() && c.b[j].pos + (uint32_t)p > (uint32_t)sp && li < split_ps.size()) {
                sp = (int)((split_ps[li].s + split_ps[li].e) / 2);
                ++li;
            }
        }
    }
    out->i_m = index;
    ct->seq[ct->len] = '\0';
    return out;
}

}  // namespace

struct ctg_cns_cfg {   // reference: ctg_cns.c:3337-3353 (only the fields this implementation needs)
    int reads_type, split;
    float ide_t;
    uint32_t ort_t, irt_t;
    int w, s;
    np2::Exec* exec;     // created lazily by the first ctg_cns_core of the process (after the caller's fork)
    int exec_pid;
};

extern "C" {

const char* np2_last_error(void) { return g_err.c_str(); }

ctg_cns_cfg* ctg_cns_init(int consensus_w, int reads_type, int split, float ide_t, float ort_t, float irt_t) {
    ctg_cns_cfg* cfg = (ctg_cns_cfg*)calloc(1, sizeof(ctg_cns_cfg));
    cfg->reads_type = reads_type;
    cfg->split = split;
    cfg->ide_t = ide_t;
    cfg->ort_t = (uint32_t)(1000 * ort_t);
    cfg->irt_t = (uint32_t)(1000 * irt_t);
    cfg->s = 1000000;
    if (!consensus_w) consensus_w = 40000000;
    else assert(consensus_w > cfg->s * 4);
    cfg->w = consensus_w;
    cfg->exec = nullptr;
    cfg->exec_pid = 0;
    return cfg;
}

void ctg_cns_destroy(ctg_cns_cfg* cfg) {
    if (!cfg) return;
    free(cfg);   // the executor (device context + buffers) belongs to the process, not to the configuration
}

void free_consensus_trimed_data(consensus_trimed_data* d) {
    for (int i = 0; i < d->i_m; ++i) free(d->data[i].seq);
    free(d->data);
    free(d);
}

}  // extern "C"

static void np2_die(const char* what, const char* ctg) {
    fprintf(stderr, "nextpolish2 (MI355X): %s (contig %s)\n", what, ctg);
    exit(1);
}

static consensus_trimed_data* ctg_cns_core_task(ctg_cns_cfg* cfg, ref_* ref, char* bam_list) {
    g_err.clear();
    if (!cfg->exec || cfg->exec_pid != (int)getpid()) {   // lazily, per process: the caller forks after ctg_cns_init
        // one executor per process, shared by every configuration and kept until the process ends: its buffers in HBM
        // only ever grow, so a worker allocates for its largest window once
        static np2::Exec* proc_exec = nullptr;
        static int proc_exec_pid = 0;
        if (!proc_exec || proc_exec_pid != (int)getpid()) {
            std::string err;
            proc_exec = np2::make_exec(&err);   // (an executor inherited through fork is abandoned, never used or freed)
            proc_exec_pid = (int)getpid();
            if (!proc_exec) { fprintf(stderr, "nextpolish2 (MI355X): %s\n", err.c_str()); exit(1); }
        }
        cfg->exec = proc_exec;
        cfg->exec_pid = proc_exec_pid;
    }
    const int reads_type = cfg->reads_type;
    const uint32_t gap_min_len = reads_type != np2k::READS_ONT ? 5 : 3;
    const float gap_min_ratio1 = reads_type != np2k::READS_ONT ? 0.3f : 0.01f;
    const float max_clip_ratio = reads_type == np2k::READS_HIFI ? 0.1f : 0.7f;

    assert(ref->length < INT_MAX);
    std::vector<char> rfseq((size_t)ref->length + 1);
    bit2seq1(ref->s, ref->length, rfseq.data());
    const int32_t b = cal_win_len(cfg->w, cfg->s, ref->length);
    np2::SvContig sv;
    sv.brk_g = ref->length > 100000 ? 1 : 0;
    if (sv.brk_g) sv.ref_ide = np2::sv_cal_ref_ide(ref->qv, ref->qv_l);
    np2::SvWindow svw;
    int32_t s = 0, e = 0;
    std::vector<WindowCons> windows;
    long fra_map = 0, total_map = 0;
    // window buffers of this process: kept between calls so the ~100 MB of records and ~80 MB of results of a 5 Mb
    // window are not re-allocated (and page-faulted in again) for every contig
    static np2::WindowInput in;
    static np2::WindowOutput out;
    static uint64_t contig_serial = 0;
    in.contig_serial = ++contig_serial;
    struct RecMeta { uint32_t l_qseq, aligned_q; Gap g; bool want_gap; };
    const bool timing = getenv("NP2_TIMING") != nullptr;   // wall time of the host stages of every window on stderr
    timespec lap_t;
    clock_gettime(CLOCK_MONOTONIC, &lap_t);
    auto lap = [&](const char* what) {
        if (!timing) return;
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        fprintf(stderr, "[np2 host] %-22s %9.2f ms\n", what, (t.tv_sec - lap_t.tv_sec) * 1e3 + (t.tv_nsec - lap_t.tv_nsec) * 1e-6);
        lap_t = t;
    };
    while (e < (int32_t)ref->length) {
        e = s + b > (int32_t)ref->length ? (int32_t)ref->length : s + b;
        const int32_t l = e - s;
        in.contig_seq = rfseq.data();
        in.s = s;
        in.e = e;
        in.gap_min_len = gap_min_len;
        in.read_type = reads_type;
        in.recs.clear();
        in.sup.clear();
        in.streams.clear();
        std::vector<RecMeta> meta;
        svw.reset(sv.brk_g ? (size_t)l / 10 + 16 : 0);
        int32_t p = 0;
        int rege = s == 0 ? (e > 15000000 ? e : 15000000) : e;
        std::string err;
        MergeIter it;
        if (!it.open(bam_list, ref->n, s > 0 ? s - 1 : 0, rege, &err)) np2_die(err.c_str(), ref->n);   // "name:s-e" -> [max(s-1,0), e)
        const np::BamRec* r;
        while ((r = it.next(&err)) != nullptr) {
            assert(r->pos >= p);
            p = r->pos;
            if (p >= e) rege = 0;
            const uint32_t* cigar = r->cigar();
            if (r->n_cigar == 0) continue;   // (the reference would read cigar[-1]; unmapped-style records carry flag 4 and are dropped below anyway)
            const int32_t l_qseq = full_query_len(*r);
            const Pos rfp1{(uint32_t)r->pos, (uint32_t)r->endpos()};
            const Pos rdp1{cigar_clip(cigar, r->n_cigar, 0), (uint32_t)l_qseq - cigar_clip(cigar, r->n_cigar, 1)};
            Gap g;
            g.score = 0;
            g.gap.s = g.gap.e = g.fs = g.ds = 0;
            const char* sa = find_sa(*r);
            if (sa) {   // set_satags (ctg_cns.c:2158-2182): rname,pos,strand,CIGAR,mapQ,NM;...
                const int strand = (r->flag & 16) ? 1 : 0;
                std::string buf(sa);
                size_t at = 0;
                while (at < buf.size()) {
                    size_t semi = buf.find(';', at);
                    if (semi == std::string::npos) semi = buf.size();
                    std::string item = buf.substr(at, semi - at);
                    at = semi + 1;
                    std::vector<std::string> f;
                    size_t a0 = 0;
                    for (;;) {
                        const size_t c = item.find(',', a0);
                        if (c == std::string::npos) { f.push_back(item.substr(a0)); break; }
                        f.push_back(item.substr(a0, c - a0));
                        a0 = c + 1;
                    }
                    if (f.size() < 4) break;
                    if (f[0] == ref->n && ((f[2][0] == '+') ? 0 : 1) == strand) {
                        const uint32_t sp = (uint32_t)(atoll(f[1].c_str()) - 1);
                        const SaCigar sc = read_sa_cigar(f[3].c_str());
                        const Piece other{Pos{sp, sp + sc.ref_span}, Pos{sc.lead_clip, (uint32_t)l_qseq - sc.tail_clip}};
                        consider_split(&g, l_qseq, Piece{rfp1, rdp1}, other);
                    }
                }
            }
            if (rege && (r->flag & 0xD04) && sv.brk_g && g.score) {   // update_sup_alns (ctg_cns.c:2660-2681)
                np2::SvSupAln sa_;
                sa_.fs = rfp1.s;
                sa_.ds = rdp1.s;
                sa_.cigar.assign(cigar, cigar + r->n_cigar);
                svw.sup_alns.push_back(std::move(sa_));
            }
            if (r->flag & 0xD04) continue;
            ++total_map;
            const double frac = (double)(rdp1.e - rdp1.s) / (double)l_qseq;
            if (frac < 0.7) ++fra_map;
            if (!g.score && frac <= max_clip_ratio) continue;
            if (sv.brk_g) {
                const np2::SvPos rp{rfp1.s, rfp1.e};
                if (!sv.rreads_w) {
                    sv.rreads.push_back(rp);
                    if (sv.rreads.size() >= 50000) {
                        sv.rreads_w = np2::sv_cal_rreads_w(sv.rreads);
                        for (const np2::SvPos& q : sv.rreads) np2::sv_update_ref_d(svw, sv.rreads_w, q, s);
                    }
                } else {
                    np2::sv_update_ref_d(svw, sv.rreads_w, rp, s);
                }
            }
            if (!rege) continue;
            in.recs.add(r->pos, cigar, r->n_cigar, r->seq(), ((size_t)r->l_qseq + 1) / 2, rdp1.s);
            meta.push_back(RecMeta{(uint32_t)l_qseq, rdp1.e - rdp1.s, g, false});
            meta.back().want_gap = sv.brk_g && g.score && g.gap.s >= (uint32_t)s && g.gap.e <= (uint32_t)e;
        }
        if (it.failed()) np2_die(err.c_str(), ref->n);
        in.recs.seq.resize(in.recs.seq.size() + 8, 0);
        lap("decode + merge");
        if (timing) {
            double ms[4];
            uint64_t nw[2];
            np::bgzf_prof_take(ms, nw);
            fprintf(stderr, "[np2 host]   of which BGZF windows: read %.2f ms, block scan %.2f ms, device inflate %.2f ms (%llu windows), host inflate %.2f ms (%llu windows)\n", ms[0],
                    ms[1], ms[2], (unsigned long long)nw[0], ms[3], (unsigned long long)nw[1]);
        }
        // ---- spans of every candidate, then the order-dependent keep rules (ctg_cns.c:3540-3545)
        std::vector<np2::SpanOut> spans;
        if (!cfg->exec->compute_spans(in, 0, &spans, &err)) np2_die(err.c_str(), ref->n);
        for (const np2::SpanOut& a : spans)
            if (a.bad) { fprintf(stderr, "bamaln error, %s\n", ref->n); exit(1); }   // ctg_cns.c:3534-3537
        lap("spans");
        {
            // coverage of a column = number of kept streams whose [aln_t_s, aln_t_e) covers it (every draft position of
            // a stream has exactly one non-insertion column), so the caps are decided from the spans alone
            std::vector<uint32_t> cand;
            for (uint32_t i = 0; i < spans.size(); ++i) {
                const np2::SpanOut& a = spans[i];
                if (a.aln_t_s > a.aln_t_e - 500u) continue;   // unsigned, as in the reference
                const uint32_t ts = a.aln_t_s - (uint32_t)s, te = a.aln_t_e - (uint32_t)s;
                if (ts > (uint32_t)l || te > (uint32_t)l) np2_die("alignment outside its window", ref->n);
                cand.push_back(i);
            }
            std::vector<int32_t> diff((size_t)l + 2, 0);
            diff[0] += 1;
            diff[(size_t)l] -= 1;   // seed
            for (uint32_t i : cand) { ++diff[spans[i].aln_t_s - (uint32_t)s]; --diff[spans[i].aln_t_e - (uint32_t)s]; }
            int32_t run = 0, mx = 0;
            for (int32_t q = 0; q <= l; ++q) { run += diff[(size_t)q]; mx = std::max(mx, run); }
            std::vector<uint32_t> cov;
            if (mx > 500) {   // deep pileup: replay the reference's decisions on a running coverage track
                cov.assign((size_t)l + 1, 0);
                for (int32_t q = 0; q < l; ++q) cov[(size_t)q] = 1;
            }
            for (uint32_t i : cand) {
                const uint32_t ts = spans[i].aln_t_s - (uint32_t)s, te = spans[i].aln_t_e - (uint32_t)s;
                if (mx > 500) {
                    if ((cov[ts] > 3000 && cov[te] > 3000) ||
                        (cov[ts] > 500 && cov[te] > 500 && (double)meta[i].aligned_q < meta[i].l_qseq * 0.9)) continue;
                    for (uint32_t q = ts; q < te; ++q) ++cov[q];
                }
                np2::StreamRef sr;
                sr.set = 0;
                sr.rec = i;
                sr.span = spans[i];
                in.streams.push_back(sr);
                if (meta[i].want_gap) {   // update_gap_info (ctg_cns.c:2627-2652)
                    np2::SvGapRead gr;
                    gr.gap = np2::SvPos{meta[i].g.gap.s, meta[i].g.gap.e};
                    gr.p_id = (uint32_t)in.streams.size();   // stream index with the seed at 0
                    gr.p_s = spans[i].aln_q_s;
                    gr.s_id = meta[i].g.fs;
                    gr.s_s = meta[i].g.ds;
                    gr.l = 0;
                    const uint8_t* sq = in.recs.seq.data() + in.recs.seq_off[i];
                    const size_t nb = (i + 1 < in.recs.seq_off.size() ? in.recs.seq_off[i + 1] : in.recs.seq.size() - 8) - in.recs.seq_off[i];
                    gr.dseq.assign(sq, sq + nb);
                    gr.dseq.resize(nb + 8, 0);
                    svw.gaps.push_back(std::move(gr));
                }
            }
        }
        uint32_t seq_count = 1 + (uint32_t)in.streams.size();
        if (seq_count < 150 || sv.rreads.size() < 150 || svw.sup_alns.empty()) sv.brk_g = 0;
        // ---- structural layer: depth track, low-depth regions, gap clusters, supplementary streams (ctg_cns.c:3559-3580)
        if (sv.brk_g) {
            if (!sv.rreads_w) {
                sv.rreads_w = np2::sv_cal_rreads_w(sv.rreads);
                for (const np2::SvPos& q : sv.rreads) np2::sv_update_ref_d(svw, sv.rreads_w, q, s);
            }
            svw.finish_depth();
            if (!sv.ref_d) sv.ref_d = np2::sv_cal_ref_d(svw.ref_ds, l / 10);
            np2::sv_update_ld_regs(&svw.ld_regs, svw.ref_ds, l / 10, sv.rreads_w, sv.ref_d);
            FILE* lg = getenv("NP2_SV_LOG") ? fopen(getenv("NP2_SV_LOG"), "a") : nullptr;   // same lines as tests/shim/np2_ref_shim.c
            if (lg) {
                fprintf(lg, "update_ld_regs l %d w %d d %d s %d -> %zu regions\n", l / 10, sv.rreads_w, sv.ref_d, s, svw.ld_regs.size());
                for (size_t i = 0; i < svw.ld_regs.size(); ++i) fprintf(lg, "  ld %zu %u %u\n", i, svw.ld_regs[i].s, svw.ld_regs[i].e);
            }
            if (sv.ref_ide) {
                const int32_t d_t = (int32_t)(sv.ref_d * 0.3);
                const uint32_t ide_t = (uint32_t)(sv.ref_ide * cfg->ide_t);
                np2::sv_update_ld_regs_with_refqv(&svw.ld_regs, svw.ref_ds, ref, sv.rreads_w * 20, s, e, d_t, ide_t, cfg->ort_t, cfg->irt_t);
                if (lg) {
                    fprintf(lg, "update_ld_regs_with_refqv w %d d_t %d ide_t %u ort_t %u irt_t %u -> %zu regions\n", sv.rreads_w * 20, d_t, ide_t, cfg->ort_t,
                            cfg->irt_t, svw.ld_regs.size());
                    for (size_t i = 0; i < svw.ld_regs.size(); ++i) fprintf(lg, "  ld %zu %u %u\n", i, svw.ld_regs[i].s, svw.ld_regs[i].e);
                }
            }
            const uint32_t n_gaps_before = (uint32_t)svw.gaps.size();
            const int cl_total = np2::sv_update_gap_cluster(&svw, sv.rreads_w, sv.ref_d, s);
            if (lg) {
                fprintf(lg, "update_gap_cluster gaps %u w %d d %d ref_s %d -> clusters %zu total %d\n", n_gaps_before, sv.rreads_w, sv.ref_d, s, svw.clusters.size(), cl_total);
                for (size_t i = 0; i < svw.clusters.size(); ++i) fprintf(lg, "  cluster %zu i_m %u median %u\n", i, svw.clusters[i].i_m, svw.clusters[i].median);
                fclose(lg);
            }
            // the supplementary alignment of every split read: bases of the primary, CIGAR of the supplementary record
            for (const np2::SvGapRead& gr : svw.gaps) {
                const np2::SvSupAln* sa_ = nullptr;
                for (const np2::SvSupAln& x : svw.sup_alns)
                    if (x.fs == gr.s_id && x.ds == gr.s_s) { sa_ = &x; break; }
                assert(sa_ != nullptr);   // find_sup_alns (ctg_cns.c:2828-2835)
                in.sup.add((int32_t)sa_->fs, sa_->cigar.data(), (uint32_t)sa_->cigar.size(), gr.dseq.data(), gr.dseq.size(), sa_->ds);
            }
            std::vector<np2::SpanOut> sup_span;
            if (!cfg->exec->compute_spans(in, 1, &sup_span, &err)) np2_die(err.c_str(), ref->n);
            const uint32_t sc0 = seq_count;
            seq_count = np2::sv_update_align_tags(&svw, sup_span, seq_count, s, &in.streams);
            if (getenv("NP2_SV_LOG")) { FILE* lg2 = fopen(getenv("NP2_SV_LOG"), "a"); if (lg2) { fprintf(lg2, "update_align_tags streams %u -> %u\n", sc0, seq_count); fclose(lg2); } }
        }
        lap("keep rules + structural");
        in.want_tags = false;           // (round 5: generate_gapseqs asks the executor for the read coordinates it needs instead of walking the streams here)
        in.lq_ratio1 = reads_type == np2k::READS_HIFI ? 0.f : gap_min_ratio1;   // the executor marks where the low-quality scans have to look (NP2_LQ_TRIGGERS=0: off)
        if (!cfg->exec->run_window(in, &out, &err)) np2_die(err.c_str(), ref->n);
        lap("window (executor)");
        std::vector<np2::LqCluster> clusters;
        if (sv.brk_g) {
            if (!np2::sv_generate_gapseqs(&svw, out, s, cfg->exec, &err)) np2_die(err.c_str(), ref->n);
            FILE* lg = getenv("NP2_SV_LOG") ? fopen(getenv("NP2_SV_LOG"), "a") : nullptr;
            if (lg) {
                for (size_t i = 0; i < svw.clusters.size(); ++i) {
                    const np2::SvCluster& c = svw.clusters[i];
                    uint32_t l2 = 0;
                    for (uint32_t j = 0; j < c.i_m; ++j) l2 += svw.gaps[c.gap[j]].l == 2;
                    fprintf(lg, "generate_gapseqs cluster %zu r %u %u i_m %u usable %u\n", i, c.r.s, c.r.e, c.i_m, l2);
                    for (uint32_t j = 0; j < c.i_m; ++j) {
                        const np2::SvGapRead& g = svw.gaps[c.gap[j]];
                        fprintf(lg, "    gap %u l %u read %u..%u p_id %u s_id %u\n", j, g.l, g.gap.s, g.gap.e, g.p_id, g.s_id);
                    }
                }
            }
            if (sv.ref_d > 15) np2::sv_update_split_p(&sv.split_ps, svw, s, e - s, ref);
            if (lg) {
                if (sv.ref_d > 15) {
                    fprintf(lg, "update_split_p -> %zu split points\n", sv.split_ps.size());
                    for (size_t i = 0; i < sv.split_ps.size(); ++i) fprintf(lg, "  split %zu %u %u\n", i, sv.split_ps[i].s, sv.split_ps[i].e);
                }
                fclose(lg);
            }
            clusters = np2::sv_lq_clusters(svw);
        }
        // ---- low-quality regions
        std::vector<np2::LqRegionIn> regs;
        if (reads_type == np2k::READS_HIFI) {
            regs = hifi_regions(&out.cons, out.stat, clusters);   // also applies the HiFi case rule (qv > 80)
        } else {
            LqCtx lx{&out.stat, &out.cons, reads_type, gap_min_ratio1, &clusters, out.trig_del.empty() ? nullptr : &out.trig_del, out.trig_ins.empty() ? nullptr : &out.trig_ins};
            for (const LqReg& q : lq_regions(lx)) regs.push_back(np2::LqRegionIn{q.start, q.end, q.l});
        }
        lap("lq regions");
        if (!regs.empty() || reads_type == np2k::READS_HIFI) {
            if (!np2::lq_stage(cfg->exec, gap_min_len, reads_type == np2k::READS_HIFI, regs, clusters, out, &out.cons, &err)) np2_die(err.c_str(), ref->n);
            if (timing) fprintf(stderr, "[np2 host] %zu low-quality regions\n", regs.size());
            lap("lq stage");
        }
        WindowCons wc;
        wc.b.swap(out.cons);   // (update_consensus_trimed with no regions would copy, ctg_cns.c:1165-1211)
        wc.uncorrected_len = s;
        windows.push_back(std::move(wc));
        s = e - cfg->s;
    }
    const double fra = (double)fra_map / (double)(total_map + 1);
    if (reads_type == np2k::READS_HIFI && fra > 0.1)   // ctg_cns.c:3593-3597
        fprintf(stderr, "Warning, Too many (%.3f%%) fragment mappings in %s, please polish the genome with other reads first, or"
                " adjust the mapping parameters to tolerate more errors, such as use asm20/map-pb instead of asm5 for minimap2,"
                " continue anyway...\n", (double)fra_map * 100 / (double)(total_map + 1), ref->n);
    consensus_trimed_data* result = link_windows(windows, sv.split_ps, (int)ref->length, 50, cfg->split, cfg->s);
    if (!windows.empty()) out.cons.swap(windows.back().b);   // keep the buffer (its capacity) for the next contig's window
    return result;
}

// A C++ exception must not leave through the C boundary (the caller is ctypes): it ends the worker like every other failure here does.
extern "C" consensus_trimed_data* ctg_cns_core(ctg_cns_cfg* cfg, ref_* ref, char* bam_list) {
    try {
        return ctg_cns_core_task(cfg, ref, bam_list);
    } catch (const std::exception& e) {
        np2_die(e.what(), ref && ref->n ? ref->n : "?");
    } catch (...) {
        np2_die("unknown exception", ref && ref->n ? ref->n : "?");
    }
    return nullptr;
}
