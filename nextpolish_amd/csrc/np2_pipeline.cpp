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
// Not built yet (fails loudly, never silently): the structural gap-cluster layer (B15).  See DESIGN.md section "path B".
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
#include "np_bam.h"

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
uint32_t cigarstr_clip(const char* s, int end) {   // cigarstr2ul
    if (end) {
        int index = 0;
        while (*(s + 1) != '\0') {
            if (*s >= '0' && *s <= '9') ++index; else index = 0;
            ++s;
        }
        s -= index;
    }
    uint32_t result = 0;
    while (*s >= '0' && *s <= '9') { result = result * 10 + (uint32_t)(*s - '0'); ++s; }
    if (*s != 'H' && *s != 'S') result = 0;
    return result;
}
int32_t cigarstr_rlen(const char* s) {   // cigarstr2rlen
    uint32_t rlen = 0, clen = 0;
    while (*s != '\0') {
        if (*s >= '0' && *s <= '9') clen = clen * 10 + (uint32_t)(*s - '0');
        else { if (*s == 'M' || *s == 'D') rlen += clen; clen = 0; }
        ++s;
    }
    return (int32_t)rlen;
}
inline uint32_t mabs(uint32_t x, uint32_t y) { return x > y ? x - y : y - x; }
void check_indel(Gap* g, int32_t rlen, const Pos* rfp1, const Pos* rdp1, const Pos* rfp2, const Pos* rdp2) {   // ctg_cns.c:2463-2492
    int l = 0;
    const int32_t mclen = (int32_t)(rlen * 0.1);
    if (rfp1->s > rfp2->s) {
        l = 1;
        const Pos* t = rfp1; rfp1 = rfp2; rfp2 = t;
        t = rdp1; rdp1 = rdp2; rdp2 = t;
    }
    if (rfp2->e > rfp1->e && rdp2->e > rdp1->e && (int64_t)rdp1->s < mclen && (int64_t)rdp2->e > (int64_t)rlen - mclen &&
        mabs(rfp2->s, rfp1->e) < 30000 && mabs(rdp2->s, rdp1->e) < 30000 && rfp1->s != rfp2->s) {
        const uint32_t score = rdp1->s + (uint32_t)rlen - rdp2->e + mabs(rfp2->s, rfp1->e) + mabs(rdp2->s, rdp1->e);
        if (score < g->score || !g->score) {
            g->score = score;
            g->ds = l ? rdp1->s : rdp2->s;
            g->fs = l ? rfp1->s : rfp2->s;
            if (rfp1->e < rfp2->s) { g->gap.s = rfp1->e; g->gap.e = rfp2->s; }
            else { g->gap.s = rfp2->s; g->gap.e = rfp1->e; }
        }
    }
}

// SA:Z value of a record (SAMv1 4.2.4 aux layout), nullptr when absent
const char* find_sa(const np::BamRec& r) {
    const uint8_t* p = r.qual() + r.l_qseq;
    const uint8_t* end = r.data.data() + r.data.size();
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
};

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
    for (int i = 1; i < len; ++i) {
        if (st(i).l_del < st(i).coverage * 0.3 && cb[(size_t)i].pos < cb[(size_t)i - 1].pos + 20) continue;
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

// regions in DEscending order of position, like the reference builds them (no gap clusters: the structural layer
// that provides them is not built)
std::vector<LqReg> lq_regions(const LqCtx& x) {
    const std::vector<ColStat>& msa = *x.st;
    const std::vector<ConsBase>& cb = *x.c;
    const int len = (int)cb.size();
    std::vector<LqReg> lq(200);
    lq[0].start = lq[0].end = 0;
    int index = 0;
    std::vector<Del> dels = l_del_regions(x);
    int dels_i = (int)dels.size();
    auto st = [&](int i) -> const ColStat& { return msa[(size_t)cb[(size_t)i].pos]; };
    for (int i = len - 1; i >= 0; --i) {
        if ((float)st(i).l_ins < (float)st(i).coverage * x.gap_min_ratio1) continue;
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
std::vector<np2::LqRegionIn> hifi_regions(std::vector<ConsBase>* cons, const std::vector<ColStat>& st) {
    std::vector<np2::LqRegionIn> regs;
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
            if (!regs.empty() && R(lq_s).pos >= regs.back().start) {
                regs.back().start = R(lq_e).pos;
            } else {
                regs.push_back(np2::LqRegionIn{R(lq_e).pos, R(lq_s).pos, 4});
            }
            lq = 0;
            lq_s = -1;
        }
        const char up = (char)toupper(R(p).base);
        R(p).base = (cov > 4 && qv > 80) ? up : (char)tolower(up);
    }
    return regs;
}

struct WindowCons {
    std::vector<ConsBase> b;
    uint32_t lstrip = 0, rstrip = 0;
    int32_t uncorrected_len = 0;
};

// link_consensus (ctg_cns.c:3121-3223) without split points (the structural layer that produces them is not built)
consensus_trimed_data* link_windows(std::vector<WindowCons>& w, int len, int k, int split, int overlap_s) {
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
    out->i_m = 1;   // split ? split_ps->i + 1 : 1 with no split points
    out->data = (consensus_trimed*)calloc((size_t)out->i_m, sizeof(consensus_trimed));
    (void)split;
    (void)len;
    size_t total = 0;
    for (auto& c : w) total += c.b.size();
    consensus_trimed* ct = &out->data[0];
    ct->seq = (char*)malloc(total + 1);
    for (auto& c : w)
        for (size_t j = c.lstrip; j + c.rstrip < c.b.size(); ++j) ct->seq[ct->len++] = c.b[j].base;
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
    if (cfg->exec && cfg->exec_pid == (int)getpid()) delete cfg->exec;
    free(cfg);
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

extern "C" consensus_trimed_data* ctg_cns_core(ctg_cns_cfg* cfg, ref_* ref, char* bam_list) {
    g_err.clear();
    if (!cfg->exec || cfg->exec_pid != (int)getpid()) {   // lazily, per process: the caller forks after ctg_cns_init
        std::string err;
        cfg->exec = np2::make_exec(&err);
        cfg->exec_pid = (int)getpid();
        if (!cfg->exec) { fprintf(stderr, "nextpolish2 (MI355X): %s\n", err.c_str()); exit(1); }
    }
    const int reads_type = cfg->reads_type;
    const uint32_t gap_min_len = reads_type != np2k::READS_ONT ? 5 : 3;
    const float gap_min_ratio1 = reads_type != np2k::READS_ONT ? 0.3f : 0.01f;
    const float max_clip_ratio = reads_type == np2k::READS_HIFI ? 0.1f : 0.7f;

    assert(ref->length < INT_MAX);
    std::vector<char> rfseq((size_t)ref->length + 1);
    bit2seq1(ref->s, ref->length, rfseq.data());
    const int32_t b = cal_win_len(cfg->w, cfg->s, ref->length);
    int brk_g = ref->length > 100000 ? 1 : 0;
    int32_t s = 0, e = 0;
    int rreads_i = 0;
    std::vector<WindowCons> windows;
    long fra_map = 0, total_map = 0;
    np2::WindowInput in;
    np2::WindowOutput out;
    static uint64_t contig_serial = 0;
    in.contig_serial = ++contig_serial;
    while (e < (int32_t)ref->length) {
        e = s + b > (int32_t)ref->length ? (int32_t)ref->length : s + b;
        in.contig_seq = rfseq.data();
        in.s = s;
        in.e = e;
        in.gap_min_len = gap_min_len;
        in.read_type = reads_type;
        in.pos.clear(); in.n_cigar.clear(); in.l_qseq.clear(); in.aligned_q.clear(); in.cigar_off.clear(); in.seq_off.clear();
        in.cigar.clear(); in.seq.clear();
        uint32_t sup_aln_i = 0;
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
                        const Pos rfp2{sp, sp + (uint32_t)cigarstr_rlen(f[3].c_str())};
                        const Pos rdp2{cigarstr_clip(f[3].c_str(), 0), (uint32_t)l_qseq - cigarstr_clip(f[3].c_str(), 1)};
                        check_indel(&g, l_qseq, &rfp1, &rdp1, &rfp2, &rdp2);
                    }
                }
            }
            if (rege && (r->flag & 0xD04) && brk_g && g.score) ++sup_aln_i;
            if (r->flag & 0xD04) continue;
            ++total_map;
            const double frac = (double)(rdp1.e - rdp1.s) / (double)l_qseq;
            if (frac < 0.7) ++fra_map;
            if (!g.score && frac <= max_clip_ratio) continue;
            if (brk_g && rreads_i < 50000) ++rreads_i;
            if (!rege) continue;
            in.pos.push_back(r->pos);
            in.n_cigar.push_back(r->n_cigar);
            in.l_qseq.push_back((uint32_t)l_qseq);
            in.aligned_q.push_back(rdp1.e - rdp1.s);
            in.cigar_off.push_back(in.cigar.size());
            in.seq_off.push_back(in.seq.size());
            in.cigar.insert(in.cigar.end(), cigar, cigar + r->n_cigar);
            const uint8_t* sq = r->seq();
            in.seq.insert(in.seq.end(), sq, sq + ((size_t)r->l_qseq + 1) / 2);
        }
        if (it.failed()) np2_die(err.c_str(), ref->n);
        in.seq.resize(in.seq.size() + 8, 0);
        if (!cfg->exec->run_window(in, &out, &err)) np2_die(err.c_str(), ref->n);
        if (out.bad_cigar) { fprintf(stderr, "bamaln error, %s\n", ref->n); exit(1); }   // ctg_cns.c:3534-3537
        if (out.seq_count < 150 || rreads_i < 150 || sup_aln_i == 0) brk_g = 0;
        if (brk_g) np2_die("split-read structural layer (gap clusters, ctg_cns.c:3559-3580) is not built yet", ref->n);
        // ---- low-quality regions: their re-consensus is not built yet
        std::vector<np2::LqRegionIn> regs;
        if (reads_type == np2k::READS_HIFI) {
            regs = hifi_regions(&out.cons, out.stat);   // also applies the HiFi case rule (qv > 80)
        } else {
            LqCtx lx{&out.stat, &out.cons, reads_type, gap_min_ratio1};
            for (const LqReg& r : lq_regions(lx)) regs.push_back(np2::LqRegionIn{r.start, r.end, r.l});
        }
        if (!regs.empty() || reads_type == np2k::READS_HIFI) {
            timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            if (!np2::lq_stage(cfg->exec, gap_min_len, reads_type == np2k::READS_HIFI, regs, out, &out.cons, &err)) np2_die(err.c_str(), ref->n);
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (getenv("NP2_TIMING")) fprintf(stderr, "[np2 lq stage] %zu regions, %.2f ms\n", regs.size(), (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
        }
        WindowCons wc;
        wc.b = out.cons;   // update_consensus_trimed with no regions: a copy (ctg_cns.c:1165-1211)
        wc.uncorrected_len = s;
        windows.push_back(std::move(wc));
        s = e - cfg->s;
    }
    const double fra = (double)fra_map / (double)(total_map + 1);
    if (reads_type == np2k::READS_HIFI && fra > 0.1)   // ctg_cns.c:3593-3597
        fprintf(stderr, "Warning, Too many (%.3f%%) fragment mappings in %s, please polish the genome with other reads first, or"
                " adjust the mapping parameters to tolerate more errors, such as use asm20/map-pb instead of asm5 for minimap2,"
                " continue anyway...\n", (double)fra_map * 100 / (double)(total_map + 1), ref->n);
    return link_windows(windows, (int)ref->length, 50, cfg->split, cfg->s);
}
