// What the reference's region iterator hands to kmer_count / snp_valid, re-derived on the host from the BAM index and the records'
// virtual offsets (reference: source/lib/contig.c:982-1043 contig_update_iter / contig_next_iter over htslib 1.9
// hts.c:2086-2189 hts_itr_query and hts.c:2614-2655 hts_itr_next).
//
// Why: ss_kmer_correct (kmercount.c:175-261) walks the parts of a contig with ONE iterator per loop.  A part whose end lies before
// the record behind the first chunk of the current chunk list re-uses that list -- made for another region -- and resumes at a saved
// offset; the scan can then run out of chunks before it meets a record that starts behind the part, and the record left in the
// buffer (which the level-1 fallback of kmercount.c:212-217 parses) is not "the next record in file order".  On drafts covered a
// few times this changes a few bases per contig.  Everything here is a function of record positions, end
// positions, virtual offsets and the index -- not of the votes -- except which parts run the second loop.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "np_bam.h"

namespace np1replay {

struct Span { uint64_t beg, end; };   // one chunk of a chunk list: [beg, end) in virtual offsets

// One reference sequence of a BAI as htslib holds it after loading: zeros of the linear index filled from the left, every bin with
// the linear offset of its first 16 kb window (`loff`).
class RefIndex {
public:
    explicit RefIndex(const np::BaiRef& ref) : ref_(ref), lin_(ref.linear) {
        for (size_t j = 1; j < lin_.size(); ++j)
            if (lin_[j] == 0) lin_[j] = lin_[j - 1];
    }
    bool empty() const { return ref_.bins.empty(); }
    // chunk list of a query [beg, end): min / max offset, bins of the six levels, sort, containment, overlap, same-block merge
    std::vector<Span> query(int32_t beg, int32_t end) const {
        std::vector<Span> off;
        if (beg < 0) beg = 0;
        if (empty() || end < beg) return off;
        // smallest offset a record overlapping beg's window can have: loff of the lowest existing bin at or left of it, going up
        uint64_t min_off = 0;
        {
            int bin = kFirst5 + (beg >> 14);
            const std::vector<np::BaiChunk>* hit = nullptr;
            uint32_t hit_bin = 0;
            do {
                auto it = ref_.bins.find((uint32_t)bin);
                if (it != ref_.bins.end()) { hit = &it->second; hit_bin = (uint32_t)bin; break; }
                const int first = (parent(bin) << 3) + 1;
                if (bin > first) --bin; else bin = parent(bin);
            } while (bin);
            if (!hit && bin == 0) {
                auto it = ref_.bins.find(0u);
                if (it != ref_.bins.end()) { hit = &it->second; hit_bin = 0; }
            }
            if (hit) min_off = loff(hit_bin);
        }
        // largest offset worth reading: first chunk of the next existing bin to the right of end, going up
        uint64_t max_off = ~0ull;
        {
            int bin = kFirst5 + ((end - 1) >> 14) + 1;
            if (bin >= kBins) bin = 0;
            for (;;) {
                while (bin % 8 == 1) bin = parent(bin);
                if (bin == 0) break;
                auto it = ref_.bins.find((uint32_t)bin);
                if (it != ref_.bins.end() && !it->second.empty()) { max_off = it->second[0].beg; break; }
                ++bin;
            }
        }
        {
            int64_t e = end;
            int s = 14 + 15;
            if (beg < e) {
                if (e >= (1LL << s)) e = 1LL << s;
                --e;
                for (int l = 0, t = 0; l <= 5; s -= 3, t += 1 << (3 * l), ++l) {
                    const int b = t + (int)(beg >> s), ee = t + (int)(e >> s);
                    for (int bb = b; bb <= ee; ++bb) {
                        auto it = ref_.bins.find((uint32_t)bb);
                        if (it == ref_.bins.end()) continue;
                        for (const np::BaiChunk& c : it->second)
                            if (c.end > min_off && c.beg < max_off) off.push_back(Span{c.beg, c.end});
                    }
                }
            }
        }
        if (off.empty()) return off;
        std::sort(off.begin(), off.end(), [](const Span& a, const Span& b) { return a.beg < b.beg; });
        size_t l = 0;
        for (size_t i = 1; i < off.size(); ++i)
            if (off[l].end < off[i].end) off[++l] = off[i];
        off.resize(l + 1);
        for (size_t i = 1; i < off.size(); ++i)
            if (off[i - 1].end >= off[i].beg) off[i - 1].end = off[i].beg;
        l = 0;
        for (size_t i = 1; i < off.size(); ++i) {
            if (off[l].end >> 16 == off[i].beg >> 16) off[l].end = off[i].end;
            else off[++l] = off[i];
        }
        off.resize(l + 1);
        return off;
    }

private:
    static constexpr int kFirst5 = 4681, kBins = 37449;
    static int parent(int b) { return (b - 1) >> 3; }
    uint64_t loff(uint32_t bin) const {
        if (bin >= (uint32_t)kBins) return 0;
        int lvl = 0;
        for (uint32_t t = bin; t; t = (t - 1) >> 3) ++lvl;
        const uint32_t first = ((1u << (3 * lvl)) - 1) / 7;
        const uint64_t bot = (uint64_t)(bin - first) << ((5 - lvl) * 3);
        return bot < lin_.size() ? lin_[bot] : 0;
    }
    const np::BaiRef& ref_;
    std::vector<uint64_t> lin_;
};

// the records of one contig in file order, as the reader of the BAM meets them
struct Records {
    const uint64_t* voff;       // first byte of each record
    const uint64_t* voff_end;   // byte behind it
    const int32_t* pos;
    const int32_t* endpos;
    int64_t n;
    bool followed;              // records of another contig follow in the file
    int32_t length;             // contig length
};

// One iterator of ss_kmer_correct together with the variables contig_next_iter threads through it.
class Scanner {
public:
    Scanner(const RefIndex& ix, const Records& rec) : ix_(ix), r_(rec) {}
    // first call of a part (contig_next_iter with flag 1: swapped interval = records with pos < start and endpos > end + 1)
    void begin(int32_t start, int32_t end, int32_t next_part_end) {
        nextposend_ = next_part_end;
        if (have_) {
            if (end < iterend_) { beg_ = start; end_ = end + 1; finished_ = false; }
            else have_ = false;
        }
        if (!have_) {
            off_ = ix_.query(start, end + 1);
            have_ = true;
            beg_ = start; end_ = end + 1;
            i_ = -1;
            curr_off_ = 0;
            finished_ = off_.empty();
            if (!off_.empty()) {
                const int64_t k = at(off_[0].end);
                iterend_ = k >= 0 ? r_.pos[k] : r_.length;
            }
        }
        if (curr_off_) {
            curr_off_ = saved_off_;
            curr_end_ = saved_end_;
            fpos_ = curr_off_;
            if (curr_off_ == 0) i_ = -1;
        } else {
            saved_off_ = 0;
        }
        std::swap(beg_, end_);
    }
    // next record the loop gets, or -1
    int64_t next() {
        const int64_t k = off_.empty() ? -1 : advance();
        if (k >= 0 && curr_end_ <= nextposend_) { saved_off_ = curr_off_; saved_end_ = curr_end_; }
        else nextposend_ = -1;
        return k;
    }
    // record the reader read last (what is in the bam1_t): -1 nothing yet, -2 a record of another contig
    int64_t buffer() const { return buffer_; }

private:
    int64_t at(uint64_t v) const {   // record starting at (or first behind) a virtual offset; -1 behind the contig's records
        const uint64_t* p = std::lower_bound(r_.voff, r_.voff + r_.n, v);
        return p < r_.voff + r_.n ? (int64_t)(p - r_.voff) : -1;
    }
    int64_t advance() {
        if (finished_) return -1;
        for (;;) {
            if (curr_off_ == 0 || (i_ >= 0 && curr_off_ >= off_[(size_t)i_].end)) {
                if (i_ == (int)off_.size() - 1) break;
                if (i_ < 0 || off_[(size_t)i_].end != off_[(size_t)i_ + 1].beg) { fpos_ = off_[(size_t)i_ + 1].beg; curr_off_ = fpos_; }
                ++i_;
            }
            const int64_t k = at(fpos_);
            if (k < 0) {
                if (r_.followed) buffer_ = -2;
                break;
            }
            fpos_ = r_.voff_end[k];
            curr_off_ = fpos_;
            buffer_ = k;
            if (r_.pos[k] >= end_) break;
            if (r_.endpos[k] > beg_ && end_ > r_.pos[k]) { curr_end_ = r_.endpos[k]; return k; }
        }
        finished_ = true;
        return -1;
    }
    const RefIndex& ix_;
    const Records r_;
    std::vector<Span> off_;
    bool have_ = false, finished_ = true;
    int i_ = -1;
    uint64_t curr_off_ = 0, saved_off_ = 0, fpos_ = 0;
    int32_t beg_ = 0, end_ = 0, curr_end_ = 0, saved_end_ = 0, iterend_ = 0, nextposend_ = -1;
    int64_t buffer_ = -1;
};

// First loop of every part of one contig (kmercount.c:196-207): per part the records the loop gets, in order, and the record left in
// the buffer.  part_se: (start, end) pairs in list order; next_end[p] = the end the reference passes as nextposend.
// limit (optional): limit[p] > 0 = the loop of part p leaves through the max_count_kmer break (kmercount.c:201-203) after that many
// records -- the iterator then keeps the chunk position, the saved offset and the buffered record of that moment, and the next part
// that re-uses it resumes from there (which records count towards the break is decided by the votes: the device reports it,
// np1_device.hip:replay_votes).  skip[p] != 0: the pair is not iterated at all.
struct FirstLoop {
    std::vector<uint32_t> first;   // n_parts + 1 offsets into list
    std::vector<uint32_t> list;    // local record indices
    std::vector<int64_t> stale;    // per part: local record index, -1 none, -2 another contig's record
};
inline FirstLoop first_loop(const RefIndex& ix, const Records& rec, const int32_t* part_se, const int32_t* next_end, uint32_t n_parts,
                            const uint32_t* limit = nullptr, const uint8_t* skip = nullptr) {
    FirstLoop o;
    Scanner sc(ix, rec);
    o.first.push_back(0);
    for (uint32_t p = 0; p < n_parts; ++p) {
        if (!(skip && skip[p])) {
            const uint32_t lim = limit ? limit[p] : 0;
            uint32_t got = 0;
            sc.begin(part_se[2 * p], part_se[2 * p + 1], next_end[p]);
            for (int64_t k; (k = sc.next()) >= 0;) {
                o.list.push_back((uint32_t)k);
                if (++got == lim) break;
            }
        }
        o.first.push_back((uint32_t)o.list.size());
        o.stale.push_back(sc.buffer());
    }
    return o;
}
// Second loop (kmercount.c:209-218) over the parts whose first loop left no candidate (`empty`): how many passes it makes
inline std::vector<uint32_t> second_loop_passes(const RefIndex& ix, const Records& rec, const int32_t* part_se, const int32_t* next_end, uint32_t n_parts,
                                               const uint8_t* empty) {
    std::vector<uint32_t> n2(n_parts, 0);
    Scanner sc(ix, rec);
    for (uint32_t p = 0; p < n_parts; ++p) {
        if (!empty[p]) continue;    // (a skipped pair is never `empty`)
        sc.begin(part_se[2 * p], part_se[2 * p + 1], next_end[p]);
        while (sc.next() >= 0) ++n2[p];
    }
    return n2;
}

}  // namespace np1replay
