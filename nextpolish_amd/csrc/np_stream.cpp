// BAM + FASTA -> decoded record stream.  See np_stream.h.
#include "np_stream.h"

#include <cstdlib>

#include <cstring>

#include "np_bam.h"

namespace np {

void ReadStream::clear() { *this = ReadStream(); }

uint64_t ReadStream::algorithmic_input_bytes(bool with_qual) const {
    uint64_t b = draft.size();
    for (size_t i = 0; i < pos.size(); ++i) {
        b += 32 + 4ull * n_cigar[i] + ((uint64_t)l_qseq[i] + 1) / 2;
        if (with_qual) b += (uint64_t)l_qseq[i];
    }
    return b;
}

static void append_record(const BamRec& r, uint32_t c, bool with_qual, ReadStream* s, voff_t v0 = 0, voff_t v1 = 0) {
    s->voff.push_back(v0);
    s->voff_end.push_back(v1);
    s->pos.push_back(r.pos);
    s->ctg.push_back(c);
    s->flag.push_back(r.flag);
    s->n_cigar.push_back(r.n_cigar);
    s->l_qseq.push_back(r.l_qseq);
    s->mapq.push_back(r.mapq);
    s->isize.push_back(r.isize);
    s->cigar_off.push_back(s->cigar.size());
    s->seq_off.push_back(s->seq.size());
    s->cigar.insert(s->cigar.end(), r.cigar(), r.cigar() + r.n_cigar);
    size_t sb = ((size_t)r.l_qseq + 1) / 2;
    s->seq.insert(s->seq.end(), r.seq(), r.seq() + sb);
    s->qual_off.push_back(s->qual.size());
    if (with_qual) s->qual.insert(s->qual.end(), r.qual(), r.qual() + r.l_qseq);
}

bool load_stream(const std::string& fasta, const std::string& bam, const std::vector<std::string>& names,
                 bool with_qual, ReadStream* out, std::string* err) {
    out->clear();
    Fai fai;
    if (!fai.load(fasta)) { *err = "cannot load FASTA/index: " + fasta; return false; }
    BamReader rd;
    if (!rd.open(bam)) { *err = "cannot open BAM: " + bam; return false; }
    // records could be used in place inside the BGZF window (every record is consumed before the next one is read), but
    // on the GPU box's host the copying reader is ~10 % faster end to end (the copy doubles as a prefetch): opt-in only
    rd.set_zero_copy(getenv("NP_BAM_ZERO_COPY") != nullptr);
    const BamHeader& hdr = rd.header();

    std::vector<int> fai_ids;
    bool all = names.empty();
    if (all) {
        for (int i = 0; i < fai.nseq(); ++i) fai_ids.push_back(i);
    } else {
        for (const std::string& n : names) {
            int id = fai.find(n);
            if (id < 0) { *err = "contig not in FASTA index: " + n; return false; }
            fai_ids.push_back(id);
        }
    }
    size_t nc = fai_ids.size();
    out->ctg_off.push_back(0);
    std::vector<int> tid_of(nc);
    std::string seq;
    for (size_t c = 0; c < nc; ++c) {
        const FaiEntry& e = fai.entry(fai_ids[c]);
        if (!fai.fetch(fai_ids[c], &seq)) { *err = "cannot fetch contig: " + e.name; return false; }
        out->names.push_back(e.name);
        out->ctg_len.push_back((int32_t)seq.size());
        out->draft += seq;
        out->ctg_off.push_back((uint32_t)out->draft.size());
        tid_of[c] = hdr.name2id(e.name);   // -1: contig absent from the BAM header => no records
    }

    BamRec r;
    out->read_begin.assign(nc + 1, 0);
    // Is the request "contigs in ascending BAM tid order"?  Then one forward pass is enough.
    bool ascending = true;
    for (size_t c = 1; c < nc; ++c)
        if (tid_of[c] >= 0 && tid_of[c - 1] >= 0 && tid_of[c] <= tid_of[c - 1]) ascending = false;
    BaiIndex bai;
    bool have_bai = false;
    if (!all || !ascending) {
        have_bai = bai.load(bam + ".bai");
        if (!have_bai) { *err = "cannot load BAM index: " + bam + ".bai"; return false; }
    }
    bool pending = false;   // r holds a record read but not yet consumed (sequential mode)
    voff_t rec_v0 = 0, rec_v1 = 0;   // virtual offsets of r and of the byte behind it
    for (size_t c = 0; c < nc; ++c) {
        out->read_begin[c] = out->n_reads();
        int tid = tid_of[c];
        if (tid < 0) continue;
        int32_t L = out->ctg_len[c];
        if (have_bai) {
            voff_t v;
            if (!bai.region_start(tid, 0, L, &v)) continue;
            if (!rd.seek(v)) { *err = "BAM seek failed"; return false; }
            pending = false;
        }
        for (;;) {
            if (!pending) {
                rec_v0 = rd.tell();
                int st = rd.next(r);
                rec_v1 = rd.tell();
                if (st < 0) { *err = "corrupt BAM record in " + bam; return false; }
                if (st == 0) break;
            }
            pending = false;
            if (r.tid < 0 || r.tid > tid) { pending = true; break; }   // past this contig (or unplaced tail)
            if (r.tid < tid) continue;
            if (r.pos >= L) { pending = true; break; }                  // iterator stop: beg >= end
            if (r.pos < 0 || r.endpos() <= 0) continue;
            append_record(r, (uint32_t)c, with_qual, out, rec_v0, rec_v1);
        }
        if (r.tid < 0 && pending) { /* unplaced reads follow: nothing more for any contig */ }
    }
    out->read_begin[nc] = out->n_reads();
    return true;
}

bool load_stream_region(const std::string& bam, const BaiIndex& bai, const std::string& name, const std::string& draft_whole, int32_t e_lo, int32_t e_hi,
                        ReadStream* out, int32_t* lo_out, int32_t* hi_out, std::string* err) {
    out->clear();
    BamReader rd;
    if (!rd.open(bam)) { *err = "cannot open BAM: " + bam; return false; }
    const int tid = rd.header().name2id(name);
    const int32_t L = (int32_t)draft_whole.size();
    // the records come sorted by position: the first one that is kept has the smallest start, which fixes the hull's left end
    int32_t lo = e_lo > 0 ? e_lo - 1 : 0, hi = e_hi;
    bool first = true;
    out->names.push_back(name);
    out->read_begin.assign(2, 0);
    voff_t v;
    // (one base earlier than the interval: a record whose last base is e_lo - 1 still pads the columns behind it)
    if (tid >= 0 && bai.region_start(tid, e_lo > 0 ? e_lo - 1 : 0, e_hi < L ? e_hi + 1 : L, &v)) {
        if (!rd.seek(v)) { *err = "BAM seek failed"; return false; }
        BamRec r;
        for (;;) {
            const voff_t v0 = rd.tell();
            const int st = rd.next(r);
            const voff_t v1 = rd.tell();
            if (st < 0) { *err = "corrupt BAM record in " + bam; return false; }
            if (st == 0) break;
            if (r.tid >= 0 && r.tid < tid) continue;
            if (r.tid != tid) break;
            if (r.pos > e_hi || r.pos >= L) break;                                   // nothing further touches the tile
            if (r.pos < 0 || r.endpos() <= 0) continue;                              // (what the whole-contig iterator drops)
            int32_t e = r.pos;
            const uint32_t* cg = r.cigar();
            for (uint32_t i = 0; i < r.n_cigar; ++i)
                if ((cg[i] & 15u) == 0 || (cg[i] & 15u) == 2) e += (int32_t)(cg[i] >> 4);
            if (e < e_lo) continue;
            if (first) {
                first = false;
                const int32_t m = r.pos < e_lo ? r.pos : e_lo;
                lo = m > 0 ? m - 1 : 0;
            }
            if (e > hi) hi = e;
            r.pos -= lo;
            append_record(r, 0u, false, out, v0, v1);
        }
    }
    hi = hi < L ? hi + 1 : L;
    if (hi > L) hi = L;
    out->ctg_len.push_back(hi - lo);
    out->draft.assign(draft_whole, (size_t)lo, (size_t)(hi - lo));
    out->ctg_off.push_back(0);
    out->ctg_off.push_back((uint32_t)out->draft.size());
    out->read_begin[1] = out->n_reads();
    *lo_out = lo;
    *hi_out = hi;
    return true;
}

bool bam_insert_probe(const std::string& bam, uint32_t count_read_ins, uint32_t max_ins_len, uint32_t* mean_out,
                      int32_t* read_len_out) {
    BamReader rd;
    if (!rd.open(bam)) return false;
    BamRec r;
    uint32_t sum = 0, count = 1;
    *read_len_out = 0;
    while (rd.next(r) > 0 && count < count_read_ins) {
        if (r.isize > 0 && (uint32_t)r.isize < max_ins_len) {
            sum += (uint32_t)r.isize;
            if (*read_len_out == 0) *read_len_out = r.l_qseq;
            ++count;
        }
    }
    *mean_out = sum / count;
    return true;
}

}  // namespace np
