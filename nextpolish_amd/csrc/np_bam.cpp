// BAM / BAI / FAI reader + writer (SAMv1 §4.2, §5.2, §5.3).  See np_bam.h.
#include "np_bam.h"

#include <fcntl.h>
#include <unistd.h>

#include <atomic>

#include "np_threads.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>

namespace np {

int BamHeader::name2id(const std::string& n) const {
    for (size_t i = 0; i < names.size(); ++i)
        if (names[i] == n) return (int)i;
    return -1;
}

int32_t BamRec::rlen() const {
    int32_t l = 0;
    const uint32_t* c = cigar();
    for (uint32_t i = 0; i < n_cigar; ++i) {
        uint32_t op = c[i] & 0xf;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) l += (int32_t)(c[i] >> 4);
    }
    return l;
}

int32_t BamRec::endpos() const {
    // the reference's vendored htslib 1.9 (sam.c:391-397): pos + rlen for a mapped record with a CIGAR -- also when the CIGAR
    // consumes no reference base (only S/I/H/P ops), where later htslib versions return pos + 1
    if (!(flag & 4) && n_cigar > 0) return pos + rlen();
    return pos + 1;
}

bool BamReader::open(const std::string& path) {
    if (!bg_.open(path)) return false;
    char magic[4];
    if (bg_.read(magic, 4) != 4 || memcmp(magic, "BAM\1", 4) != 0) return false;
    int32_t l_text;
    if (bg_.read(&l_text, 4) != 4 || l_text < 0) return false;
    hdr_.text.resize(l_text);
    if (l_text && bg_.read(&hdr_.text[0], l_text) != l_text) return false;
    int32_t n_ref;
    if (bg_.read(&n_ref, 4) != 4 || n_ref < 0) return false;
    hdr_.names.clear();
    hdr_.lens.clear();
    for (int i = 0; i < n_ref; ++i) {
        int32_t l_name;
        if (bg_.read(&l_name, 4) != 4 || l_name <= 0) return false;
        std::string nm(l_name, '\0');
        if (bg_.read(&nm[0], l_name) != l_name) return false;
        nm.resize(strlen(nm.c_str()));
        uint32_t l_ref;
        if (bg_.read(&l_ref, 4) != 4) return false;
        hdr_.names.push_back(nm);
        hdr_.lens.push_back(l_ref);
    }
    first_rec_ = bg_.tell();
    return true;
}

int BamReader::next(BamRec& r) {
    int32_t block_len;
    int64_t got = bg_.read(&block_len, 4);
    if (got == 0) return 0;
    if (got != 4 || block_len < 32) return -1;
    uint32_t x[8];
    r.ext = nullptr;
    const uint8_t* view = zero_copy_ ? bg_.take_contiguous((size_t)block_len, 16) : nullptr;
    if (view) memcpy(x, view, 32);
    else if (bg_.read(x, 32) != 32) return -1;
    r.tid = (int32_t)x[0];
    r.pos = (int32_t)x[1];
    r.l_qname = (uint8_t)(x[2] & 0xff);
    r.mapq = (uint8_t)((x[2] >> 8) & 0xff);
    r.bin = (uint16_t)(x[2] >> 16);
    r.n_cigar = x[3] & 0xffff;
    r.flag = (uint16_t)(x[3] >> 16);
    r.l_qseq = (int32_t)x[4];
    r.mtid = (int32_t)x[5];
    r.mpos = (int32_t)x[6];
    r.isize = (int32_t)x[7];
    size_t rest = (size_t)block_len - 32;
    if (r.l_qseq < 0 || (size_t)r.l_qname + 4 * (size_t)r.n_cigar + ((size_t)r.l_qseq + 1) / 2 + (size_t)r.l_qseq > rest)
        return -1;
    if (view) {
        uint32_t c0 = 0;
        if (r.n_cigar) memcpy(&c0, view + 32 + r.l_qname, 4);
        if (!(r.n_cigar && (c0 & 0xfu) == 4 && (int64_t)(c0 >> 4) == (int64_t)r.l_qseq)) {   // (a CG placeholder takes the copying path)
            r.ext = view + 32;
            r.ext_len = rest;
            return 1;
        }
    }
    r.data.resize(rest + 8);   // small tail pad: the trim loops may peek one nibble past the sequence
    if (view) memcpy(r.data.data(), view + 32, rest);
    else if (rest && bg_.read(r.data.data(), rest) != (int64_t)rest) return -1;
    memset(r.data.data() + rest, 0, 8);
    // A CIGAR with more than 65 535 operations does not fit the 16-bit count: the record then carries the placeholder
    // "<l_qseq>S<rlen>N" and the real operations in the tag CG:B:I (SAMv1 section 4.2.2).  htslib swaps them in while
    // reading (bam_tag2cigar), so the reference never sees the placeholder; same here.
    if (r.n_cigar >= 1 && r.tid >= 0 && r.pos >= 0) {
        uint32_t c0;
        memcpy(&c0, r.data.data() + r.l_qname, 4);
        if ((c0 & 0xfu) == 4 && (int64_t)(c0 >> 4) == (int64_t)r.l_qseq) {
            const size_t aux0 = (size_t)r.l_qname + 4 * (size_t)r.n_cigar + ((size_t)r.l_qseq + 1) / 2 + (size_t)r.l_qseq;
            const uint8_t* p = r.data.data() + aux0;
            const uint8_t* end = r.data.data() + rest;
            while (p + 3 <= end) {
                const uint8_t* tag = p;
                const char ty = (char)p[2];
                p += 3;
                size_t sz = 0;
                if (ty == 'B') {
                    if (p + 5 > end) break;
                    const char sub = (char)p[0];
                    uint32_t cnt;
                    memcpy(&cnt, p + 1, 4);
                    const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    sz = 5 + w * (size_t)cnt;
                    if (tag[0] == 'C' && tag[1] == 'G' && sub == 'I' && cnt >= r.n_cigar && cnt < (1u << 29) && p + sz <= end) {
                        std::vector<uint8_t> nd;
                        nd.reserve(rest + 4 * (size_t)cnt + 8);
                        nd.insert(nd.end(), r.data.begin(), r.data.begin() + r.l_qname);                       // name
                        nd.insert(nd.end(), p + 5, p + 5 + 4 * (size_t)cnt);                                  // real CIGAR
                        nd.insert(nd.end(), r.data.begin() + r.l_qname + 4 * (size_t)r.n_cigar, r.data.begin() + (tag - r.data.data()));   // seq, qual, aux before CG
                        nd.insert(nd.end(), p + sz, end);                                                       // aux behind CG
                        nd.resize(nd.size() + 8, 0);
                        r.data.swap(nd);
                        r.n_cigar = cnt;
                        break;
                    }
                } else if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
                else if (ty == 's' || ty == 'S') sz = 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
                else if (ty == 'Z' || ty == 'H') { const uint8_t* z = p; while (z < end && *z) ++z; sz = (size_t)(z - p) + 1; }
                else break;
                p += sz;
            }
        }
    }
    return 1;
}

int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

static void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>* out) {
    out->clear();
    if (beg >= end) return;
    --end;
    out->push_back(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) out->push_back((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) out->push_back((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) out->push_back((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) out->push_back((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) out->push_back((uint32_t)k);
}

static const uint32_t kMetaBin = 37450;

bool BaiIndex::load(const std::string& path) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    auto rd = [&](void* p, size_t n) { return fread(p, 1, n, fp) == n; };
    char magic[4];
    int32_t n_ref;
    bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0;
    if (ok) {
        refs.assign(n_ref, BaiRef());
        for (int i = 0; ok && i < n_ref; ++i) {
            int32_t n_bin;
            ok = rd(&n_bin, 4);
            for (int b = 0; ok && b < n_bin; ++b) {
                uint32_t bin;
                int32_t n_chunk;
                ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0;
                if (!ok) break;
                std::vector<BaiChunk> ch(n_chunk);
                if (n_chunk) ok = rd(ch.data(), sizeof(BaiChunk) * (size_t)n_chunk);
                refs[i].bins[bin] = std::move(ch);
            }
            int32_t n_intv;
            ok = ok && rd(&n_intv, 4) && n_intv >= 0;
            if (ok) {
                refs[i].linear.resize(n_intv);
                if (n_intv) ok = rd(refs[i].linear.data(), 8 * (size_t)n_intv);
            }
        }
    }
    fclose(fp);
    return ok;
}

bool BaiIndex::region_start(int tid, int32_t beg, int32_t end, voff_t* out) const {
    if (tid < 0 || tid >= (int)refs.size()) return false;
    const BaiRef& r = refs[tid];
    if (beg < 0) beg = 0;
    voff_t min_off = 0;
    size_t w = (size_t)(beg >> 14);
    if (!r.linear.empty()) min_off = r.linear[w < r.linear.size() ? w : r.linear.size() - 1];
    std::vector<uint32_t> bins;
    reg2bins(beg, end, &bins);
    bool found = false;
    voff_t best = 0;
    for (uint32_t b : bins) {
        auto it = r.bins.find(b);
        if (it == r.bins.end()) continue;
        for (const BaiChunk& c : it->second) {
            if (c.end <= min_off) continue;
            if (!found || c.beg < best) { best = c.beg; found = true; }
        }
    }
    if (found) *out = best;
    return found;
}

bool BamWriter::open(const std::string& path, const BamHeader& hdr, int level) {
    path_ = path;
    if (!bg_.open(path, level)) return false;
    std::vector<uint8_t> h;
    auto put32 = [&](int32_t v) { uint8_t b[4]; memcpy(b, &v, 4); h.insert(h.end(), b, b + 4); };
    h.insert(h.end(), {'B', 'A', 'M', 1});
    put32((int32_t)hdr.text.size());
    h.insert(h.end(), hdr.text.begin(), hdr.text.end());
    put32((int32_t)hdr.names.size());
    for (size_t i = 0; i < hdr.names.size(); ++i) {
        put32((int32_t)hdr.names[i].size() + 1);
        h.insert(h.end(), hdr.names[i].begin(), hdr.names[i].end());
        h.push_back(0);
        put32((int32_t)hdr.lens[i]);
    }
    if (!bg_.write(h.data(), h.size())) return false;
    if (!bg_.flush_block()) return false;   // records start on a fresh block, as samtools does
    size_t n = hdr.names.size();
    refs_.assign(n, BaiRef());
    n_mapped_.assign(n, 0);
    n_unmapped_.assign(n, 0);
    ref_beg_.assign(n, 0);
    ref_end_.assign(n, 0);
    return true;
}

void BamWriter::index_record(int32_t tid, int32_t beg, int32_t end, voff_t v0, voff_t v1, bool mapped) {
    if (tid < 0) { ++n_no_coor_; return; }
    BaiRef& r = refs_[tid];
    if (n_mapped_[tid] + n_unmapped_[tid] == 0) ref_beg_[tid] = v0;
    ref_end_[tid] = v1;
    if (mapped) ++n_mapped_[tid]; else ++n_unmapped_[tid];
    uint32_t bin = (uint32_t)reg2bin(beg, end);
    std::vector<BaiChunk>& ch = r.bins[bin];
    if (!ch.empty() && ch.back().end == v0) ch.back().end = v1;
    else ch.push_back(BaiChunk{v0, v1});
    size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
    if (r.linear.size() <= w1) r.linear.resize(w1 + 1, (voff_t)-1);
    for (size_t w = w0; w <= w1; ++w)
        if (r.linear[w] == (voff_t)-1) r.linear[w] = v0;
}

bool BamWriter::write(int32_t tid, int32_t pos, uint8_t mapq, uint16_t flag, int32_t mtid, int32_t mpos,
                      int32_t isize, const std::string& qname, const uint32_t* cigar, uint32_t n_cigar,
                      const uint8_t* seq4, const uint8_t* qual, int32_t l_qseq, const uint8_t* aux, size_t aux_len) {
    if (n_cigar > 0xffffu) {
        // SAMv1 4.2.2: the 16-bit count cannot hold this CIGAR -- the record carries the placeholder "<l_qseq>S<reference length>N"
        // and the operations go into the tag CG:B:I (what htslib's bam_write1 does; readers swap them back)
        int64_t ref_len = 0;
        for (uint32_t i = 0; i < n_cigar; ++i) {
            const uint32_t op = cigar[i] & 0xf;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += cigar[i] >> 4;
        }
        const uint32_t fake[2] = {(uint32_t)l_qseq << 4 | 4u, (uint32_t)ref_len << 4 | 3u};
        std::vector<uint8_t> aux2(aux_len + 8 + 4 * (size_t)n_cigar);
        if (aux_len) memcpy(aux2.data(), aux, aux_len);
        uint8_t* t = aux2.data() + aux_len;
        memcpy(t, "CGBI", 4);
        memcpy(t + 4, &n_cigar, 4);
        memcpy(t + 8, cigar, 4 * (size_t)n_cigar);
        return write(tid, pos, mapq, flag, mtid, mpos, isize, qname, fake, 2, seq4, qual, l_qseq, aux2.data(), aux2.size());
    }
    int32_t rl = 0;
    for (uint32_t i = 0; i < n_cigar; ++i) {
        uint32_t op = cigar[i] & 0xf;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += (int32_t)(cigar[i] >> 4);
    }
    bool mapped = !(flag & 4);
    int32_t end = pos + ((mapped && rl > 0) ? rl : 1);
    uint32_t bin = (uint32_t)reg2bin(pos < 0 ? 0 : pos, end < 1 ? 1 : end);
    uint32_t l_qname = (uint32_t)qname.size() + 1;
    size_t body = 32 + l_qname + 4 * (size_t)n_cigar + ((size_t)l_qseq + 1) / 2 + (size_t)l_qseq + aux_len;
    buf_.resize(4 + body);
    uint8_t* p = buf_.data();
    int32_t block_len = (int32_t)body;
    uint32_t x[8];
    x[0] = (uint32_t)tid;
    x[1] = (uint32_t)pos;
    x[2] = (bin << 16) | ((uint32_t)mapq << 8) | l_qname;
    x[3] = ((uint32_t)flag << 16) | (n_cigar & 0xffff);
    x[4] = (uint32_t)l_qseq;
    x[5] = (uint32_t)mtid;
    x[6] = (uint32_t)mpos;
    x[7] = (uint32_t)isize;
    memcpy(p, &block_len, 4); p += 4;
    memcpy(p, x, 32); p += 32;
    memcpy(p, qname.c_str(), l_qname); p += l_qname;
    if (n_cigar) memcpy(p, cigar, 4 * (size_t)n_cigar);
    p += 4 * (size_t)n_cigar;
    size_t sb = ((size_t)l_qseq + 1) / 2;
    if (sb) memcpy(p, seq4, sb);
    p += sb;
    if (l_qseq) {
        if (qual) memcpy(p, qual, l_qseq); else memset(p, 0xff, l_qseq);
    }
    p += l_qseq;
    if (aux_len) memcpy(p, aux, aux_len);
    voff_t v0 = bg_.tell();
    if (!bg_.write(buf_.data(), buf_.size())) return false;
    voff_t v1 = bg_.tell();
    index_record(tid, pos < 0 ? 0 : pos, end < 1 ? 1 : end, v0, v1, mapped);
    return true;
}

void BamWriter::resolve_offsets() {
    // the offsets collected while writing are provisional (block sequence numbers): make them real
    for (size_t i = 0; i < refs_.size(); ++i) {
        for (auto& kv : refs_[i].bins)
            for (BaiChunk& c : kv.second) { c.beg = bg_.resolve(c.beg); c.end = bg_.resolve(c.end); }
        for (voff_t& v : refs_[i].linear)
            if (v != (voff_t)-1) v = bg_.resolve(v);
        ref_beg_[i] = bg_.resolve(ref_beg_[i]);
        ref_end_[i] = bg_.resolve(ref_end_[i]);
    }
}

BamWriter::PartIndex BamWriter::take_part_index() {
    PartIndex ix;
    ix.refs.swap(refs_);
    ix.n_mapped.swap(n_mapped_);
    ix.n_unmapped.swap(n_unmapped_);
    ix.ref_beg.swap(ref_beg_);
    ix.ref_end.swap(ref_end_);
    ix.n_no_coor = n_no_coor_;
    return ix;
}

bool BamWriter::open_part(size_t n_refs, int level) {
    path_.clear();
    if (!bg_.open_memory(level, 1)) return false;
    refs_.assign(n_refs, BaiRef());
    n_mapped_.assign(n_refs, 0);
    n_unmapped_.assign(n_refs, 0);
    ref_beg_.assign(n_refs, 0);
    ref_end_.assign(n_refs, 0);
    n_no_coor_ = 0;
    return true;
}

bool BamWriter::finish_part() {
    if (!bg_.finish_memory()) return false;
    resolve_offsets();
    return true;
}

bool BamWriter::write_bai(const std::string& bai_path, PartIndex& ix) {
    FILE* fp = fopen(bai_path.c_str(), "wb");
    if (!fp) return false;
    auto wr = [&](const void* p, size_t n) { return fwrite(p, 1, n, fp) == n; };
    bool ok = wr("BAI\1", 4);
    int32_t n_ref = (int32_t)ix.refs.size();
    ok = ok && wr(&n_ref, 4);
    for (int i = 0; ok && i < n_ref; ++i) {
        BaiRef& r = ix.refs[i];
        bool has = (ix.n_mapped[i] + ix.n_unmapped[i]) > 0;
        int32_t n_bin = (int32_t)r.bins.size() + (has ? 1 : 0);
        ok = wr(&n_bin, 4);
        for (auto& kv : r.bins) {
            uint32_t bin = kv.first;
            int32_t n_chunk = (int32_t)kv.second.size();
            ok = ok && wr(&bin, 4) && wr(&n_chunk, 4) && wr(kv.second.data(), sizeof(BaiChunk) * kv.second.size());
        }
        if (has) {
            uint32_t bin = kMetaBin;
            int32_t n_chunk = 2;
            uint64_t meta[4] = {ix.ref_beg[i], ix.ref_end[i], ix.n_mapped[i], ix.n_unmapped[i]};
            ok = ok && wr(&bin, 4) && wr(&n_chunk, 4) && wr(meta, 32);
        }
        // back-fill unset linear slots from the next set one (hts_idx_finish behaviour)
        for (size_t w = r.linear.size(); w-- > 0;)
            if (r.linear[w] == (voff_t)-1) r.linear[w] = (w + 1 < r.linear.size()) ? r.linear[w + 1] : 0;
        int32_t n_intv = (int32_t)r.linear.size();
        ok = ok && wr(&n_intv, 4);
        if (n_intv) ok = ok && wr(r.linear.data(), 8 * (size_t)n_intv);
    }
    ok = ok && wr(&ix.n_no_coor, 8);
    ok = (fclose(fp) == 0) && ok;
    return ok;
}

bool BamWriter::close() {
    // tell() of the last record may point inside an unflushed block; offsets stay valid after flush
    if (!bg_.close()) return false;
    resolve_offsets();
    PartIndex ix = take_part_index();
    return write_bai(path_ + ".bai", ix);
}

bool Fai::build(const std::string& fasta, std::vector<FaiEntry>* out) {
    FILE* fp = fopen(fasta.c_str(), "rb");
    if (!fp) return false;
    out->clear();
    std::vector<char> buf(1 << 20);
    int64_t off = 0;          // file offset of buf[0]
    FaiEntry cur;
    bool in_seq = false;
    int64_t line_start = 0;   // file offset where the current line began
    int32_t line_bases_cur = 0;
    enum { HDR, SEQ } st = SEQ;
    bool at_line_start = true;
    std::string name;
    bool name_done = false;
    size_t n;
    auto finish_line = [&](int64_t line_end_excl /* offset after '\n' or EOF */, bool had_nl) {
        if (!in_seq) return;
        int32_t width = (int32_t)(line_end_excl - line_start);
        if (line_bases_cur == 0 && width == 0) return;
        if (cur.line_bases == 0 && line_bases_cur > 0) {
            cur.line_bases = line_bases_cur;
            cur.line_width = had_nl ? width : line_bases_cur + 1;
        }
        cur.len += line_bases_cur;
    };
    while ((n = fread(buf.data(), 1, buf.size(), fp)) > 0) {
        for (size_t i = 0; i < n; ++i) {
            char c = buf[i];
            int64_t here = off + (int64_t)i;
            if (at_line_start) {
                line_start = here;
                line_bases_cur = 0;
                at_line_start = false;
                if (c == '>') {
                    if (in_seq) out->push_back(cur);
                    st = HDR;
                    name.clear();
                    name_done = false;
                    in_seq = false;
                    continue;
                }
                st = SEQ;
            }
            if (c == '\n') {
                if (st == HDR) {
                    cur = FaiEntry();
                    cur.name = name;
                    cur.len = 0;
                    cur.offset = here + 1;
                    cur.line_bases = 0;
                    cur.line_width = 0;
                    in_seq = true;
                } else {
                    finish_line(here + 1, true);
                }
                at_line_start = true;
                continue;
            }
            if (st == HDR) {
                if (!name_done) {
                    if (isspace((unsigned char)c)) name_done = true; else name.push_back(c);
                }
            } else if (isgraph((unsigned char)c)) {
                ++line_bases_cur;
            }
        }
        off += (int64_t)n;
    }
    if (!at_line_start && st == SEQ) finish_line(off, false);
    if (in_seq) out->push_back(cur);
    fclose(fp);
    for (FaiEntry& e : *out)
        if (e.line_bases == 0) { e.line_bases = 1; e.line_width = 2; }   // empty sequence
    return true;
}

bool Fai::load(const std::string& fasta) {
    fasta_ = fasta;
    entries_.clear();
    by_name_.clear();
    std::string fai = fasta + ".fai";
    FILE* fp = fopen(fai.c_str(), "r");
    if (fp) {
        char line[65536];
        while (fgets(line, sizeof(line), fp)) {
            FaiEntry e;
            char* tab = strchr(line, '\t');
            if (!tab) continue;
            e.name.assign(line, tab - line);
            long long len, offset;
            int lb, lw;
            if (sscanf(tab + 1, "%lld\t%lld\t%d\t%d", &len, &offset, &lb, &lw) != 4) continue;
            e.len = len; e.offset = offset; e.line_bases = lb; e.line_width = lw;
            entries_.push_back(e);
        }
        fclose(fp);
    } else {
        if (!build(fasta, &entries_)) return false;
        FILE* out = fopen(fai.c_str(), "w");   // fai_load writes the index next to the FASTA
        if (out) {
            for (const FaiEntry& e : entries_)
                fprintf(out, "%s\t%lld\t%lld\t%d\t%d\n", e.name.c_str(), (long long)e.len, (long long)e.offset,
                        e.line_bases, e.line_width);
            fclose(out);
        }
    }
    for (size_t i = 0; i < entries_.size(); ++i)
        if (!by_name_.count(entries_[i].name)) by_name_[entries_[i].name] = (int)i;
    return true;
}

int Fai::find(const std::string& name) const {
    auto it = by_name_.find(name);
    return it == by_name_.end() ? -1 : it->second;
}

// Whole sequence, printable characters only (fai_fetch).  The index gives the line layout (line_bases characters in line_width bytes),
// so line k of the record lands at output offset k * line_bases whatever thread copies it: the lines are dealt over the host threads,
// each pread()s its byte range and copies line by line, checking that every byte it takes is printable (one vectorisable pass).  A
// record whose lines are not what the index says (blanks inside a line, ragged lines) takes the byte-by-byte path, like round 3 did
// for every contig at ~250 MB/s -- a second per 250 Mb contig on the staging thread of the from-files pipeline.
bool Fai::fetch(int i, std::string* out) const {
    if (i < 0 || i >= (int)entries_.size()) return false;
    const FaiEntry& e = entries_[i];
    out->clear();
    const int64_t lb = e.line_bases, lw = e.line_width;
    if (e.len > 0 && lb > 0 && lw > lb && lw - lb <= 2) {
        const int fd = open(fasta_.c_str(), O_RDONLY);
        if (fd >= 0) {
            out->resize((size_t)e.len);
            const int64_t n_lines = (e.len + lb - 1) / lb;
            std::atomic<bool> ok{true};
            const int64_t lines_per_job = std::max<int64_t>(1, (8 << 20) / lw);
            parallel_for((size_t)((n_lines + lines_per_job - 1) / lines_per_job), 1, [&](size_t j0, size_t j1) {
                std::vector<char> buf;
                for (size_t j = j0; j < j1 && ok; ++j) {
                    const int64_t l0 = (int64_t)j * lines_per_job, l1 = std::min<int64_t>(n_lines, l0 + lines_per_job);
                    const int64_t f0 = e.offset + l0 * lw;
                    const int64_t last_len = std::min<int64_t>(lb, e.len - (l1 - 1) * lb);
                    // (a job that is not the contig's last one reads its last line WITH the terminator, so that every line but the contig's
                    // last is checked to end where the index says -- round 4 left the last line of each 8 MiB job unchecked: ADVICE r4)
                    const bool more = l1 < n_lines;
                    const int64_t bytes = (l1 - 1 - l0) * lw + last_len + (more ? 1 : 0);
                    buf.resize((size_t)bytes);
                    int64_t done = 0;
                    while (done < bytes) {
                        const ssize_t g = pread(fd, buf.data() + done, (size_t)(bytes - done), (off_t)(f0 + done));
                        if (g <= 0) { ok = false; return; }
                        done += g;
                    }
                    unsigned bad = 0;
                    for (int64_t l = l0; l < l1; ++l) {
                        const int64_t n = l + 1 < l1 ? lb : last_len;
                        const unsigned char* src = (const unsigned char*)buf.data() + (l - l0) * lw;
                        char* dst = &(*out)[(size_t)(l * lb)];
                        for (int64_t k = 0; k < n; ++k) { dst[k] = (char)src[k]; bad |= (unsigned)((unsigned char)(src[k] - 0x21) > 0x5d); }
                        if (l + 1 < l1 || more) bad |= (unsigned)(src[lb] != '\n' && src[lb] != '\r');      // the line ends where the index says
                    }
                    if (bad) ok = false;
                }
            });
            close(fd);
            if (ok) return true;
            out->clear();
        }
    }
    out->reserve((size_t)e.len);
    FILE* fp = fopen(fasta_.c_str(), "rb");
    if (!fp) return false;
    if (fseeko(fp, (off_t)e.offset, SEEK_SET) != 0) { fclose(fp); return false; }
    std::vector<char> buf(1 << 20);
    while ((int64_t)out->size() < e.len) {
        size_t n = fread(buf.data(), 1, buf.size(), fp);
        if (n == 0) break;
        for (size_t k = 0; k < n && (int64_t)out->size() < e.len; ++k)
            if (isgraph((unsigned char)buf[k])) out->push_back(buf[k]);
    }
    fclose(fp);
    return (int64_t)out->size() == e.len;
}

}  // namespace np
