// Minimal BAM / BAI / FAI access for the polishing hot path, written from the SAM/BAM
// specification (SAMv1 §4.2, §5.2, §5.3).  It replaces exactly the htslib surface the
// reference's nextpolish1 core uses:
//   bam_hdr_read / bam_name2id          (reference: source/lib/contig.c:65-72)
//   bam_itr_queryi + sam_itr_next       (reference: source/lib/contig.c:172-174, 692-694)
//   bam_read1 sequential scan           (reference: source/lib/config.c:80-101)
//   fai_load / faidx_seq_len / fai_fetch(reference: source/lib/contig.c:35-36, 1119-1128)
// plus a writer side (BAM + BAI + FAI) used by the synthetic-workload generator and the tests.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "np_bgzf.h"

namespace np {

struct BamHeader {
    std::string text;
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    int name2id(const std::string& n) const;
};

// One alignment record, decoded in place from the BAM block.
struct BamRec {
    int32_t tid = -1, pos = -1;
    uint8_t mapq = 0;
    uint16_t bin = 0, flag = 0;
    uint32_t n_cigar = 0;
    int32_t l_qseq = 0, mtid = -1, mpos = -1, isize = 0;
    uint8_t l_qname = 0;
    std::vector<uint8_t> data;   // qname | cigar | seq | qual | aux   (as stored in BAM)
    const uint8_t* ext = nullptr;   // zero-copy readers: the same bytes inside the BGZF window (valid until the next record)
    size_t ext_len = 0;             // ... and how many there are (the record without its 32 fixed bytes)
    const uint8_t* end() const { return ext ? ext + ext_len : data.data() + data.size(); }   // behind the optional fields (an owned copy: + 8 zero bytes of pad)
    const uint8_t* base() const { return ext ? ext : data.data(); }
    const char* qname() const { return reinterpret_cast<const char*>(base()); }
    const uint32_t* cigar() const { return reinterpret_cast<const uint32_t*>(base() + l_qname); }
    const uint8_t* seq() const { return base() + l_qname + 4 * (size_t)n_cigar; }
    const uint8_t* qual() const { return seq() + ((size_t)l_qseq + 1) / 2; }
    // reference-consumed length, htslib bam_cigar2rlen semantics (M,D,N,=,X consume the reference)
    int32_t rlen() const;
    // bam_endpos of the reference's htslib 1.9: pos + rlen (also when rlen is 0), or pos + 1 for unmapped / CIGAR-less records
    int32_t endpos() const;
};

class BamReader {
public:
    bool open(const std::string& path);
    const BamHeader& header() const { return hdr_; }
    // returns 1 on success, 0 on clean EOF, -1 on error
    int next(BamRec& r);
    // records may point into the reader's window instead of owning a copy (r.ext; r.data is then stale): for callers
    // that consume a record before asking for the next one and only use the accessors
    void set_zero_copy(bool on) { zero_copy_ = on; }
    bool seek(voff_t v) { return bg_.seek(v); }
    voff_t tell() const { return bg_.tell(); }
    voff_t first_record_offset() const { return first_rec_; }

private:
    BgzfReader bg_;
    BamHeader hdr_;
    voff_t first_rec_ = 0;
    bool zero_copy_ = false;
};

// BAI index (SAMv1 §5.2).
struct BaiChunk { voff_t beg, end; };
struct BaiRef {
    std::map<uint32_t, std::vector<BaiChunk>> bins;
    std::vector<voff_t> linear;
};
struct BaiIndex {
    std::vector<BaiRef> refs;
    bool load(const std::string& path);
    // Smallest virtual offset at which a record of `tid` overlapping [beg,end) can start,
    // following htslib's hts_itr_query min_off + reg2bins chunk selection; returns false if
    // the index holds no chunk for the region (=> the query yields no records).
    bool region_start(int tid, int32_t beg, int32_t end, voff_t* out) const;
};

int reg2bin(int64_t beg, int64_t end);

// Writer: coordinate-sorted BAM + its BAI.
class BamWriter {
public:
    bool open(const std::string& path, const BamHeader& hdr, int level = 1);
    // cigar: BAM-encoded ops; seq4: 4-bit packed ((l_qseq+1)/2 bytes); qual: l_qseq bytes
    bool write(int32_t tid, int32_t pos, uint8_t mapq, uint16_t flag, int32_t mtid, int32_t mpos, int32_t isize,
               const std::string& qname, const uint32_t* cigar, uint32_t n_cigar, const uint8_t* seq4,
               const uint8_t* qual, int32_t l_qseq, const uint8_t* aux = nullptr, size_t aux_len = 0);   // aux: raw optional fields
    bool close();   // also writes <path>.bai
    // Part mode (several writers side by side, one thread each: np_synth.cpp write_streams_files): the records of a subset of the
    // reference sequences into a memory sink, no header, no EOF marker; finish_part() leaves the compressed bytes in part_bytes()
    // and the index entries in part_index(), their offsets relative to the first byte of the part.
    struct PartIndex {
        std::vector<BaiRef> refs;
        std::vector<uint64_t> n_mapped, n_unmapped;
        std::vector<voff_t> ref_beg, ref_end;
        uint64_t n_no_coor = 0;
    };
    bool open_part(size_t n_refs, int level);
    bool finish_part();
    std::vector<uint8_t>& part_bytes() { return bg_.memory(); }
    PartIndex take_part_index();
    // <bam>.bai from index entries whose offsets are final
    static bool write_bai(const std::string& bai_path, PartIndex& ix);
private:
    void resolve_offsets();
    void index_record(int32_t tid, int32_t beg, int32_t end, voff_t v0, voff_t v1, bool mapped);
    std::string path_;
    BgzfWriter bg_;
    std::vector<BaiRef> refs_;
    std::vector<uint64_t> n_mapped_, n_unmapped_;
    std::vector<voff_t> ref_beg_, ref_end_;
    std::vector<uint8_t> buf_;
    uint64_t n_no_coor_ = 0;
};

// FASTA index (.fai) + whole-sequence fetch.
struct FaiEntry { std::string name; int64_t len, offset; int32_t line_bases, line_width; };
class Fai {
public:
    // Loads <fasta>.fai, building (and trying to write) it when absent, like fai_load.
    bool load(const std::string& fasta);
    int nseq() const { return (int)entries_.size(); }
    const FaiEntry& entry(int i) const { return entries_[i]; }
    int find(const std::string& name) const;
    // whole sequence, printable characters only, case preserved (fai_fetch "name:0-len")
    bool fetch(int i, std::string* out) const;
    static bool build(const std::string& fasta, std::vector<FaiEntry>* out);
private:
    std::string fasta_;
    std::vector<FaiEntry> entries_;
    std::map<std::string, int> by_name_;
};

}  // namespace np
