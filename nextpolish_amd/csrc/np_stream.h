// Decoded record stream: the host-side (and, mirrored, HBM-side) layout of "draft contigs +
// their position-sorted alignment records" that every kernel of the short-read polishing
// path consumes.  One ReadStream = one batch of whole contigs.
//
// It carries exactly the BAM core fields the reference's per-read code touches
// (reference: source/lib/contig.c:202-358,632-686 read core.pos/flag/n_cigar/l_qseq/isize/qual,
// the CIGAR array, the 4-bit sequence and, for kmer_count, the base qualities
// source/lib/kmercount.c:365-465).  Algorithmic bytes per record = 32 (fixed fields below)
// + 4*n_cigar + ceil(l_qseq/2) [+ l_qseq with qualities]  (SURVEY.md §8d).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace np {

struct ReadStream {
    // ---- contigs of this batch, in request order
    std::vector<std::string> names;
    std::vector<int32_t> ctg_len;       // draft length of each contig
    std::vector<uint32_t> ctg_off;      // size n+1: offset of contig c in `draft` (global draft coordinate)
    std::string draft;                  // concatenated raw FASTA characters (case preserved)
    std::vector<uint64_t> read_begin;   // size n+1: reads [read_begin[c], read_begin[c+1]) belong to contig c

    // ---- records, BAM file order inside each contig (fixed part: 32 B / record)
    std::vector<int32_t> pos;           // 0-based leftmost draft coordinate inside the contig
    std::vector<uint32_t> ctg;          // contig index inside this batch
    std::vector<uint16_t> flag;
    std::vector<uint32_t> n_cigar;   // 32 bits: a CIGAR swapped in from a CG tag has more than 65 535 operations (htslib: bam1_core_t.n_cigar is 32-bit too)
    std::vector<int32_t> l_qseq;
    std::vector<uint64_t> cigar_off;    // index of the first op in `cigar`
    std::vector<uint64_t> seq_off;      // byte offset of the first base pair in `seq`
    // ---- only needed by kmer_count (filters + haplotype scoring)
    std::vector<uint8_t> mapq;
    std::vector<int32_t> isize;
    std::vector<uint64_t> qual_off;     // byte offset into `qual` (when loaded)

    // ---- pools
    std::vector<uint32_t> cigar;        // BAM-encoded ops: len<<4 | op
    std::vector<uint8_t> seq;           // 4-bit bases, two per byte, high nibble first; each record byte aligned
    std::vector<uint8_t> qual;          // phred bytes; empty unless requested
    // ---- only filled by load_stream: BGZF virtual offset of every record and of the byte behind it (htslib's bgzf_tell convention),
    // what an index-driven reader of the same file sees (the replay of the reference's region iterator needs them)
    std::vector<uint64_t> voff, voff_end;

    size_t n_reads() const { return pos.size(); }
    size_t n_contigs() const { return names.size(); }
    void clear();
    // bytes the roofline accounting uses: records + draft (+ quals)
    uint64_t algorithmic_input_bytes(bool with_qual) const;
};

// Loads the given contigs (all contigs of the FASTA index, in index order, when `names` is empty)
// and every BAM record placed on them.  Uses <bam>.bai to seek when a subset is requested and a
// single sequential pass otherwise.  Returns false and fills *err on any I/O / format problem.
bool load_stream(const std::string& fasta, const std::string& bam, const std::vector<std::string>& names,
                 bool with_qual, ReadStream* out, std::string* err);

// Mean-insert-size probe of config_init (reference: source/lib/config.c:80-101): scans the first
// records of the BAM; returns sum/count with count starting at 1, and the first qualifying l_qseq.
struct BaiIndex;
// One TILE of one contig (intra-contig tiling): the records of `name` that touch draft bases [e_lo, e_hi] -- as the
// reference's walk sees them: a record ends where its M and D operations end, contig.c:262-326 -- read through the index, in file order, as
// a stream of ONE contig whose draft is the hull [*lo, *hi) of those records and of the interval itself, widened by one base on each side
// (position 0 and the last base of a contig are special to the walk, and no record of the tile may meet an artificial one).  Record
// positions are relative to *lo.  draft_whole: the contig's draft (fetched once by the caller); *n_seen = records read to find them.
bool load_stream_region(const std::string& bam, const BaiIndex& bai, const std::string& name, const std::string& draft_whole, int32_t e_lo, int32_t e_hi,
                        ReadStream* out, int32_t* lo, int32_t* hi, std::string* err);
bool bam_insert_probe(const std::string& bam, uint32_t count_read_ins, uint32_t max_ins_len, uint32_t* mean_out,
                      int32_t* read_len_out);

}  // namespace np
