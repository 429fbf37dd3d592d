// Window executor interface of the long-read path: the host pipeline (np2_pipeline.cpp) prepares one window's
// records in merge order and hands them to an executor that runs alignment spans -> tags -> link graph -> chain DP
// -> backtrace.  The product links the HIP executor (np2_exec_hip.hip); the tests' lockstep model links a host
// executor that runs the same per-lane bodies (tests/model/np2_exec_host.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <cstdlib>
#include <new>
#include <vector>

#include "np2_core.h"

namespace np2 {

struct ConsBase {   // consensus_base, ctg_cns.h:103-107
    uint32_t pos;   // window-relative draft position
    char qv;
    char base;
};

// alignment records handed to the executor: BAM core fields of the path (position, CIGAR, packed bases)
// Storage of the two big host arrays that cross PCIe with every window (CIGAR operations and packed bases: ~170 MB of a 5 Mb / 20x
// window).  It comes from a pair of hooks so that the HIP executor can make it page-locked memory of its own (np2_exec_hip.hip,
// NP2_PINNED_RECORDS=1): np_hostcopy.h then lets the DMA engine read the arrays where they are, instead of copying them through its pinned
// ring first (measured: no gain on the long-read leg, so it is opt-in).  Default and host model: malloc.  A block remembers what it was made by.
struct BigMem {
    static void* (*make)(size_t bytes);      // returns page-locked memory or nullptr
    static void (*drop)(void* p);
};
template <class T> struct BigAlloc {
    using value_type = T;
    BigAlloc() = default;
    template <class U> BigAlloc(const BigAlloc<U>&) {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T) + 64;
        char* p = BigMem::make ? static_cast<char*>(BigMem::make(bytes)) : nullptr;
        const bool hooked = p != nullptr;
        if (!p) p = static_cast<char*>(malloc(bytes));
        if (!p) throw std::bad_alloc();
        p[0] = hooked ? 1 : 0;
        return reinterpret_cast<T*>(p + 64);
    }
    void deallocate(T* q, size_t) {
        char* p = reinterpret_cast<char*>(q) - 64;
        if (p[0]) BigMem::drop(p); else free(p);
    }
    template <class U> bool operator==(const BigAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const BigAlloc<U>&) const { return false; }
};

struct RecordSet {
    std::vector<int32_t> pos;
    std::vector<uint32_t> n_cigar;
    std::vector<uint32_t> q0;           // leading clip length = query coordinate of the first aligned base
    std::vector<uint64_t> cigar_off, seq_off;
    std::vector<uint32_t, BigAlloc<uint32_t>> cigar;
    std::vector<uint8_t, BigAlloc<uint8_t>> seq;
    size_t size() const { return pos.size(); }
    void clear() { pos.clear(); n_cigar.clear(); q0.clear(); cigar_off.clear(); seq_off.clear(); cigar.clear(); seq.clear(); }
    void add(int32_t p, const uint32_t* cg, uint32_t nc, const uint8_t* sq, size_t seq_bytes, uint32_t q_start) {
        pos.push_back(p); n_cigar.push_back(nc); q0.push_back(q_start);
        cigar_off.push_back(cigar.size()); seq_off.push_back(seq.size());
        cigar.insert(cigar.end(), cg, cg + nc);
        seq.insert(seq.end(), sq, sq + seq_bytes);
    }
};

// clip_aln + get_align_shift of one record against the window (np2k::align_span)
struct SpanOut {
    uint32_t col0, aln_len, aln_t_s, aln_t_e, aln_q_s;
    uint32_t bad;   // the CIGAR carries an op the reference aborts on
};

struct StreamRef {      // one tag stream of the window besides the seed
    uint32_t set;       // 0: window records, 1: supplementary alignments of split reads (structural layer)
    uint32_t rec;       // index into that record set
    SpanOut span;
};

struct WindowInput {
    const char* contig_seq = nullptr;   // decoded contig (A/C/G/T), indexable by contig coordinate
    uint64_t contig_serial = 0;         // changes whenever contig_seq holds a different contig (executors may cache the upload)
    int32_t s = 0, e = 0;               // window [s, e)
    uint32_t gap_min_len = 3;           // 3 ONT, 5 otherwise (ctg_cns.c:3436-3442)
    int read_type = np2k::READS_ONT;
    RecordSet recs;                     // candidate records in merge order
    RecordSet sup;                      // supplementary alignments (read bases of the primary, CIGAR of the supplementary record)
    bool want_tags = false;             // copy the tag streams back to the host too (only the structural layer reads them there)
    // > 0: the executor may also mark where the low-quality scans of the window consensus have anything to look at (WindowOutput::
    // trig_del / trig_ins, ctg_cns.c:1562-1725 loop heads) with this gap_min_ratio1; the scans then visit those positions only
    float lq_ratio1 = 0.f;
    std::vector<StreamRef> streams;     // the streams to pile up, in order (the seed -- the window against itself -- comes first, implicitly)
};

struct WindowOutput {
    uint32_t seq_count = 0;                       // seed + streams (aligned_seq_count of the reference)
    std::vector<np2k::ColStat> stat;              // e - s + 1 columns
    // tag streams of the seed (index 0) and the given streams, in order: stream i = tags[tag_off[i] ..), window-relative
    // start / exclusive end positions
    std::vector<uint64_t> tag_off;
    std::vector<uint32_t> aln_t_s, aln_t_e;
    std::vector<uint8_t> tags;
    std::vector<ConsBase> cons;                   // main-line consensus, window order (before the LQ stage)
    // Optional (empty = not computed: the scans test every position themselves).  Bit i of word i / 64:
    //   trig_del  consensus base i (i >= 1) passes the loop head of get_l_del_regions: NOT (l_del < coverage * 0.3 and pos < previous pos + 20)
    //   trig_ins  consensus base i passes the loop head of the insertion scan: NOT ((float) l_ins < (float) coverage * gap_min_ratio1)
    std::vector<uint64_t> trig_del, trig_ins;
};

// the concatenated low-quality regions of a window: up to 30 gapped string pairs over one target coordinate space
// (generate_consensus_trimed, ctg_cns.c:1287-1414)
struct LqInput {
    std::vector<std::string> t, q;   // same length per pair; t uses '-' for insertion columns
    uint32_t t_len = 0;              // target positions (non-gap characters of every t)
    uint32_t gap_min_len = 3;
    bool hifi = false;               // HiFi branch of the DP (coefficient 4, CLR-style tie rule)
};

// generate_consensus_trimed (ctg_cns.c:1287-1414) with the candidate-to-seed alignments done by the executor: per valid
// low-quality region (in the order of concatenation: descending position) its current seed and its ranked candidates; the
// executor aligns every candidate that qualifies to the seed (align.c:39-177), applies the fill rules for the others, builds
// the LQSEQ_MAX_COUNT concatenated gapped string pairs and runs the graph consensus on them.
struct LqAlignRegion {
    uint32_t seed_off, seed_len;     // into LqAlignInput::chars
    uint32_t first_cand, n_cand;     // candidates [first_cand, first_cand + n_cand) of cand_off / cand_len, best first; n_cand >= 1
};
struct LqAlignInput {
    std::string chars;                       // seeds and candidate strings
    std::vector<uint32_t> cand_off, cand_len;
    std::vector<LqAlignRegion> regions;
    uint32_t gap_min_len = 3;
    bool hifi = false;
};
constexpr int LQ_ROUNDS = 30;   // LQSEQ_MAX_COUNT of the reference: concatenated alignments per consensus

// pseudo-seeds of many low-quality regions at once (poa_to_consensus, dag.c:658-694): job j = strings [job_first[j], job_first[j] +
// job_n[j]) of `chars` (every string followed by a NUL: the reference's graph can pick the terminator up)
struct PoaBatch {
    std::string chars;
    std::vector<uint32_t> str_off, str_len;
    std::vector<uint32_t> job_first, job_n;
};

// "the bases of stream `stream` at window positions [start, end]" (inclusive; gap tags dropped): one candidate string
// of a low-quality region (generate_lqseqs_from_tags, ctg_cns.c:822-870)
struct SubReq { uint32_t stream, start, end; };
struct CoordReq { uint32_t stream, col, through_col; };

class Exec {
  public:
    virtual ~Exec() {}
    // false + *err on failure (never a silent fallback)
    // spans of every record of in.recs (set 0) or in.sup (set 1) against the window
    virtual bool compute_spans(const WindowInput& in, int set, std::vector<SpanOut>* spans, std::string* err) = 0;
    // tags of in.streams -> link graph -> chain DP -> backtrace
    virtual bool run_window(const WindowInput& in, WindowOutput* out, std::string* err) = 0;
    // link graph + DP variant + backtrace of get_lqseqs_from_align_tags (ctg_cns.c:986-1163, non-HiFi branch):
    // *cons_rev = consensus characters in backtrace order (last column first), as the reference leaves them
    virtual bool run_lq(const LqInput& in, std::string* cons_rev, std::string* err) = 0;
    virtual bool run_lq_aligned(const LqAlignInput& in, std::string* cons_rev, std::string* err) = 0;
    // candidate strings from the tag streams of the LAST run_window call (they stay with the executor until the next
    // run_window / run_lq): request i -> bases[off[i] .. off[i + 1]).  Requests come grouped by ascending stream.
    virtual bool extract(const std::vector<SubReq>& req, std::vector<uint32_t>* off, std::string* bases, std::string* err) = 0;
    // one consensus string per job of the batch
    virtual bool run_poa(const PoaBatch& in, std::vector<std::string>* out, std::string* err) = 0;
    // "how many bases of its read has stream `stream` used up when it reaches window column `col`" on the tag streams of the LAST
    // run_window call: bases carried by the tags in front of the column's own (first) tag, plus that tag's base when through_col is set
    // (the structural layer's cut of a split read across a gap cluster: generate_gapseqs, ctg_cns.c:2898-2971).  col must lie inside
    // the stream's span.
    virtual bool read_coords(const std::vector<CoordReq>& req, std::vector<uint32_t>* bases, std::string* err) = 0;
};

// provided by whichever executor is linked (HIP in the product library)
Exec* make_exec(std::string* err);

}  // namespace np2
