// Device-side BAM ingest of the short-read path (np1_ingest.hip): host-visible interface, no HIP types.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np_bam.h"

namespace np1ingest {

// One BAM + FASTA pair opened for a run: FASTA index, BAM header, BAM index, a descriptor for pread().
struct BamSource {
    np::Fai fai;
    np::BamHeader hdr;
    np::BaiIndex bai;
    bool have_bai = false;
    int fd = -1;
    uint64_t file_size = 0;
    const uint8_t* map = nullptr;      // the BAM mapped read-only (null: mapping failed, the batches take the host loader)
    bool open(const std::string& fasta, const std::string& bam, std::string* err);
    ~BamSource();
};

// Host half of one batch: draft strings, where the compressed BGZF blocks of the batch's contigs lie in the file (runs of bytes + the block
// table, from a walk over the headers in the mapped file) and the record anchors taken from the index.  Reused from batch to batch.
struct Staging {
    struct Impl;
    Impl* impl;
    Staging();
    ~Staging();
    Staging(const Staging&) = delete;
    Staging& operator=(const Staging&) = delete;
    const std::vector<std::string>& names() const;
    uint64_t compressed_bytes() const;
};

struct Scratch;   // HBM scratch of one device lane (compressed + inflated bytes, record offsets, scan buffers)
Scratch* scratch_create();
void scratch_destroy(Scratch* s);
uint64_t scratch_host_blocks(const Scratch* s);   // blocks the device decoder handed back to the host so far
void scratch_stats(const Scratch* s, double out[5]);      // += {decoder ms, CRC ms, compressed bytes in, inflated bytes out, launches} (HIP events)
void scratch_stats_reset(Scratch* s);

// 0 = staged; 1 = this batch has to take the host loader (index without per-contig offsets); -1 = error (*err set)
int prepare(BamSource& src, const std::vector<std::string>& names, Staging* st, std::string* err);
// 0 = the batch object holds the decoded record stream; 1 = take the host loader (CIGARs in CG tags); -1 = error (np1_last_error)
// replay_bai != nullptr (tasks 2 and 4): the records' virtual offsets come down with the batch and kmer_count / snp_valid replay the
// reference's region iterator on them (np1_replay.h); the index has to stay alive until the pass is done
int ingest(np1_batch* b, Staging* st, bool with_qual, Scratch* scr, const np::BaiIndex* replay_bai = nullptr);

}  // namespace np1ingest
