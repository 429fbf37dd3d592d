// Synthetic polishing workload generator (SURVEY.md §8d "synthetic inputs"): a truth genome, a
// draft assembly derived from it by long-read-assembly-like edits, and paired-end short reads
// sampled from the truth whose CIGARs against the *draft* follow from the known edit script.
// Output is a decoded ReadStream (for the HBM-resident benchmark) that can also be serialised as
// FASTA(+.fai) and coordinate-sorted BAM(+.bai) so the CPU reference binary reads the same data.
// Deterministic for a given parameter block (own xoshiro256** stream, no libc rand).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "np_stream.h"

#include "../../include/nextpolish1.h"
typedef np1_synth_params np_synth_params;   // field documentation: see the comments below
// seed; n_contigs + contig_len[] (truth lengths); depth (mean short-read depth); read_len (150);
// frag_mean/frag_sd (300/30); draft_sub/draft_indel (per-base draft error rates 0.001/0.005);
// draft_lower (fraction of draft bases written lowercase); read_sub/read_indel (0.001/0.0001);
// softclip_rate (0.005); dup/supp/sec/unmapped_rate (flag mix); lowmapq_rate (mapq in [0,30]);
// weird_rate (rare CIGAR shapes: H on a primary record, N, =/X ...; 0 for benchmarks); with_qual.

typedef np1_synth_long_params np_synth_long_params;

namespace np {
bool synth_long_stream(const np_synth_long_params& p, const std::string& contig_name_prefix, ReadStream* out);
void synth_default_params(np_synth_params* p);
// Generates the batch.  `contig_name_prefix` + index names the contigs.
bool synth_stream(const np_synth_params& p, const std::string& contig_name_prefix, ReadStream* out);
// The same workloads over contigs that are handed in (names + sequences: the assembly a polishing step wrote) -- the re-mapped reads of
// the next step of a multi-step run, by construction instead of by a mapper (np_synth.cpp).
bool synth_stream_on(const np_synth_params& p, const std::vector<std::string>& names, const std::vector<std::string>& drafts, ReadStream* out);
bool synth_long_stream_on(const np_synth_long_params& p, const std::vector<std::string>& names, const std::vector<std::string>& drafts, ReadStream* out);
// Serialises a stream (needs qualities loaded unless write_qual_ff) to fasta(+.fai) and bam(+.bai).
bool write_stream_files(const ReadStream& s, const std::string& fasta, const std::string& bam, int bgzf_level,
                        std::string* err, const uint8_t* aux_pool = nullptr,
                        const uint64_t* aux_off = nullptr);   // aux: raw optional fields per record (tests: SA tags)
// Several streams as ONE FASTA + ONE coordinate-sorted BAM (contigs of stream 0, then of stream 1, ...).
// qual_model (streams without qualities): 0 = the BAM's "no qualities" bytes (0xff), 1 = Illumina-like binned qualities
// (synth_binned_qualities), 2 = uniformly random in [25, 40].  More than one stream: the streams are written side by side on the host threads (parts of one file).
bool write_streams_files(const std::vector<const ReadStream*>& ss, const std::string& fasta, const std::string& bam, int bgzf_level,
                         std::string* err, const uint8_t* aux_pool = nullptr, const uint64_t* aux_off = nullptr, int qual_model = 0);
void synth_binned_qualities(uint64_t serial, bool reverse, int32_t l, uint8_t* out);
// diploid workload of task 3: short-read and long-read stream over the same contigs (np1_diploid_params, include/nextpolish1.h)
bool synth_diploid_streams(const np1_diploid_params& p, const std::string& contig_name_prefix, ReadStream* sr, ReadStream* lr);
}  // namespace np
