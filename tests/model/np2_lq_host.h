// TEST INFRASTRUCTURE (host executor of tests/model): sequential restatement of the candidate-to-seed alignment
// (source/lib/align.c:39-177) and of the piece rules of generate_consensus_trimed (source/lib/ctg_cns.c:1325-1391) that the
// product runs on the device (nextpolish_amd/csrc/np2_ond_dev.h, np2_exec_hip.hip:run_lq_aligned).  The product never links this.
#pragma once
#include <string>
#include <vector>

#include "../../nextpolish_amd/csrc/np2_exec.h"

namespace np2 {
struct OndAln {
    int aln_len = 0, aln_t_len = 0, aln_q_len = 0;
    std::string t_aln_str, q_aln_str;
};
// align (align.c:39-177); false = no alignment within the diagonal/band limits
bool ond_align(const char* query_seq, int q_len, const char* target_seq, int t_len, OndAln* aln);
// the 30 concatenated gapped string pairs of an LqAlignInput
void lq_concatenate_host(const LqAlignInput& in, LqInput* out);
}  // namespace np2
