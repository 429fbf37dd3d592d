// Low-quality-region stage of the long-read consensus (host side): candidate extraction, pseudo-seed by partial-order
// alignment, and the splice back into the window consensus; the alignment of the candidates to the seed and the graph
// consensus of the concatenated regions run in the executor (device).
// reference: source/lib/ctg_cns.c:405-449,620-633,822-1473, dag.c, align.c
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "np2_exec.h"

namespace np2 {

// heaviest-path consensus of up to 50 strings (poa_to_consensus, dag.c:658-694)
std::string poa_consensus(const std::vector<std::string>& seqs);

// a split-read gap cluster as the low-quality stage sees it (ctg_cns.h:210-215; generate_lqseqs_from_cluster)
struct LqCluster {
    uint32_t rs = 0, re = 0;          // window-relative region
    uint32_t i_m = 0;                 // 0 = dropped
    std::vector<std::string> cands;   // read substrings across the gap, in cluster order
};

struct LqRegionIn {   // one low-quality region of the window consensus (window-relative draft positions, inclusive)
    uint32_t start, end;
    uint8_t l;         // 0 insertion-driven, 1 gap cluster, 2 / 3 deletion-driven (ctg_cns.c:1577-1583), 4 HiFi low-qv run (ctg_cns.c:1786)
};
// Re-consensus of the regions (given in DEscending position order, as the reference builds them) and splice into
// *cons (the window's main-line consensus).  The graph consensus of the concatenated regions runs in `exec`.
bool lq_stage(Exec* exec, uint32_t gap_min_len, bool hifi, const std::vector<LqRegionIn>& regions, const std::vector<LqCluster>& clusters,
              const WindowOutput& wo, std::vector<ConsBase>* cons, std::string* err);

}  // namespace np2
