// Split-read structural layer of the long-read consensus (host side): depth track of fully spanning reads,
// low-depth regions, clusters of split-read gaps, the supplementary alignments that become extra tag streams, the read
// substrings across every cluster (candidates of the cluster's low-quality region) and the split points.
// reference: source/lib/ctg_cns.c:2494-2796 (gap clusters, low-depth regions), :2836-3051 (update_align_tags,
// generate_gapseqs, update_split_p), :3224-3328 (depth statistics); only active for contigs > 100 kb with >= 150
// reads and split (SA) supplementary records (:3450,3557).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/nextpolish2.h"
#include "np2_exec.h"
#include "np2_lq.h"

namespace np2 {

struct SvPos { uint32_t s, e; };

struct SvGapRead {            // gap_ (ctg_cns.h:197-203): one split read's view of a gap
    SvPos gap;                // contig coordinates of the gap; later the read coordinates of the substring across the cluster
    uint32_t p_id, s_id;      // tag stream of the primary / (fs of the supplementary, later its tag stream)
    uint32_t p_s, s_s;        // query start of the primary's kept columns / (ds of the supplementary, later its query start)
    uint32_t l = 0;
    std::vector<uint8_t> dseq;   // packed bases of the primary record
};
struct SvSupAln {             // sup_aln (ctg_cns.h:221-225)
    uint32_t fs, ds;
    std::vector<uint32_t> cigar;
};
struct SvCluster {            // gap_cluster (ctg_cns.h:210-215)
    SvPos r{0, 0};
    uint32_t median = 0, i_m = 0;
    std::vector<uint32_t> gap;   // indices into SvWindow::gaps (at most 120)
};

struct SvWindow {             // per-window state
    std::vector<uint16_t> ref_ds;      // reads per 10-bp bin (after finish_depth)
    std::vector<int32_t> depth_diff;   // +1 / -1 at the ends of every counted read span
    void finish_depth();               // sums the difference array into ref_ds
    std::vector<SvSupAln> sup_alns;
    std::vector<SvGapRead> gaps;
    std::vector<SvPos> ld_regs;
    std::vector<SvCluster> clusters;
    void reset(size_t n_ds) { ref_ds.assign(n_ds, 0); depth_diff.assign(n_ds + 2, 0); sup_alns.clear(); gaps.clear(); ld_regs.clear(); clusters.clear(); }
};
struct SvContig {             // state that persists across the windows of a contig (ctg_cns.c:3450-3454)
    int brk_g = 0, rreads_w = 0, ref_d = 0, ref_ide = 0;
    std::vector<SvPos> rreads;        // spans of the first 50 000 kept reads
    std::vector<SvPos> split_ps;
};

int sv_cal_rreads_w(std::vector<SvPos>& rs);                                   // cal_rreads_w (reorders rs like the reference)
void sv_update_ref_d(SvWindow& win, int w, const SvPos& p, int32_t s);
int sv_cal_ref_d(const std::vector<uint16_t>& r, int32_t l);
int sv_cal_ref_ide(const ref_qv* qv, uint32_t l);
void sv_update_ld_regs(std::vector<SvPos>* regs, const std::vector<uint16_t>& r, int32_t l, int w, int d);
void sv_update_ld_regs_with_refqv(std::vector<SvPos>* regs, const std::vector<uint16_t>& r, const ref_* ref, int32_t w, int32_t s_t, int32_t e_t,
                                  int32_t d_t, uint32_t ide_t, uint32_t ort_t, uint32_t irt_t);
int sv_update_gap_cluster(SvWindow* w, int rw, int d, int32_t ref_s);
// update_align_tags: which supplementary alignments become streams.  sup_span[k] = span of gaps[k]'s supplementary
// alignment against the window (computed for every gap up front).  Appends StreamRefs (set 1, rec = gap index) and
// returns the new stream count.
uint32_t sv_update_align_tags(SvWindow* w, const std::vector<SpanOut>& sup_span, uint32_t seq_count, int32_t ref_s, std::vector<StreamRef>* streams);
// generate_gapseqs: the window's tag streams stay with the executor; the read coordinates at the cluster's ends are asked of it
// (Exec::read_coords), for all clusters of the window at once
bool sv_generate_gapseqs(SvWindow* w, const WindowOutput& wo, int32_t s_, Exec* exec, std::string* err);
void sv_update_split_p(std::vector<SvPos>* split_ps, const SvWindow& w, int32_t s, int32_t l, const ref_* ref);
// clusters as the low-quality stage sees them (ascending position)
std::vector<LqCluster> sv_lq_clusters(const SvWindow& w);

}  // namespace np2
