// Host lockstep executor of the long-read window pipeline -- TEST INFRASTRUCTURE (tests/model): runs the same
// per-lane bodies the HIP kernels compile (nextpolish_amd/csrc/np2_core.h) in plain loops, so the algorithm can be
// validated against the compiled reference on the CPU.  Never linked into the product library.
#include <algorithm>
#include <cstring>

#include "../../nextpolish_amd/csrc/np2_exec.h"

namespace np2 {
namespace {

using namespace np2k;

struct HostStat {
    ColStat* st;
    void coverage(uint32_t p) { ++st[p].coverage; }
    void max_size(uint32_t p, uint32_t delta) { if (delta >= st[p].max_size) st[p].max_size = (uint16_t)(delta + 1); }
    void l_ins(uint32_t p) { ++st[p].l_ins; }
    void l_del(uint32_t p) { ++st[p].l_del; }
};

class HostExec : public Exec {
  public:
    bool run_window(const WindowInput& in, WindowOutput* out, std::string* err) override {
        const int32_t s = in.s, e = in.e, l = e - s;
        const size_t n = in.n_reads();
        out->kept.assign(n, 0);
        out->stat.assign((size_t)l + 1, ColStat{0, 0, 0, 0});
        out->tag_off.clear(); out->aln_t_s.clear(); out->aln_t_e.clear(); out->tags.clear(); out->cons.clear();
        out->bad_cigar = false;
        HostStat hs{out->stat.data()};
        // ---- seed: the window against itself (ctg_cns.c:3458-3469)
        std::vector<uint8_t> seed_seq(((size_t)l + 1) / 2 + 1, 0);
        for (int32_t i = 0; i < l; ++i) {
            const char c = in.contig_seq[s + i];
            const uint32_t code = c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : 8;
            seed_seq[(size_t)i >> 1] |= (uint8_t)(code << ((~i & 1) << 2));
        }
        const uint32_t seed_cigar = (uint32_t)l << 4;
        auto add_stream = [&](const ReadView& rv, const AlnSpan& a) {
            out->tag_off.push_back(out->tags.size());
            out->tags.resize(out->tags.size() + (a.aln_len + 1) / 2 + 1, 0);
            const uint32_t te = emit_tags(rv, in.contig_seq, a, s, in.gap_min_len, out->tags.data() + out->tag_off.back(), hs);
            out->aln_t_s.push_back(a.aln_t_s - (uint32_t)s);
            out->aln_t_e.push_back(te);
        };
        {
            ReadView rv{s, 1, &seed_cigar, seed_seq.data()};
            add_stream(rv, AlnSpan{0, (uint32_t)l, (uint32_t)s, (uint32_t)e});
        }
        // ---- candidate records in merge order
        for (size_t i = 0; i < n; ++i) {
            ReadView rv{in.pos[i], in.n_cigar[i], in.cigar.data() + in.cigar_off[i], in.seq.data() + in.seq_off[i]};
            uint32_t N, rf_len, rd_len;
            bool bad;
            cigar_totals(rv, &N, &rf_len, &rd_len, &bad);
            if (bad) { out->bad_cigar = true; return true; }
            const AlnSpan a = align_span(rv, in.contig_seq, s, e);
            if (a.aln_t_s > a.aln_t_e - 500u) continue;                   // unsigned, as in the reference (ctg_cns.c:3540)
            const uint32_t ts = a.aln_t_s - (uint32_t)s, te = a.aln_t_e - (uint32_t)s;
            if (ts > (uint32_t)l || te > (uint32_t)l) { *err = "alignment outside its window"; return false; }
            const ColStat& cs = out->stat[ts];
            const ColStat& ce = out->stat[te];
            if ((cs.coverage > 3000 && ce.coverage > 3000) ||
                (cs.coverage > 500 && ce.coverage > 500 && (double)in.aligned_q[i] < in.l_qseq[i] * 0.9)) continue;
            out->kept[i] = 1;
            add_stream(rv, a);
        }
        out->seq_count = (uint32_t)out->tag_off.size();
        // ---- link observations per column (update_msa, ctg_cns.c:324-365)
        std::vector<uint32_t> col_cnt((size_t)l + 2, 0);
        auto walk = [&](bool count_only, std::vector<LinkObs>* obs, std::vector<uint32_t>* cursor) {
            for (uint32_t rd = 0; rd < out->seq_count; ++rd) {
                const uint8_t* tg = out->tags.data() + out->tag_off[rd];
                uint32_t d = 0;
                Tag p1{0, 0, 0};
                uint64_t pp = KEY_HEAD, ppp = KEY_HEAD;
                uint32_t pp_base = 0;
                while (next_tag(tg, out->aln_t_s[rd], &d, &p1)) {
                    const uint64_t key = node_key(p1.t_pos, p1.delta, p1.q_base);
                    if (p1.q_base == 6 || pp_base == 6) { ppp = pp; pp = key; pp_base = p1.q_base; continue; }
                    if (count_only) ++col_cnt[(size_t)p1.t_pos];
                    else {
                        LinkObs& o = (*obs)[(*cursor)[(size_t)p1.t_pos]++];
                        o.pp = pp; o.ppp = ppp; o.rd = rd; o.delta = (uint16_t)p1.delta; o.base = (uint8_t)p1.q_base; o.pad = 0;
                    }
                    ppp = pp; pp = key; pp_base = p1.q_base;
                }
            }
        };
        walk(true, nullptr, nullptr);
        std::vector<uint32_t> col_off((size_t)l + 2, 0);
        for (int32_t p = 0; p <= l; ++p) col_off[(size_t)p + 1] = col_off[(size_t)p] + col_cnt[(size_t)p];
        const size_t total = col_off[(size_t)l + 1];
        std::vector<LinkObs> obs(total);
        {
            std::vector<uint32_t> cursor(col_off.begin(), col_off.end());
            walk(false, &obs, &cursor);
        }
        // (streams are walked in order here, so every bucket is already ordered by (rd, delta); the device sorts)
        std::vector<Entry> entries(total);
        std::vector<Node> nodes(total);
        std::vector<uint32_t> col_nn((size_t)l + 1, 0);
        for (int32_t p = 0; p <= l; ++p)
            col_nn[(size_t)p] = build_column(obs.data() + col_off[(size_t)p], col_cnt[(size_t)p], entries.data() + col_off[(size_t)p],
                                             nodes.data() + col_off[(size_t)p]);
        // ---- chain DP, column after column
        MsaView mv{col_off.data(), col_nn.data(), nodes.data(), entries.data(), out->stat.data()};
        long long gbest = INT64_MIN;
        uint64_t gkey = node_key(0, 0, 0xff);
        for (int32_t p = 0; p < l; ++p) {
            switch (in.read_type) {
                case READS_CLR: dp_column<READS_CLR>(mv, p, l, &gbest, &gkey); break;
                case READS_HIFI: dp_column<READS_HIFI>(mv, p, l, &gbest, &gkey); break;
                case READS_RS: dp_column<READS_RS>(mv, p, l, &gbest, &gkey); break;
                default: dp_column<READS_ONT>(mv, p, l, &gbest, &gkey); break;
            }
        }
        if (key_base(gkey) == 0xff) { *err = "no alignment column reaches the end of the window"; return false; }
        // ---- backtrace (generate_cns_from_best_score, ctg_cns.c:1836-1858)
        uint64_t cur = gkey;
        const int min_cov = 4;
        for (;;) {
            const int32_t tp = key_tpos(cur);
            Node* nd = find_node(mv, tp, key_delta(cur) << 8 | key_base(cur));
            if (!nd) { *err = "backtrace left the graph"; return false; }
            const Entry& be = entries[col_off[(size_t)tp] + nd->start + nd->best];
            if (key_base(cur) != 4) {
                ConsBase cb;
                const uint32_t cov = out->stat[(size_t)tp].coverage;
                if (cov == 0) { *err = "zero coverage on the consensus path"; return false; }
                cb.qv = (char)(100 * be.link / cov);
                cb.base = (cov > (uint32_t)min_cov && cb.qv > 20) ? int_to_base(key_base(cur)) : (char)tolower(int_to_base(key_base(cur)));
                cb.pos = (uint32_t)tp;
                out->cons.push_back(cb);
            }
            cur = be.pp;
            if (key_tpos(cur) == -1) break;
        }
        std::reverse(out->cons.begin(), out->cons.end());
        return true;
    }
};

}  // namespace

Exec* make_exec(std::string*) { return new HostExec(); }

}  // namespace np2
