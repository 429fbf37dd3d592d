// Host lockstep executor of the long-read window pipeline -- TEST INFRASTRUCTURE (tests/model): runs the same
// per-lane bodies the HIP kernels compile (nextpolish_amd/csrc/np2_core.h) in plain loops, so the algorithm can be
// validated against the compiled reference on the CPU.  Never linked into the product library.
#include <cstdio>
#include <algorithm>
#include <cstring>

#include "../../nextpolish_amd/csrc/np2_exec.h"
#include "np2_lq_host.h"
#include "../../nextpolish_amd/csrc/np2_lq.h"

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

// link graph of a set of tag streams (update_msa, ctg_cns.c:324-365)
struct HostGraph {
    std::vector<uint32_t> col_cnt, col_off, col_nn;
    std::vector<Entry> entries;
    std::vector<Node> nodes;
    void build(const std::vector<uint64_t>& tag_off, const std::vector<uint32_t>& aln_t_s, const std::vector<uint8_t>& tags,
               uint32_t n_streams, uint32_t n_cols) {
        col_cnt.assign((size_t)n_cols + 1, 0);
        std::vector<LinkObs> obs;
        std::vector<uint32_t> cursor;
        auto walk = [&](bool count_only) {
            for (uint32_t rd = 0; rd < n_streams; ++rd) {
                const uint8_t* tg = tags.data() + tag_off[rd];
                uint32_t d = 0;
                Tag p1{0, 0, 0};
                uint64_t pp = KEY_HEAD, ppp = KEY_HEAD;
                uint32_t pp_base = 0;
                while (next_tag(tg, aln_t_s[rd], &d, &p1)) {
                    const uint64_t key = node_key(p1.t_pos, p1.delta, p1.q_base);
                    if (p1.q_base == 6 || pp_base == 6) { ppp = pp; pp = key; pp_base = p1.q_base; continue; }
                    if (count_only) ++col_cnt[(size_t)p1.t_pos];
                    else {
                        LinkObs& o = obs[cursor[(size_t)p1.t_pos]++];
                        o.pp = pp; o.ppp = ppp; o.rd = rd; o.delta = (uint16_t)p1.delta; o.base = (uint8_t)p1.q_base; o.pad = 0;
                    }
                    ppp = pp; pp = key; pp_base = p1.q_base;
                }
            }
        };
        walk(true);
        col_off.assign((size_t)n_cols + 1, 0);
        for (uint32_t p = 0; p < n_cols; ++p) col_off[(size_t)p + 1] = col_off[p] + col_cnt[p];
        const size_t total = col_off[n_cols];
        obs.resize(total);
        cursor.assign(col_off.begin(), col_off.end());
        walk(false);
        // (streams are walked in order here, so every bucket is already ordered by (rd, delta); the device sorts)
        entries.assign(total, Entry{});
        nodes.assign(total, Node{});
        col_nn.assign(n_cols, 0);
        for (uint32_t p = 0; p < n_cols; ++p)
            col_nn[p] = build_column(obs.data() + col_off[p], col_cnt[p], entries.data() + col_off[p], nodes.data() + col_off[p]);
    }
};

class HostExec : public Exec {
  public:
    bool run_lq_aligned(const LqAlignInput& in, std::string* cons_rev, std::string* err) override {
        LqInput li;
        lq_concatenate_host(in, &li);      // sequential alignment + piece rules (tests/model/np2_lq_host.cpp)
        return run_lq(li, cons_rev, err);
    }
    bool run_lq(const LqInput& in, std::string* cons_rev, std::string* err) override {
        const uint32_t n_streams = (uint32_t)in.t.size();
        const uint32_t n_cols = in.t_len + 1;
        std::vector<ColStat> stat((size_t)n_cols + 64, ColStat{0, 0, 0, 0});   // (slack: see the fill quirk in np2_lq.cpp)
        HostStat hs{stat.data()};
        std::vector<uint64_t> tag_off;
        std::vector<uint32_t> aln_t_s;
        std::vector<uint8_t> tags;
        for (uint32_t i = 0; i < n_streams; ++i) {
            const uint32_t len = (uint32_t)in.t[i].size();
            tag_off.push_back(tags.size());
            tags.resize(tags.size() + (len + 1) / 2 + 1, 0);
            StrColIter f{in.t[i].c_str(), in.q[i].c_str(), 0};
            const uint32_t te = emit_tags_from(f, len, 0u, in.gap_min_len, tags.data() + tag_off.back(), hs);
            if (te > n_cols + 32) { *err = "low-quality concatenation overruns its target length"; return false; }
            aln_t_s.push_back(0);
        }
        HostGraph g;
        g.build(tag_off, aln_t_s, tags, n_streams, n_cols + 32);
        MsaView mv{g.col_off.data(), g.col_nn.data(), g.nodes.data(), g.entries.data(), stat.data()};
        const int32_t len = (int32_t)in.t_len;
        for (int32_t p = 0; p < len; ++p) {
            if (in.hifi) dp_column_lq<true>(mv, p);
            else dp_column_lq<false>(mv, p);
        }
        // start: the last node visited by the reference's loops = (len - 1, max_size - 1, base 5)
        uint64_t cur = node_key(len - 1, (uint32_t)stat[(size_t)len - 1].max_size - 1, 5);
        cons_rev->clear();
        for (;;) {
            const int32_t tp = key_tpos(cur);
            Node* nd = find_node(mv, tp, key_delta(cur) << 8 | key_base(cur));
            if (!nd || nd->len == 0) { *err = "low-quality backtrace left the graph"; return false; }
            const Entry& be = g.entries[g.col_off[(size_t)tp] + nd->start + nd->best];
            if (key_base(cur) != 4) {
                const char up = int_to_base(key_base(cur));
                const uint32_t q = be.link & 0xffffu;
                cons_rev->push_back((q * 5 > stat[(size_t)tp].coverage || up == 'N') ? up : (char)tolower(up));
            }
            cur = be.pp;
            if (key_tpos(cur) == -1) break;
        }
        return true;
    }

    bool compute_spans(const WindowInput& in, int set, std::vector<SpanOut>* spans, std::string*) override {
        const RecordSet& rs = set ? in.sup : in.recs;
        spans->assign(rs.size(), SpanOut{0, 0, 0, 0, 0, 0});
        for (size_t i = 0; i < rs.size(); ++i) {
            ReadView rv{rs.pos[i], rs.n_cigar[i], rs.cigar.data() + rs.cigar_off[i], rs.seq.data() + rs.seq_off[i]};
            uint32_t N, rf_len, rd_len;
            bool bad;
            cigar_totals(rv, &N, &rf_len, &rd_len, &bad);
            if (bad) { (*spans)[i].bad = 1; continue; }
            const AlnSpan a = align_span(rv, in.contig_seq, in.s, in.e, rs.q0[i]);
            (*spans)[i] = SpanOut{a.col0, a.aln_len, a.aln_t_s, a.aln_t_e, a.aln_q_s, 0};
        }
        return true;
    }

    bool run_window(const WindowInput& in, WindowOutput* out, std::string* err) override {
        const int32_t s = in.s, e = in.e, l = e - s;
        out->stat.assign((size_t)l + 1, ColStat{0, 0, 0, 0});
        out->tag_off.clear(); out->aln_t_s.clear(); out->aln_t_e.clear(); out->tags.clear(); out->cons.clear();
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
            add_stream(rv, AlnSpan{0, (uint32_t)l, (uint32_t)s, (uint32_t)e, 0});
        }
        for (const StreamRef& sr : in.streams) {
            const RecordSet& rs = sr.set ? in.sup : in.recs;
            ReadView rv{rs.pos[sr.rec], rs.n_cigar[sr.rec], rs.cigar.data() + rs.cigar_off[sr.rec], rs.seq.data() + rs.seq_off[sr.rec]};
            add_stream(rv, AlnSpan{sr.span.col0, sr.span.aln_len, sr.span.aln_t_s, sr.span.aln_t_e, sr.span.aln_q_s});
        }
        out->seq_count = (uint32_t)out->tag_off.size();
        // ---- link graph (update_msa, ctg_cns.c:324-365)
        HostGraph g;
        g.build(out->tag_off, out->aln_t_s, out->tags, out->seq_count, (uint32_t)l + 1);
        std::vector<uint32_t>& col_off = g.col_off;
        std::vector<Entry>& entries = g.entries;
        // ---- chain DP, column after column
        MsaView mv{g.col_off.data(), g.col_nn.data(), g.nodes.data(), g.entries.data(), out->stat.data()};
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
        // where the low-quality scans have anything to look at (the HIP executor's k2_lq_triggers, restated): NP2_LQ_TRIGGERS=0 leaves
        // the lists empty and the scans test every position themselves, as they did before round 4
        out->trig_del.clear();
        out->trig_ins.clear();
        if (in.lq_ratio1 > 0.f && !(getenv("NP2_LQ_TRIGGERS") && getenv("NP2_LQ_TRIGGERS")[0] == '0') && !out->cons.empty()) {
            const size_t len = out->cons.size();
            out->trig_del.assign((len + 63) / 64, 0);
            out->trig_ins.assign((len + 63) / 64, 0);
            for (size_t i = 0; i < len; ++i) {
                const uint32_t pos = out->cons[i].pos;
                const ColStat c = out->stat[pos];
                if (i >= 1 && !((double)c.l_del < (double)c.coverage * 0.3 && pos < out->cons[i - 1].pos + 20u)) out->trig_del[i >> 6] |= 1ull << (i & 63);
                if (!((float)c.l_ins < (float)c.coverage * in.lq_ratio1)) out->trig_ins[i >> 6] |= 1ull << (i & 63);
            }
        }
        win_tags_ = out->tags;
        win_tag_off_ = out->tag_off;
        win_ts_ = out->aln_t_s;
        return true;
    }

    // plain walk of the stream (the reference's order: every tag from the stream start up to the last region)
    bool run_poa(const PoaBatch& in, std::vector<std::string>* out, std::string* err) override {
        (void)err;
        out->assign(in.job_first.size(), std::string());
        std::vector<std::string> v;
        FILE* dump = getenv("NP2_POA_DUMP") ? fopen(getenv("NP2_POA_DUMP"), "a") : nullptr;      // test hook: the jobs as text, one line each (tab-separated strings)
        for (size_t j = 0; j < in.job_first.size(); ++j) {
            v.clear();
            for (uint32_t k = 0; k < in.job_n[j]; ++k) v.emplace_back(in.chars.data() + in.str_off[in.job_first[j] + k], in.str_len[in.job_first[j] + k]);
            (*out)[j] = poa_consensus(v);
            if (dump) {
                for (size_t k = 0; k < v.size(); ++k) fprintf(dump, "%s%s", k ? "\t" : "", v[k].c_str());
                fprintf(dump, "\n");
            }
        }
        if (dump) fclose(dump);
        return true;
    }
    bool extract(const std::vector<SubReq>& req, std::vector<uint32_t>* off, std::string* bases, std::string* err) override {
        off->assign(req.size() + 1, 0);
        bases->clear();
        std::vector<Tag> at;
        size_t i = 0;
        while (i < req.size()) {
            const uint32_t st = req[i].stream;
            if (st >= win_tag_off_.size()) { *err = "extract: bad stream"; return false; }
            size_t k = i;
            uint32_t last_end = 0;
            for (; k < req.size() && req[k].stream == st; ++k) last_end = std::max(last_end, req[k].end);
            at.clear();
            Tag tag{0, 0, 0};
            uint32_t p = 0;
            const uint8_t* tg = win_tags_.data() + win_tag_off_[st];
            while (next_tag(tg, win_ts_[st], &p, &tag)) {
                if ((uint32_t)tag.t_pos > last_end) break;
                at.push_back(tag);
            }
            for (; i < k; ++i) {
                for (const Tag& t : at)
                    if ((uint32_t)t.t_pos >= req[i].start && (uint32_t)t.t_pos <= req[i].end && t.q_base != 4) bases->push_back(int_to_base(t.q_base));
                (*off)[i + 1] = (uint32_t)bases->size();
            }
        }
        return true;
    }

    // the host walk the reference does (ctg_cns.c:2898-2971, get_qpos-style loops over the stream's tags)
    bool read_coords(const std::vector<CoordReq>& req, std::vector<uint32_t>* bases, std::string* err) override {
        bases->assign(req.size(), 0);
        for (size_t i = 0; i < req.size(); ++i) {
            const uint32_t st = req[i].stream;
            if (st >= win_tag_off_.size()) { *err = "read_coords: bad stream"; return false; }
            Tag tag{0, 0, 0};
            uint32_t p = 0, q = 0;
            const uint8_t* tg = win_tags_.data() + win_tag_off_[st];
            while (next_tag(tg, win_ts_[st], &p, &tag)) {
                if (req[i].through_col) { if (tag.q_base != 4) ++q; if ((uint32_t)tag.t_pos == req[i].col) break; }
                else { if ((uint32_t)tag.t_pos == req[i].col) break; if (tag.q_base != 4) ++q; }
            }
            (*bases)[i] = q;
        }
        return true;
    }

  private:
    std::vector<uint8_t> win_tags_;
    std::vector<uint64_t> win_tag_off_;
    std::vector<uint32_t> win_ts_;
};

}  // namespace

Exec* make_exec(std::string*) { return new HostExec(); }

}  // namespace np2
