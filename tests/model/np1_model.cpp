// tests/model/np1_model.cpp -- TEST INFRASTRUCTURE, never loaded by the product.
//
// Host-side lockstep model of the HIP launch sequence in nextpolish_amd/csrc/np1_device.hip:
// the same per-lane bodies (np1_core.h, compiled for the host) driven by plain loops, with the
// wave-cooperative k_vote re-stated over 64-entry lane arrays.  It lets the CPU test-suite check
// the *staged algorithm* (slot space, symbol rows, single-state shortcut, run-wise exact DP,
// emission) against the oracle without a GPU; the real kernels are checked on the GPU (-m gpu).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/nextpolish1.h"
#include "../../nextpolish_amd/csrc/np1_core.h"
#include "../../nextpolish_amd/csrc/np1_desc.h"
#include "../../nextpolish_amd/csrc/np1_kmer.h"
#include "../../nextpolish_amd/csrc/np1_tile9.h"
#include "../../nextpolish_amd/csrc/np1_replay.h"
#include "../../nextpolish_amd/csrc/np1_upload.h"
#include "../../nextpolish_amd/csrc/np_bam.h"
#include "../../nextpolish_amd/csrc/np_stream.h"

using namespace np1k;

extern "C" int np1m_fused;
namespace {
struct HostState {
    long long sc_[2][16];
    uint16_t km_[2][16];
    uint8_t rk_[2][16];
    long long& sc(int b, uint32_t i) { return sc_[b][i]; }
    uint16_t& km(int b, uint32_t i) { return km_[b][i]; }
    uint8_t& rk(int b, uint32_t i) { return rk_[b][i]; }
};

template <int E>
bool vote_chunk(uint32_t c, const std::vector<uint4>& meta, const std::vector<uint8_t>& rows,
                const std::vector<uint8_t>& slot_info, uint32_t S, const std::vector<uint32_t>& chunk_first,
                const std::vector<uint32_t>& chunk_last, std::vector<uint16_t>& slot_res, std::vector<uint32_t>& slot_rec,
                std::vector<uint32_t>& pool, std::vector<uint32_t>& heads, uint32_t flag_single) {
    std::vector<uint32_t> L((E - 2) * 64);
    VoteLane<E> vl[64];
    uint32_t info[64], dsym[64], basemask[64], sym[64];
    bool valid[64], first[64];
    uint32_t s[64];
    for (int l = 0; l < 64; ++l) {
        int64_t s64 = (int64_t)c * VOTE_CH - 2 + l;
        valid[l] = s64 >= 0 && s64 < (int64_t)S;
        s[l] = (uint32_t)s64;
        info[l] = valid[l] ? slot_info[s[l]] : 0u;
        dsym[l] = info[l] & 0xf;
        first[l] = (info[l] & SI_FIRST) != 0;
    }
    uint32_t prev_dsym[64];
    for (int l = 0; l < 64; ++l) {
        // __shfl_up returns the lane's own value when there is no source lane
        uint32_t d1 = l >= 1 ? dsym[l - 1] : dsym[l], d2 = l >= 2 ? dsym[l - 2] : dsym[l];
        uint32_t f1 = l >= 1 ? (uint32_t)first[l - 1] : (uint32_t)first[l];
        prev_dsym[l] = d1;
        if (first[l]) { d1 = 0; d2 = 0; }
        else if (f1) d2 = 0;
        vl[l].init(d2 << 8 | d1 << 4 | dsym[l]);
        basemask[l] = 1u << dsym[l];
    }
    uint32_t r0 = chunk_first[c], r1 = chunk_last[c];
    if (r0 != 0xffffffffu)
        for (uint32_t r = r0; r <= r1; ++r) {
            const uint4 m = meta[r];
            bool cov[64];
            for (int l = 0; l < 64; ++l) {
                cov[l] = valid[l] && s[l] >= m.x && s[l] <= m.y;
                sym[l] = 0;
                if (cov[l]) {
                    uint32_t nn = s[l] - m.z;
                    uint32_t byte = rows[(uint64_t)m.w * 4 + (nn >> 1)];
                    sym[l] = (byte >> ((nn & 1) * 4)) & 0xf;
                }
            }
            for (int l = 0; l < 64; ++l) {
                uint32_t p1 = l >= 1 ? sym[l - 1] : 0, p2 = l >= 2 ? sym[l - 2] : 0;
                if (cov[l]) {
                    basemask[l] |= 1u << sym[l];
                    if (l >= 2) vl[l].tally(p2 << 8 | p1 << 4 | sym[l], L.data(), l);
                }
            }
        }
    for (int l = 0; l < 64; ++l)
        if (vl[l].ovf) return false;
    bool single[64];
    for (int l = 0; l < 64; ++l) single[l] = __builtin_popcount(basemask[l]) == 1;
    for (int l = 2; l < 64; ++l) {
        if (!valid[l]) continue;
        uint32_t total = vl[l].total(L.data(), l);
        bool prev_is_single = first[l] || single[l - 1];
        bool is_head = !single[l] && prev_is_single;
        bool need_rec = !single[l] || !prev_is_single;
        uint32_t res = 0xffu;
        if (single[l]) res = dsym[l] | (((total == 1 ? 1u : 0u) | flag_single) << 8);
        slot_res[s[l]] = (uint16_t)res;
        uint32_t my_off = 0xffffffffu;
        if (need_rec) {
            my_off = (uint32_t)pool.size();
            pool.resize(pool.size() + vl[l].n + REC_FIXED_WORDS, 0xdeadbeefu);
            uint32_t hdr = (single[l] ? REC_SINGLE : 0u) | ((info[l] & SI_LAST) ? REC_CTG_LAST : 0u) |
                           (first[l] ? REC_CTG_FIRST : 0u) | (prev_dsym[l] << 4);
            vl[l].write_record(pool.data() + my_off, s[l], total, hdr, L.data(), l);
        }
        slot_rec[s[l]] = my_off;
        if (is_head) heads.push_back(my_off);
    }
    return true;
}
// fused sequence (k_desc + k_tile3): lanes evaluate record descriptors instead of reading symbol rows
template <int E>
bool vote_chunk_desc(uint32_t c, const std::vector<uint32_t>& desc, const std::vector<uint32_t>& ovf, const ReadsDev& R, const std::vector<uint32_t>& soff,
                     const std::vector<uint8_t>& slot_info, const std::vector<uint32_t>& slot_g, uint32_t S,
                     const std::vector<uint32_t>& chunk_first, const std::vector<uint32_t>& chunk_last,
                     std::vector<uint16_t>& slot_res, std::vector<uint32_t>& slot_rec, std::vector<uint32_t>& pool,
                     std::vector<uint32_t>& heads, uint32_t flag_single) {
    std::vector<uint32_t> L((E - 2) * 64);
    VoteLane<E> vl[64];
    uint32_t info[64], dsym[64], basemask[64], sym[64], g[64], s[64], prev_dsym[64], d1eff[64];
    int32_t jj[64];
    bool valid[64], first[64];
    for (int l = 0; l < 64; ++l) {
        int64_t s64 = (int64_t)c * VOTE_CH - 2 + l;
        valid[l] = s64 >= 0 && s64 < (int64_t)S;
        s[l] = (uint32_t)s64;
        info[l] = valid[l] ? slot_info[s[l]] : 0u;
        g[l] = valid[l] ? slot_g[s[l]] : 0u;
        jj[l] = (valid[l] && (info[l] & SI_INSERT)) ? (int32_t)(s[l] - soff[g[l]]) - 1 : -1;
        dsym[l] = info[l] & 0xf;
        first[l] = (info[l] & SI_FIRST) != 0;
    }
    for (int l = 0; l < 64; ++l) {   // wave_shr1: lane 0 receives 0
        uint32_t d1 = l >= 1 ? dsym[l - 1] : 0, d2 = l >= 2 ? dsym[l - 2] : 0;
        uint32_t f1 = l >= 1 ? (uint32_t)first[l - 1] : 0;
        prev_dsym[l] = d1;
        if (first[l]) { d1 = 0; d2 = 0; }
        else if (f1) d2 = 0;
        vl[l].init(d2 << 8 | d1 << 4 | dsym[l]);
        d1eff[l] = d1;
        basemask[l] = 1u << dsym[l];
    }
    uint32_t r0 = chunk_first[c], r1 = chunk_last[c];
    if (r0 != 0xffffffffu)
        for (uint32_t r = r0; r <= r1; ++r) {
            const uint32_t* d = desc.data() + (uint64_t)r * DESC_WORDS;
            for (int l = 0; l < 64; ++l) sym[l] = 0;   // rsym: kept across the parts of a chained record
            for (;;) {
                bool cov[64];
                for (int l = 0; l < 64; ++l) {
                    cov[l] = valid[l] && s[l] >= d[0] && s[l] <= d[1];
                    if (cov[l]) sym[l] = desc_symbol(d, g[l], jj[l], SeqBytes{R.seq + R.seq_off[r]});
                }
                for (int l = 0; l < 64; ++l) {
                    uint32_t p1 = l >= 1 ? sym[l - 1] : 0, p2 = l >= 2 ? sym[l - 2] : 0;
                    if (cov[l]) {
                        basemask[l] |= 1u << sym[l];
                        if (l >= 2) vl[l].tally(p2 << 8 | p1 << 4 | sym[l], L.data(), l);
                    }
                }
                if (d[DESC_NEXT] == 0) break;
                d = ovf.data() + (uint64_t)(d[DESC_NEXT] - 1) * DESC_WORDS;
            }
        }
    for (int l = 0; l < 64; ++l)
        if (vl[l].ovf) return false;
    bool single[64];
    for (int l = 0; l < 64; ++l) single[l] = __builtin_popcount(basemask[l]) == 1;
    for (int l = 2; l < 64; ++l) {
        if (!valid[l]) continue;
        uint32_t total = vl[l].total(L.data(), l);
        bool prev_is_single = first[l] || single[l - 1];
        bool is_head = !single[l] && prev_is_single;
        bool need_rec = !single[l] || !prev_is_single;
        uint32_t res = 0xffu;
        if (single[l]) res = dsym[l] | (((total == 1 ? 1u : 0u) | flag_single) << 8);
        slot_res[s[l]] = (uint16_t)res;
        uint32_t my_off = 0xffffffffu;
        if (need_rec) {
            my_off = (uint32_t)pool.size();
            pool.resize(pool.size() + vl[l].n + REC_FIXED_WORDS, 0xdeadbeefu);
            uint32_t hdr = (single[l] ? REC_SINGLE : 0u) | ((info[l] & SI_LAST) ? REC_CTG_LAST : 0u) |
                           (first[l] ? REC_CTG_FIRST : 0u) | (prev_dsym[l] << 4);
            vl[l].write_record(pool.data() + my_off, s[l], total, hdr, L.data(), l);
        }
        slot_rec[s[l]] = my_off;
        if (is_head) heads.push_back(my_off);
    }
    return true;
}
// k_tile9 (np1m_fused == 2): four slots per lane, agreeing records counted per window, everything else deferred (np1_tile9.h).
// One call = one wave = T9_CH vote chunks, in the kernel's own order of events: record loop -> the wave's dense list of deferred
// (record, lane) pairs -> evaluation -> tally (a lane's own chain, or, for a lane with many entries, the list in batches of 64 with
// distinct contexts and counts per batch).  Returns false when the list or a context list overflows (the device redoes the chunks with k_tile3).
template <int E>
bool vote_tile9(uint32_t t, const std::vector<uint32_t>& desc, const std::vector<uint32_t>& ovf, const ReadsDev& R, const uint8_t* seq_padded,
                const std::vector<uint32_t>& soff, const std::vector<uint8_t>& slot_info, const std::vector<uint32_t>& slot_g, uint32_t S,
                const std::vector<uint32_t>& chunk_first, const std::vector<uint32_t>& chunk_last, uint32_t n_chunks, std::vector<uint16_t>& slot_res,
                std::vector<uint32_t>& slot_rec, std::vector<uint32_t>& pool, std::vector<uint32_t>& heads, uint32_t flag_single, uint64_t* n_agree, uint64_t* n_entries,
                uint64_t* n_hot) {
    const uint32_t tile_s0 = t * T9_SLOTS;
    T9Win w[64];
    uint32_t info[64][6];
    int32_t jj[64][6];
    VoteLane<E> vl[64][4];
    uint32_t basemask[64][4], c_all[64], my_n[64];
    std::vector<uint32_t> L(4 * (E - 2) * 64);
    struct Ent { uint32_t x; int owner; };
    std::vector<Ent> dl;
    for (int l = 0; l < 64; ++l) {
        t9_window(tile_s0, l, S, slot_info.data(), slot_g.data(), &w[l], info[l]);
        c_all[l] = my_n[l] = 0;
        for (int p = 0; p < 6; ++p) jj[l][p] = ((w[l].imask >> p) & 1u) ? (int32_t)(w[l].s0 - 2u + (uint32_t)p - soff[w[l].g[p]]) - 1 : -1;
        for (int j = 0; j < 4; ++j) {
            const int p = j + 2;
            uint32_t d0 = info[l][p] & 0xf, d1 = info[l][p - 1] & 0xf, d2 = info[l][p - 2] & 0xf;
            if (info[l][p] & SI_FIRST) { d1 = 0; d2 = 0; }
            else if (info[l][p - 1] & SI_FIRST) d2 = 0;
            vl[l][j].init(d2 << 8 | d1 << 4 | d0);
            basemask[l][j] = 1u << d0;
        }
    }
    uint32_t r0 = 0xffffffffu, r1 = 0;
    for (uint32_t c = t * T9_CH; c < t * T9_CH + T9_CH && c < n_chunks; ++c)
        if (chunk_first[c] != 0xffffffffu) { r0 = std::min(r0, chunk_first[c]); r1 = std::max(r1, chunk_last[c]); }
    if (r0 != 0xffffffffu)
        for (uint32_t r = r0; r <= r1; ++r) {
            const T9Rec rec = t9_rec(desc.data() + (uint64_t)r * DESC_WORDS);
            for (int l = 0; l < 64; ++l) {
                const int k = t9_step(rec, seq_padded + R.seq_off[r], w[l]);
                if (k == T9_AGREE) { ++c_all[l]; ++*n_agree; }
                else if (k == T9_ENTRY) {
                    if (dl.size() >= T9_DL) return false;      // the wave's deferred list is full: its chunks go to k_tile3
                    dl.push_back(Ent{r, l});
                    ++my_n[l];
                    ++*n_entries;
                }
            }
        }
    for (Ent& e : dl) {
        const uint32_t r = e.x;
        e.x = t9_code(desc.data() + (uint64_t)r * DESC_WORDS, ovf.data(), seq_padded + R.seq_off[r], w[e.owner].s0, w[e.owner].g, jj[e.owner]);
    }
    for (int l = 0; l < 64; ++l) {
        if (my_n[l] <= T9_HOT) {
            for (const Ent& e : dl)
                if (e.owner == l) t9_tally<E>(e.x, vl[l], basemask[l], L.data(), l);
            continue;
        }
        ++*n_hot;
        for (size_t base = 0; base < dl.size(); base += 64) {       // the whole wave on one lane's entries, 64 list entries at a time
            const size_t end = std::min(dl.size(), base + 64);
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t p = j + 2;
                std::vector<uint8_t> left(end - base, 0);
                for (size_t x = base; x < end; ++x) {
                    const uint32_t lo = (dl[x].x >> 24) & 7u, hi = (dl[x].x >> 27) & 7u;
                    left[x - base] = dl[x].owner == l && p >= lo && p <= hi;
                }
                for (size_t x = base; x < end; ++x) {
                    if (!left[x - base]) continue;
                    const uint32_t k = (dl[x].x >> (20 - 4 * p)) & 0xfffu;
                    uint32_t cnt = 0;
                    for (size_t y = x; y < end; ++y)
                        if (left[y - base] && ((dl[y].x >> (20 - 4 * p)) & 0xfffu) == k) { ++cnt; left[y - base] = 0; }
                    t9_tally_ctx<E>(k, cnt, true, vl[l][j], basemask[l][j], L.data() + j * (E - 2) * 64, l);
                }
            }
        }
    }
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            vl[l][j].c0 += c_all[l];
            if (vl[l][j].ovf) return false;
        }
    for (int l = 1; l <= 62; ++l)
        for (int j = 0; j < 4; ++j) {
            const uint32_t s = w[l].s0 + (uint32_t)j;
            if (!w[l].active || s >= S) continue;
            const uint32_t inf = info[l][j + 2], dsym = inf & 0xf;
            const bool first = (inf & SI_FIRST) != 0;
            const bool single = __builtin_popcount(basemask[l][j]) == 1;
            const bool psingle = j > 0 ? __builtin_popcount(basemask[l][j - 1]) == 1 : __builtin_popcount(basemask[l - 1][3]) == 1;
            const uint32_t* Lj = L.data() + j * (E - 2) * 64;
            const uint32_t total = vl[l][j].total(Lj, l);
            const bool prev_is_single = first || psingle;
            const bool is_head = !single && prev_is_single;
            const bool need_rec = !single || !prev_is_single;
            uint32_t res = 0xffu;
            if (single) res = dsym | (((total == 1 ? 1u : 0u) | flag_single) << 8);
            slot_res[s] = (uint16_t)res;
            uint32_t my_off = 0xffffffffu;
            if (need_rec) {
                my_off = (uint32_t)pool.size();
                pool.resize(pool.size() + vl[l][j].n + REC_FIXED_WORDS, 0xdeadbeefu);
                const uint32_t hdr = (single ? REC_SINGLE : 0u) | ((inf & SI_LAST) ? REC_CTG_LAST : 0u) | (first ? REC_CTG_FIRST : 0u) | ((info[l][j + 1] & 0xf) << 4);
                vl[l][j].write_record(pool.data() + my_off, s, total, hdr, Lj, l);
            }
            slot_rec[s] = my_off;
            if (is_head) heads.push_back(my_off);
        }
    return true;
}
}  // namespace

extern "C" {
int np1m_fused = 0;   // 0: staged sequence (rows in memory), 1: descriptors (k_tile3), 2: four slots per lane with deferred entries (k_tile9; k_tile3 for what it hands back)
unsigned long long np1m_t9_stats[4] = {0, 0, 0, 0};   // last call in mode 2: agreeing (record, window) pairs, deferred entries, lanes tallied by the whole wave, waves handed back


// Returns 0 on success; *out is malloc'd concatenation of the polished contigs, bounds[n_contigs+1].
// stats (optional, 4 words): slots, dp heads, pool words, max context list length escalations
static int score_chain_once(const np1_stream_view* v, const Configure* cfg, char** out, uint32_t* bounds, uint64_t* stats);
static bool g_keep_map = false;                       // score_chain_once leaves its slot geometry behind (tiling model below)
static std::vector<uint32_t> g_soff, g_opos;          // slot of every draft base; output offset of every slot
static std::vector<uint8_t> g_single;                 // slot had one state after the vote
int np1m_restarts = 0, np1m_deep_chunks = 0;   // what the last call needed: staged restarts, chunks voted with HBM-sized context lists
int np1m_score_chain(const np1_stream_view* v, const Configure* cfg, char** out, uint32_t* bounds, uint64_t* stats) {
    np1m_restarts = 0;
    np1m_deep_chunks = 0;
    int rc = score_chain_once(v, cfg, out, bounds, stats);
    if ((rc == -5 || rc == -3) && np1m_fused) {   // np1_device.hip: force_staged
        const int keep = np1m_fused;
        np1m_fused = 0;
        ++np1m_restarts;
        rc = score_chain_once(v, cfg, out, bounds, stats);
        np1m_fused = keep;
    }
    return rc;
}
static int score_chain_once(const np1_stream_view* v, const Configure* cfg, char** out, uint32_t* bounds, uint64_t* stats) {
    const uint32_t nc = (uint32_t)v->n_contigs;
    const int64_t n = v->n_reads;
    const uint64_t G = (uint64_t)v->draft_len;
    ReadsDev R{v->pos, v->ctg, v->flag, v->n_cigar, v->l_qseq, v->cigar_off, v->seq_off, v->cigar, v->seq};
    int K = -1;
    long long Rfix = 0;
    for (int k = 0; k <= 10; ++k) {
        double x = cfg->indel_balance_factor_sgs * (double)(1 << k);
        if (x == (double)(long long)x) { K = k; Rfix = (long long)x; break; }
    }
    if (K < 0) return -2;
    uint32_t flag_single = (1.0 < cfg->min_count_ratio_skip) ? 2u : 0u;
    std::vector<int32_t> qs(n), qe(n), span(n);
    std::vector<uint32_t> ins(G + 1, 0), counters(CNT_WORDS, 0);
    for (int64_t r = 0; r < n; ++r)
        prep_record(R, r, v->ctg_off, cfg->trim_len_edge, qs.data(), qe.data(), span.data(), ins.data(), counters.data());
    if (counters[CNT_ERR]) return (int)counters[CNT_ERR];
    std::vector<uint32_t> soff(G + 2);
    uint64_t acc = 0;
    for (uint64_t g = 0; g < G; ++g) { soff[g] = (uint32_t)acc; acc += 1 + ins[g]; }
    soff[G] = (uint32_t)acc;
    const uint32_t S = (uint32_t)acc;
    std::vector<uint8_t> slot_info(S + 64, 0);
    std::vector<uint32_t> slot_g(S + 64, 0);
    for (uint32_t c = 0; c < nc; ++c)
        for (uint32_t g = v->ctg_off[c]; g < v->ctg_off[c + 1]; ++g)
            slotinfo_base((const uint8_t*)v->draft, g, v->ctg_off[c], v->ctg_off[c + 1], soff.data(), slot_info.data(), slot_g.data());
    const uint32_t n_chunks = (S + VOTE_CH - 1) / VOTE_CH + 1;
    std::vector<uint32_t> chunk_first(n_chunks, 0xffffffffu), chunk_last(n_chunks, 0);
    std::vector<uint16_t> slot_res(S + 64, 0xffff);
    std::vector<uint32_t> slot_rec(S + 64, 0xffffffffu), pool, heads;
    uint64_t escal = 0;
    int& deep = np1m_deep_chunks;
    if (np1m_fused) {
        std::vector<uint32_t> desc((size_t)(n ? n : 1) * DESC_WORDS, 0xdeadbeefu);
        const uint32_t ovf_cap = (uint32_t)(v->cigar_len + 16);
        std::vector<uint32_t> ovf((size_t)ovf_cap * DESC_WORDS, 0xdeadbeefu);
        for (int64_t r = 0; r < n; ++r) {
            uint32_t c0, c1;
            desc_record(R, r, v->ctg_off, soff.data(), qs.data(), qe.data(), desc.data(), ovf.data(), ovf_cap, counters.data(), &c0, &c1);
            if (counters[CNT_ERR] & ERR_DESC_OVERFLOW) return -5;
            for (uint32_t cc = c0; cc <= c1 && c0 <= c1; ++cc) {
                if ((uint32_t)r < chunk_first[cc]) chunk_first[cc] = (uint32_t)r;
                if ((uint32_t)r > chunk_last[cc]) chunk_last[cc] = (uint32_t)r;
            }
        }
        if (stats) stats[3] = counters[CNT_OVFDESC];
        std::vector<uint8_t> seq_padded((size_t)v->seq_len + 16, 0);
        if (v->seq_len) memcpy(seq_padded.data(), v->seq, (size_t)v->seq_len);
        std::vector<uint8_t> redo_chunk(n_chunks, np1m_fused == 1 ? 1 : 0);
        if (np1m_fused == 2) {
            uint64_t na = 0, ne = 0, ng = 0, nr = 0;
            for (uint32_t t = 0; t * T9_CH < n_chunks; ++t) {
                const size_t pool_at = pool.size(), heads_at = heads.size();
                if (vote_tile9<8>(t, desc, ovf, R, seq_padded.data(), soff, slot_info, slot_g, S, chunk_first, chunk_last, n_chunks, slot_res, slot_rec, pool, heads,
                                  flag_single, &na, &ne, &ng))
                    continue;
                pool.resize(pool_at);          // (the device writes nothing for a wave it hands back)
                heads.resize(heads_at);
                ++nr;
                for (uint32_t c = t * T9_CH; c < t * T9_CH + T9_CH && c < n_chunks; ++c) redo_chunk[c] = 1;
            }
            np1m_t9_stats[0] = na; np1m_t9_stats[1] = ne; np1m_t9_stats[2] = ng; np1m_t9_stats[3] = nr;
        }
        for (uint32_t c = 0; c < n_chunks; ++c) {
            if (!redo_chunk[c]) continue;
            if (vote_chunk_desc<8>(c, desc, ovf, R, soff, slot_info, slot_g, S, chunk_first, chunk_last, slot_res, slot_rec, pool, heads, flag_single)) continue;
            ++escal;
            if (vote_chunk_desc<64>(c, desc, ovf, R, soff, slot_info, slot_g, S, chunk_first, chunk_last, slot_res, slot_rec, pool, heads, flag_single)) continue;
            if (!vote_chunk_desc<160>(c, desc, ovf, R, soff, slot_info, slot_g, S, chunk_first, chunk_last, slot_res, slot_rec, pool, heads, flag_single)) return -3;
        }
    } else {
        std::vector<uint32_t> rbase(n), capb(n);
        std::vector<uint64_t> rowoff(n + 1);
        uint64_t rb = 0;
        for (int64_t r = 0; r < n; ++r) {
            rowcap_record(R, r, v->ctg_off, soff.data(), qs.data(), qe.data(), span.data(), rbase.data(), capb.data());
            rowoff[r] = rb;
            rb += capb[r];
        }
        rowoff[n] = rb;
        std::vector<uint8_t> rows(rb + 64, 0xEE);
        std::vector<uint4> meta(n ? n : 1);
        for (int64_t r = 0; r < n; ++r)
            rows_record(R, r, v->ctg_off, soff.data(), qs.data(), qe.data(), rbase.data(), rowoff.data(), rows.data(), meta.data(),
                        chunk_first.data(), chunk_last.data());
        for (uint32_t c = 0; c < n_chunks; ++c) {
            if (vote_chunk<16>(c, meta, rows, slot_info, S, chunk_first, chunk_last, slot_res, slot_rec, pool, heads, flag_single)) continue;
            ++escal;
            if (vote_chunk<64>(c, meta, rows, slot_info, S, chunk_first, chunk_last, slot_res, slot_rec, pool, heads, flag_single)) continue;
            if (vote_chunk<160>(c, meta, rows, slot_info, S, chunk_first, chunk_last, slot_res, slot_rec, pool, heads, flag_single)) continue;
            ++deep;   // every possible context in its own entry (the device keeps these lists in HBM)
            if (!vote_chunk<VOTE_E_ALL>(c, meta, rows, slot_info, S, chunk_first, chunk_last, slot_res, slot_rec, pool, heads, flag_single)) return -3;
        }

    }
    if (g_keep_map) {   // tiling model: which slots came out of the vote with a single state (the chain restarts behind each of them)
        g_single.assign(S + 1, 0);
        for (uint32_t s = 0; s < S; ++s) g_single[s] = (slot_res[s] & 0xff) != 0xff;
    }
    HostState st;
    for (uint32_t h : heads)
        if (!dp_run<false>(h, pool.data(), slot_rec.data(), slot_res.data(), K, Rfix, 0.0, cfg->min_count_ratio_skip, st)) return -4;
    for (uint32_t c = 0; c < nc; ++c) fixfirst_contig(v->ctg_off[c], v->ctg_off[c + 1], soff.data(), slot_info.data(), slot_res.data());
    std::vector<uint32_t> opos(S + 1);
    uint32_t o = 0;
    for (uint32_t s = 0; s < S; ++s) { opos[s] = o; o += (slot_res[s] & 0xff) != 3; }
    opos[S] = o;
    if (g_keep_map) { g_soff.assign(soff.begin(), soff.begin() + G + 1); g_opos = opos; }
    char* buf = (char*)calloc(1, (size_t)o + 1);
    for (uint32_t s = 0; s < S; ++s) emit_slot(s, slot_res.data(), slot_info.data(), opos.data(), 3u, (uint8_t*)buf);
    for (uint32_t c = 0; c <= nc; ++c) bounds[c] = opos[soff[v->ctg_off[c]]];
    *out = buf;
    if (stats) { stats[0] = S; stats[1] = heads.size(); stats[2] = pool.size(); if (!np1m_fused) stats[3] = escal; }
    return 0;
}

// diagnostics: vote rounds the max_count_kmer break made necessary (replay of the region iterator) since the library was loaded
unsigned long long np1m_replay_revotes = 0, np1m_replay_breaks = 0;
// kmer_count through the per-region bodies of np1_kmer.h, driven sequentially (the GPU runs one lane per region)
// what the iterator replay needs besides the stream (np1_replay.h): the BAI, the BAM tid of every contig, the records' virtual offsets
struct ModelGeometry { const np::BaiIndex* bai; const int32_t* tid; const uint64_t* voff; const uint64_t* voff_end; };

static int kmer_model(const np1_stream_view* v, const Configure* cfg, char** out, uint32_t* bounds, bool snp_valid, const ModelGeometry* geo = nullptr) {
    const uint32_t nc = (uint32_t)v->n_contigs;
    const int64_t n = v->n_reads;
    const uint64_t G = (uint64_t)v->draft_len;
    if (v->qual_len == 0 && n > 0) return -10;
    KcCtx c;
    memset(&c, 0, sizeof(c));
    c.R = ReadsDev{v->pos, v->ctg, v->flag, v->n_cigar, v->l_qseq, v->cigar_off, v->seq_off, v->cigar, v->seq};
    c.mapq = v->mapq; c.isize = v->isize; c.qual_off = v->qual_off; c.qual = v->qual;
    c.ctg_off = v->ctg_off; c.read_begin = v->read_begin;
    c.trim = cfg->trim_len_edge; c.ext_len_edge = cfg->ext_len_edge; c.min_len_ldr = cfg->min_len_ldr;
    c.min_len_inter_kmer = cfg->min_len_inter_kmer; c.max_len_kmer = cfg->max_len_kmer; c.max_count_kmer = cfg->max_count_kmer;
    c.min_map_quality = cfg->min_map_quality; c.read_tlen = cfg->read_tlen;
    c.max_clip_ratio_sgs = cfg->max_clip_ratio_sgs; c.min_count_ratio_skip = cfg->min_count_ratio_skip;
    c.K = -1;
    for (int k = 0; k <= 10; ++k) {
        double x = cfg->indel_balance_factor_sgs * (double)(1 << k);
        if (x == (double)(long long)x) { c.K = k; c.Rfix = (long long)x; break; }
    }
    c.rate = cfg->indel_balance_factor_sgs;   // K < 0: general rate, doubles
    uint32_t err = 0;
    c.err = &err;
    std::vector<uint8_t> level(n ? n : 1);
    std::vector<int32_t> endpos(n ? n : 1);
    int32_t max_span = 1;
    for (int64_t r = 0; r < n; ++r) {
        level[r] = (uint8_t)kc_filter_level(c.R, r, c.mapq, c.isize, c.read_tlen, c.max_clip_ratio_sgs, c.min_map_quality);
        endpos[r] = kc_endpos(c.R, r);
        if (endpos[r] - v->pos[r] > max_span) max_span = endpos[r] - v->pos[r];
    }
    c.level = level.data(); c.endpos = endpos.data(); c.max_span = max_span;
    std::vector<uint8_t> dcode(G + 1), dflag(G + 1);
    for (uint64_t g = 0; g < G; ++g) {
        uint32_t ch = (uint8_t)v->draft[g];
        dflag[g] = 0;
        if (ch >= 97 && ch <= 122) { ch -= 32; dflag[g] = KC_FLAG_ZERO; }
        dcode[g] = (uint8_t)draft_code(ch);
    }
    c.draft_code = dcode.data(); c.draft_flag = dflag.data();
    // regions per contig
    std::vector<std::vector<int32_t>> nodepth(nc), kreg(nc);
    std::vector<uint32_t> ins(G + 1, 0);
    for (uint32_t ct = 0; ct < nc; ++ct) {
        const uint32_t g0 = v->ctg_off[ct];
        const int32_t L = (int32_t)(v->ctg_off[ct + 1] - g0);
        if (L <= 0) continue;
        std::vector<uint32_t> fl;
        for (int32_t i = 0; i < L; ++i) if (dflag[g0 + i]) fl.push_back((uint32_t)i);
        std::vector<int32_t> buf(2 * fl.size() + 4);
        int32_t k = kc_find_regions(dcode.data() + g0, dflag.data() + g0, L, fl.data(), (uint32_t)fl.size(), 0,
                                    (uint32_t)c.min_len_ldr, c.ext_len_edge, false, buf.data(), (int32_t)buf.size());
        if (k < 0) return -11;
        k = kc_merge_regions(buf.data(), k);
        nodepth[ct].assign(buf.begin(), buf.begin() + k);
        k = kc_find_regions(dcode.data() + g0, dflag.data() + g0, L, fl.data(), (uint32_t)fl.size(), (uint32_t)c.min_len_inter_kmer, 0,
                            c.ext_len_edge, true, buf.data(), (int32_t)buf.size());
        if (k < 0) return -11;
        k = kc_merge_regions(buf.data(), k);
        kreg[ct].assign(buf.begin(), buf.begin() + k);
        for (size_t i = 0; i + 1 < kreg[ct].size(); i += 2) kc_insert_region(c, ct, kreg[ct][i], kreg[ct][i + 1], ins.data());
        if (snp_valid) nodepth[ct].clear();     // task 4 has no no-depth regions (snpvalid.c:3-36)
        for (size_t i = 0; i + 1 < nodepth[ct].size(); i += 2) kc_insert_region(c, ct, nodepth[ct][i], nodepth[ct][i + 1], ins.data());
    }
    std::vector<uint32_t> soff(G + 2);
    uint64_t acc = 0;
    for (uint64_t g = 0; g < G; ++g) { soff[g] = (uint32_t)acc; acc += 1 + ins[g]; }
    soff[G] = (uint32_t)acc;
    const uint32_t S = (uint32_t)acc;
    std::vector<uint8_t> slot_info(S + 64, 0), sbase(S + 64), sflag(S + 64);
    for (uint32_t ct = 0; ct < nc; ++ct)
        for (uint32_t g = v->ctg_off[ct]; g < v->ctg_off[ct + 1]; ++g)
            slotinfo_base((const uint8_t*)v->draft, g, v->ctg_off[ct], v->ctg_off[ct + 1], soff.data(), slot_info.data());
    for (uint32_t s = 0; s < S; ++s) { sbase[s] = slot_info[s] & 0xf; sflag[s] = (slot_info[s] & SI_LOWER) ? 1 : 0; }
    std::vector<uint16_t> srefk(S + 64, 0), scount(S + 64, 0);
    std::vector<uint32_t> lhead(S + 64, 0), lpool(2ull * (1u << 22));
    uint32_t lcount = 0, stcount = 0, hcount = 0;
    const uint32_t stcap = 1u << 20;
    std::vector<long long> stsc(16ull * stcap);
    std::vector<uint16_t> stkm(16ull * stcap);
    std::vector<uint8_t> strk(16ull * stcap);
    std::vector<uint8_t> hpool(64u << 20);
    c.soff = soff.data(); c.sbase = sbase.data(); c.sflag = sflag.data(); c.srefk = srefk.data(); c.scount = scount.data();
    c.lhead = lhead.data(); c.lpool = lpool.data(); c.lcap = 1u << 22; c.lcount = &lcount;
    c.st_score = stsc.data(); c.st_kmer = stkm.data(); c.st_rank = strk.data(); c.st_cap = stcap; c.st_count = &stcount;
    c.hpool = hpool.data(); c.hcap = (uint32_t)hpool.size(); c.hcount = &hcount;
    for (uint32_t ct = 0; ct < nc; ++ct)
        for (size_t i = 0; i + 1 < nodepth[ct].size(); i += 2) {
            stcount = 0;
            kc_score_correct_level2(c, ct, nodepth[ct][i], nodepth[ct][i + 1]);
        }
    if (err) return (int)err;
    c.keep_zero_marks = snp_valid ? 1 : 0;
    for (uint32_t ct = 0; ct < nc; ++ct) {
        const uint32_t g0 = v->ctg_off[ct];
        const bool has_next = (int64_t)v->read_begin[ct + 1] < n;
        if (geo) {
            // the way the device pass does it with the replay (np1_device.hip:replay_votes): all parts of the contig, the first loop of every
            // part from the replayed iterator, winners kept aside; the replay again wherever a loop left through the max_count_kmer break;
            // second-loop passes for the parts left empty; then the writes in part order.  snp_valid: both rounds that way.
            const int tid = geo->tid[ct];
            if (tid < 0 || (size_t)tid >= geo->bai->refs.size()) return -30;
            const np1replay::RefIndex ix(geo->bai->refs[(size_t)tid]);
            const int64_t rb = (int64_t)v->read_begin[ct], re = (int64_t)v->read_begin[ct + 1];
            const np1replay::Records rec{geo->voff + rb, geo->voff_end + rb, v->pos + rb, endpos.data() + rb, re - rb, has_next, (int32_t)(v->ctg_off[ct + 1] - g0)};
            std::vector<std::vector<uint8_t>> wins;
            std::vector<uint8_t> state;
            auto replay_votes = [&](const std::vector<int32_t>& pse, const std::vector<uint8_t>& skip) -> int {
                const uint32_t n_parts = (uint32_t)(pse.size() / 2);
                std::vector<int32_t> next_end(n_parts, -1);
                for (uint32_t p = 0; p + 1 < n_parts; ++p) next_end[p] = pse[2 * (p + 1) + 1];
                wins.assign(n_parts, {});
                state.assign(n_parts, 0);
                std::vector<uint32_t> brk(n_parts, 0), limit(n_parts, 0);
                const std::vector<uint8_t> snap(sflag);
                np1replay::FirstLoop fl, nx;
                auto vote = [&](uint32_t p, int32_t n2) {
                    if (skip[p]) { brk[p] = 0; return 0; }
                    const int32_t ps = pse[2 * p], pe = pse[2 * p + 1];
                    const int32_t length = (int32_t)(soff[g0 + pe] - soff[g0 + ps] + 1);
                    wins[p].assign((size_t)length, 0);
                    hcount = 0;
                    std::vector<uint32_t> glist(fl.list.begin() + fl.first[p], fl.list.begin() + fl.first[p + 1]);
                    for (uint32_t& x : glist) x += (uint32_t)rb;
                    uint32_t bk = 0;
                    const KcReplay rp{glist.data(), (uint32_t)glist.size(), fl.stale[p] >= 0 ? fl.stale[p] + rb : -1, n2, n2 < 0 ? &bk : nullptr};
                    const int32_t st = kc_part_winner(c, ct, ps, pe, has_next, wins[p].data(), length, &rp);
                    if (n2 < 0) brk[p] = bk;
                    return (int)st;
                };
                fl = np1replay::first_loop(ix, rec, pse.data(), next_end.data(), n_parts, limit.data(), skip.data());
                for (uint32_t p = 0; p < n_parts; ++p) state[p] = (uint8_t)vote(p, -1);
                for (uint32_t x : brk) np1m_replay_breaks += x != 0;
                for (int it = 0; brk != limit; ++it) {
                    if (it > 256) return -32;
                    limit = brk;
                    nx = np1replay::first_loop(ix, rec, pse.data(), next_end.data(), n_parts, limit.data(), skip.data());
                    bool same = true;
                    for (uint32_t p = 0; p < n_parts && same; ++p) {
                        const uint32_t n_old = fl.first[p + 1] - fl.first[p], n_new = nx.first[p + 1] - nx.first[p], n_cmp = brk[p] ? brk[p] : n_old;
                        same = n_new == n_cmp && n_cmp <= n_old && std::equal(nx.list.begin() + nx.first[p], nx.list.begin() + nx.first[p + 1], fl.list.begin() + fl.first[p]) &&
                               (state[p] != 2 || nx.stale[p] == fl.stale[p]);
                    }
                    fl = nx;
                    if (same) break;
                    if (!c.keep_zero_marks) sflag = snap;
                    ++np1m_replay_revotes;
                    for (uint32_t p = 0; p < n_parts; ++p) state[p] = (uint8_t)vote(p, -1);
                }
                std::vector<uint8_t> empty(n_parts, 0);
                for (uint32_t p = 0; p < n_parts; ++p) empty[p] = state[p] == 2;
                const std::vector<uint32_t> n2 = np1replay::second_loop_passes(ix, rec, pse.data(), next_end.data(), n_parts, empty.data());
                for (uint32_t p = 0; p < n_parts; ++p) if (empty[p]) state[p] = (uint8_t)vote(p, (int32_t)n2[p]);
                return 0;
            };
            std::vector<int32_t> pse;
            for (size_t i = 0; i + 1 < kreg[ct].size(); i += 2) {
                std::vector<int32_t> parts(2 * (size_t)(kreg[ct][i + 1] - kreg[ct][i] + 4));
                const int32_t np = kc_split_region(c, ct, kreg[ct][i], kreg[ct][i + 1], parts.data(), (int32_t)parts.size());
                if (np < 0) return -12;
                pse.insert(pse.end(), parts.begin(), parts.begin() + np);
            }
            const uint32_t n_parts = (uint32_t)(pse.size() / 2);
            if (!n_parts) continue;
            { const int rc = replay_votes(pse, std::vector<uint8_t>(n_parts, 0)); if (rc) return rc; }
            std::vector<int32_t> failed;
            for (uint32_t p = 0; p < n_parts; ++p) {
                if (state[p] == 1) {
                    const uint32_t s0 = soff[g0 + pse[2 * p]];
                    for (size_t t = 0; t < wins[p].size(); ++t) {
                        if (snp_valid) sflag[s0 + t] = (uint8_t)(sflag[s0 + t] & ~KC_FLAG_ZERO);
                        sbase[s0 + t] = wins[p][t];
                    }
                } else if (snp_valid) {
                    failed.push_back(pse[2 * p]); failed.push_back(pse[2 * p + 1]);
                }
            }
            if (snp_valid && !failed.empty()) {
                std::vector<int32_t> val(pse.size() * 2 + 4 * (size_t)(v->ctg_off[ct + 1] - g0) + 16);
                int32_t nv = 0;
                for (size_t k = 0; k + 1 < failed.size() && nv >= 0; k += 2) nv = kc_fts_split(c, ct, failed[k], failed[k + 1], val.data(), nv, (int32_t)val.size() - 1);
                if (nv < 0) return -13;
                if (nv & 1) { val[(size_t)nv] = (size_t)nv < pse.size() ? pse[(size_t)nv] : 0; ++nv; }
                val.resize((size_t)nv);
                std::vector<uint8_t> skip((size_t)nv / 2, 0);
                for (int32_t k = 0; k + 1 < nv; k += 2) skip[(size_t)k / 2] = val[(size_t)k] > val[(size_t)k + 1];
                const int rc = replay_votes(val, skip);      // (round 2 has flagzero = 0 upstream; the marks no longer matter: emitted with mask 0)
                if (rc) return rc;
                for (uint32_t p = 0; p < (uint32_t)nv / 2; ++p)
                    if (!skip[p] && state[p] == 1) {
                        const uint32_t s0 = soff[g0 + val[2 * p]];
                        for (size_t t = 0; t < wins[p].size(); ++t) sbase[s0 + t] = wins[p][t];
                    }
            }
            continue;
        }
        std::vector<int32_t> all_parts, failed;     // snp_valid: the contig's part list (flat) and the parts nothing spanned
        for (size_t i = 0; i + 1 < kreg[ct].size(); i += 2) {
            std::vector<int32_t> parts(2 * (size_t)(kreg[ct][i + 1] - kreg[ct][i] + 4));
            int32_t np = kc_split_region(c, ct, kreg[ct][i], kreg[ct][i + 1], parts.data(), (int32_t)parts.size());
            if (np < 0) return -12;
            for (int32_t k = 0; k + 1 < np; k += 2) {
                const int32_t ps = parts[k], pe = parts[k + 1];
                const int32_t length = (int32_t)(soff[g0 + pe] - soff[g0 + ps] + 1);
                std::vector<uint8_t> win(length);
                hcount = 0;
                all_parts.push_back(ps); all_parts.push_back(pe);
                if (kc_part_winner(c, ct, ps, pe, has_next, win.data(), length)) {
                    const uint32_t s0 = soff[g0 + ps];
                    for (int32_t t = 0; t < length; ++t) {
                        if (snp_valid) sflag[s0 + t] = (uint8_t)(sflag[s0 + t] & ~KC_FLAG_ZERO);   // contig_clean_flag (contig.c:823-831)
                        sbase[s0 + t] = win[t];   // contig_update_contig (contig.c:811-821)
                    }
                } else if (snp_valid) {
                    failed.push_back(ps); failed.push_back(pe);
                }
            }
        }
        if (snp_valid && !failed.empty()) {     // second round (what k_sv_round2_parts / k_sv_round2_apply do)
            std::vector<int32_t> val(all_parts.size() * 2 + 4 * (size_t)(v->ctg_off[ct + 1] - g0) + 16);
            int32_t nv = 0;
            for (size_t k = 0; k + 1 < failed.size() && nv >= 0; k += 2) nv = kc_fts_split(c, ct, failed[k], failed[k + 1], val.data(), nv, (int32_t)val.size() - 1);
            if (nv < 0) return -13;
            if (nv & 1) { val[(size_t)nv] = (size_t)nv < all_parts.size() ? all_parts[(size_t)nv] : 0; ++nv; }
            for (int32_t k = 0; k + 1 < nv; k += 2) {
                const int32_t ps = val[(size_t)k], pe = val[(size_t)k + 1];
                if (ps > pe) continue;
                const int32_t length = (int32_t)(soff[g0 + pe] - soff[g0 + ps] + 1);
                std::vector<uint8_t> win(length);
                hcount = 0;
                if (kc_part_winner(c, ct, ps, pe, has_next, win.data(), length)) {
                    const uint32_t s0 = soff[g0 + ps];
                    for (int32_t t = 0; t < length; ++t) sbase[s0 + t] = win[t];
                }
            }
        }
    }
    if (snp_valid) for (uint32_t s = 0; s < S; ++s) sflag[s] = 0;     // emitted without marks (snpvalid.c:30)
    if (err) return (int)err;
    std::vector<uint16_t> slot_res(S + 64);
    for (uint32_t s = 0; s < S; ++s) slot_res[s] = (uint16_t)(sbase[s] | sflag[s] << 8);
    std::vector<uint32_t> opos(S + 1);
    uint32_t o = 0;
    for (uint32_t s = 0; s < S; ++s) { opos[s] = o; o += (slot_res[s] & 0xff) != 3; }
    opos[S] = o;
    char* buf = (char*)calloc(1, (size_t)o + 1);
    for (uint32_t s = 0; s < S; ++s) emit_slot(s, slot_res.data(), slot_info.data(), opos.data(), 1u, (uint8_t*)buf);
    for (uint32_t ct = 0; ct <= nc; ++ct) bounds[ct] = opos[soff[v->ctg_off[ct]]];
    *out = buf;
    return 0;
}

int np1m_kmer_count(const np1_stream_view* v, const Configure* cfg, char** out, uint32_t* bounds) { return kmer_model(v, cfg, out, bounds, false); }
int np1m_snp_valid(const np1_stream_view* v, const Configure* cfg, char** out, uint32_t* bounds) { return kmer_model(v, cfg, out, bounds, true); }
// kmer_count with the replay of the reference's region iterator: `bai_path` = index of the BAM the stream was read from, tid[c] = BAM
// reference id of contig c, voff / voff_end = np1_stream_voffs of that stream
int np1m_kmer_count_replay(const np1_stream_view* v, const Configure* cfg, const char* bai_path, const int32_t* tid, const uint64_t* voff, const uint64_t* voff_end,
                           char** out, uint32_t* bounds) {
    np::BaiIndex bai;
    if (!bai.load(bai_path)) return -31;
    const ModelGeometry geo{&bai, tid, voff, voff_end};
    return kmer_model(v, cfg, out, bounds, false, &geo);
}

// task 4 the same way (both rounds of snp_valid on the replayed iterator)
int np1m_snp_valid_replay(const np1_stream_view* v, const Configure* cfg, const char* bai_path, const int32_t* tid, const uint64_t* voff, const uint64_t* voff_end,
                          char** out, uint32_t* bounds) {
    np::BaiIndex bai;
    if (!bai.load(bai_path)) return -31;
    const ModelGeometry geo{&bai, tid, voff, voff_end};
    return kmer_model(v, cfg, out, bounds, true, &geo);
}
unsigned long long np1m_replay_revote_count() { return np1m_replay_revotes; }
unsigned long long np1m_replay_break_count() { return np1m_replay_breaks; }

// ---------------------------------------------------------------------------------------------------------------------------
// Intra-contig tiling: one contig cut into tiles of tile_bp draft bases that are polished independently
// (one tile per GPU when a single contig is all there is) and joined.  What makes this exact without any exchange between tiles:
//  * votes are local: what a slot receives depends on the records covering it, nothing else (contig.c:247-331);
//  * the chain restarts behind every slot that came out of the vote with one state -- scores are exact integers and a single
//    state is every path's predecessor at the same score (np1_core.h dp_run) -- so a tile needs no carried state, only to start
//    and end its own computation at such slots.
// A tile therefore computes [a - halo, b + halo) with every record that overlaps that stretch (the draft hull of those records, so
// that none sticks out), checks that a single-state slot lies inside each halo among the slots whose votes are complete, and
// contributes the output of its own bases [a, b) (with the insertion columns behind each).  No slot of that kind in a halo (a
// multi-state run as long as the halo): the tile is recomputed with the halo doubled.  tstats: [0] tiles, [1] recomputations,
// [2] records processed over all tiles.
int np1m_score_chain_tiled(const np1_stream_view* v, const Configure* cfg, uint32_t tile_bp, uint32_t halo_bp, char** out, uint32_t* bounds, uint64_t* tstats) {
    const uint32_t nc = (uint32_t)v->n_contigs;
    std::string joined;
    uint64_t n_tiles = 0, n_redo = 0, n_rec = 0;
    std::vector<int32_t> endp((size_t)(v->n_reads ? v->n_reads : 1));
    for (int64_t r = 0; r < v->n_reads; ++r) {      // reference end: only M and D advance the walk (contig.c:262-326)
        int32_t e = v->pos[r];
        const uint32_t* cg = v->cigar + v->cigar_off[r];
        for (uint32_t i = 0; i < v->n_cigar[r]; ++i)
            if ((cg[i] & 15u) == 0 || (cg[i] & 15u) == 2) e += (int32_t)(cg[i] >> 4);
        endp[(size_t)r] = e;
    }
    g_keep_map = true;
    int rc = 0;
    for (uint32_t c = 0; c < nc && rc == 0; ++c) {
        bounds[c] = (uint32_t)joined.size();
        const int32_t L = (int32_t)(v->ctg_off[c + 1] - v->ctg_off[c]);
        const int64_t r0 = (int64_t)v->read_begin[c], r1 = (int64_t)v->read_begin[c + 1];
        for (int32_t a = 0; a < L && rc == 0; a += (int32_t)tile_bp) {
            const int32_t b = std::min<int64_t>(L, (int64_t)a + tile_bp);
            ++n_tiles;
            for (uint32_t halo = halo_bp;; halo *= 2) {
                const int32_t e_lo = std::max<int64_t>(0, (int64_t)a - halo), e_hi = std::min<int64_t>(L, (int64_t)b + halo);
                std::vector<int64_t> pick;
                int32_t lo = e_lo, hi = e_hi;
                for (int64_t r = r0; r < r1; ++r) {
                    const int32_t p = v->pos[r], e = endp[(size_t)r];
                    // (<= / >=: a record that starts at e_hi with an insertion votes on the columns behind base e_hi - 1)
                    if (p <= e_hi && e >= e_lo) {
                        pick.push_back(r);
                        lo = std::min(lo, std::max(p, 0));
                        hi = std::max(hi, std::min(e, L));
                    }
                }
                // one more base on each side: position 0 and the last base of a contig are special to the walk (an insertion before
                // the first base shifts the query window instead of voting, contig.c:315-319; none is taken behind the last base),
                // and no record of the tile may meet an artificial one
                lo = std::max(0, lo - 1);
                hi = std::min(L, hi + 1);
                const size_t m = pick.size();
                std::vector<int32_t> pos(m + 1), lq(m + 1), isz(m + 1);
                std::vector<uint32_t> ctg(m + 1, 0), ncg(m + 1);
                std::vector<uint16_t> flag(m + 1);
                std::vector<uint64_t> coff(m + 1), soffv(m + 1), qoff(m + 1);
                std::vector<uint8_t> mapq(m + 1);
                for (size_t k = 0; k < m; ++k) {
                    const int64_t r = pick[k];
                    pos[k] = v->pos[r] - lo; lq[k] = v->l_qseq[r]; ncg[k] = v->n_cigar[r]; flag[k] = v->flag[r];
                    coff[k] = v->cigar_off[r]; soffv[k] = v->seq_off[r];
                    if (v->mapq) mapq[k] = v->mapq[r];
                    if (v->isize) isz[k] = v->isize[r];
                    if (v->qual_off) qoff[k] = v->qual_off[r];
                }
                np1_stream_view sub = *v;
                const int32_t sub_len = hi - lo;
                const uint32_t sub_off[2] = {0u, (uint32_t)sub_len};
                const uint64_t sub_rb[2] = {0ull, (uint64_t)m};
                sub.n_contigs = 1; sub.n_reads = (int64_t)m; sub.ctg_len = &sub_len; sub.ctg_off = sub_off; sub.read_begin = sub_rb;
                sub.draft = v->draft + v->ctg_off[c] + lo; sub.draft_len = sub_len;
                sub.pos = pos.data(); sub.ctg = ctg.data(); sub.flag = flag.data(); sub.n_cigar = ncg.data(); sub.l_qseq = lq.data();
                sub.cigar_off = coff.data(); sub.seq_off = soffv.data(); sub.mapq = mapq.data(); sub.isize = isz.data(); sub.qual_off = qoff.data();
                char* part = nullptr;
                uint32_t pb[2] = {0, 0};
                n_rec += m;
                rc = np1m_score_chain(&sub, cfg, &part, pb, nullptr);
                if (rc != 0) break;
                // a single-state slot inside each halo, among the slots whose votes are complete (e_lo .. e_hi), clear of the two
                // slots behind an artificial start whose draft context is cut short (contig.c:373-383)
                bool left_ok = a == 0, right_ok = b == L;
                if (!left_ok) {
                    const uint32_t s_from = g_soff[(size_t)(e_lo - lo)] + (e_lo > 0 ? 2u : 0u), s_to = g_soff[(size_t)(a - lo)];
                    for (uint32_t sl = s_from; sl < s_to && !left_ok; ++sl) left_ok = g_single[sl] != 0;
                    if (e_lo == 0) left_ok = true;      // the halo reaches the real start of the contig
                }
                if (!right_ok) {
                    const uint32_t s_from = g_soff[(size_t)(b - lo)], s_to = g_soff[(size_t)(e_hi - lo)];
                    for (uint32_t sl = s_from; sl < s_to && !right_ok; ++sl) right_ok = g_single[sl] != 0;
                    if (e_hi == L) right_ok = true;
                }
                if (left_ok && right_ok) {
                    const uint32_t o0 = g_opos[g_soff[(size_t)(a - lo)]], o1 = b == L ? pb[1] : g_opos[g_soff[(size_t)(b - lo)]];
                    joined.append(part + o0, part + o1);
                    free(part);
                    break;
                }
                free(part);
                ++n_redo;
                if (halo > (1u << 28)) { rc = -20; break; }
            }
        }
    }
    g_keep_map = false;
    if (rc != 0) return rc;
    bounds[nc] = (uint32_t)joined.size();
    char* buf = (char*)calloc(1, joined.size() + 1);
    memcpy(buf, joined.data(), joined.size());
    *out = buf;
    if (tstats) { tstats[0] = n_tiles; tstats[1] = n_redo; tstats[2] = n_rec; }
    return 0;
}

// The product's tiling driver (nextpolish_amd/csrc/np1_tile.cpp) with the model in place of the device: tiles read from the FILES through
// the index (np_stream.cpp: load_stream_region), joined with the same index arithmetic.  CPU check of the region loader and of the join.
int np1m_score_chain_tiled_files(const char* fasta, const char* bam, const char* name, const Configure* cfg, int64_t tile_bp, int64_t halo_bp, int64_t first_tile,
                                 int64_t tile_stride, char** out, int64_t* out_len, uint64_t* tstats) {
    np::Fai fai;
    if (!fai.load(fasta)) return -30;
    const int id = fai.find(name);
    if (id < 0) return -31;
    std::string draft;
    if (!fai.fetch(id, &draft)) return -32;
    np::BaiIndex bai;
    if (!bai.load(std::string(bam) + ".bai")) return -33;
    const int64_t L = (int64_t)draft.size();
    std::string joined;
    uint64_t n_tiles = 0, n_redo = 0, n_rec = 0;
    g_keep_map = true;
    int rc = 0;
    int64_t tile_no = 0;
    for (int64_t a = 0; a < L && rc == 0; a += tile_bp, ++tile_no) {
        if (tile_no < first_tile || (tile_no - first_tile) % tile_stride != 0) continue;     // (np1_tile.cpp: this call's tiles)
        const int64_t b = a + tile_bp < L ? a + tile_bp : L;
        ++n_tiles;
        for (int64_t halo = halo_bp;; halo *= 2) {
            const int32_t e_lo = (int32_t)(a - halo > 0 ? a - halo : 0), e_hi = (int32_t)(b + halo < L ? b + halo : L);
            np::ReadStream rs;
            std::string err;
            int32_t lo = 0, hi = 0;
            if (!np::load_stream_region(bam, bai, name, draft, e_lo, e_hi, &rs, &lo, &hi, &err)) { rc = -34; break; }
            n_rec += rs.n_reads();
            np1_stream_view v;
            memset(&v, 0, sizeof(v));
            v.n_contigs = 1; v.n_reads = (int64_t)rs.n_reads(); v.draft_len = (int64_t)rs.draft.size(); v.draft = rs.draft.data();
            v.ctg_len = rs.ctg_len.data(); v.ctg_off = rs.ctg_off.data(); v.read_begin = rs.read_begin.data();
            v.pos = rs.pos.data(); v.ctg = rs.ctg.data(); v.flag = rs.flag.data(); v.n_cigar = rs.n_cigar.data(); v.l_qseq = rs.l_qseq.data();
            v.cigar_off = rs.cigar_off.data(); v.seq_off = rs.seq_off.data(); v.cigar = rs.cigar.data(); v.seq = rs.seq.data();
            v.cigar_len = (int64_t)rs.cigar.size(); v.seq_len = (int64_t)rs.seq.size();
            v.mapq = rs.mapq.data(); v.isize = rs.isize.data(); v.qual_off = rs.qual_off.data();
            char* part = nullptr;
            uint32_t pb[2] = {0, 0};
            rc = np1m_score_chain(&v, cfg, &part, pb, nullptr);
            if (rc != 0) break;
            const uint32_t i_elo = (uint32_t)(e_lo - lo), i_a = (uint32_t)(a - lo), i_b = (uint32_t)(b - lo), i_ehi = (uint32_t)(e_hi - lo);
            bool left_any = false, right_any = false;
            for (uint32_t sl = g_soff[i_elo] + (e_lo > 0 ? 2u : 0u); sl < g_soff[i_a]; ++sl) left_any = left_any || g_single[sl] != 0;
            for (uint32_t sl = g_soff[i_b]; sl < g_soff[i_ehi]; ++sl) right_any = right_any || g_single[sl] != 0;
            const bool left_ok = a == 0 || e_lo == 0 || left_any, right_ok = b == L || e_hi == L || right_any;
            if (left_ok && right_ok) {
                joined.append(part + g_opos[g_soff[i_a]], part + g_opos[g_soff[i_b]]);
                free(part);
                break;
            }
            free(part);
            ++n_redo;
            if (halo > ((int64_t)1 << 30)) { rc = -20; break; }
        }
    }
    g_keep_map = false;
    if (rc != 0) return rc;
    char* buf = (char*)calloc(1, joined.size() + 1);
    memcpy(buf, joined.data(), joined.size());
    *out = buf;
    *out_len = (int64_t)joined.size();
    if (tstats) { tstats[0] = n_tiles; tstats[1] = n_redo; tstats[2] = n_rec; }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The upload forms of DESIGN.md section 4 (np1_upload.h): built by the product's own builders, undone by host restatements of the
// device kernels; what comes back must be the stream's arrays.  The compact record form is built whatever the batch size (the
// product uses it from 4 M records on).  Returns 0, or the number of the first form that does not round-trip (1 bases, 2 draft,
// 3 positions, 4 operation counts, 5 read lengths, 6 operations); sizes (optional, 6 words): bytes of the bases as they are / in
// 2-bit form with exceptions, of the per-record fields as they are / compact, plain records, full positions.
int np1m_upload_roundtrip(const np1_stream_view* v, uint64_t* sizes) {
    np::ReadStream s;
    const size_t n = (size_t)v->n_reads, nc = (size_t)v->n_contigs;
    s.names.resize(nc);
    s.ctg_off.assign(v->ctg_off, v->ctg_off + nc + 1);
    s.read_begin.assign(v->read_begin, v->read_begin + nc + 1);
    s.draft.assign(v->draft, (size_t)v->draft_len);
    s.pos.assign(v->pos, v->pos + n);
    s.n_cigar.assign(v->n_cigar, v->n_cigar + n);
    s.l_qseq.assign(v->l_qseq, v->l_qseq + n);
    s.cigar_off.assign(v->cigar_off, v->cigar_off + n);
    s.cigar.assign(v->cigar, v->cigar + v->cigar_len);
    s.seq.assign(v->seq, v->seq + v->seq_len);
    std::vector<uint8_t> seq2, esc_val, draft4, desc_val;
    std::vector<uint64_t> esc_at, desc_at;
    uint64_t sz[6] = {s.seq.size(), s.seq.size(), 0, 0, 0, 0};
    if (np1up::build_seq2(s.seq, &seq2, &esc_at, &esc_val, 1)) {      // (ratio 1: kept however many exceptions there are)
        std::vector<uint8_t> back;
        np1up::undo_seq2(seq2, esc_at, esc_val, &back);
        if (back.size() < s.seq.size() || memcmp(back.data(), s.seq.data(), s.seq.size()) != 0) return 1;
        sz[1] = seq2.size() + 9 * esc_at.size();
    }
    if (np1up::build_draft4(s.draft, &draft4, &desc_at, &desc_val, 1)) {
        std::string back;
        np1up::undo_draft4(draft4, s.draft.size(), desc_at, desc_val, &back);
        if (back != s.draft) return 2;
    }
    if (n) {
        np1_stream::Compact C;
        np1up::build_compact(s, &C);
        std::vector<int32_t> pos, lq;
        std::vector<uint32_t> ncig, cigar;
        np1up::undo_compact(C, n, &pos, &ncig, &lq, &cigar);
        if (pos != s.pos) return 3;
        if (ncig != s.n_cigar) return 4;
        if (lq != s.l_qseq) return 5;
        if (cigar != s.cigar) return 6;
        sz[2] = 12 * n + 4 * s.cigar.size();
        sz[3] = np1up::compact_bytes(C, n);
        for (size_t i = 0; i < n; ++i) sz[4] += (C.plain[i >> 5] >> (i & 31u)) & 1u;
        sz[5] = C.x_pos.size();
    }
    if (sizes) memcpy(sizes, sz, sizeof(sz));
    return 0;
}

void np1m_free(void* p) { free(p); }

// The region walk in its run-parallel form (what k_kc_regions does, one step after the other) and in its literal form, on
// a bare contig: code / flag per base, the ascending list of flagged positions.  mode 0 = literal (kc_find_regions +
// kc_merge_regions), 1 = runs, neighbour test, replay chains, compaction, register merge.  Returns the value count.
int np1m_regions(const uint8_t* code, const uint8_t* flag, int32_t L, const uint32_t* F, uint32_t m, uint32_t gap, uint32_t con, int32_t ext,
                 int with_ext, int mode, int32_t* out, int32_t cap) {
    using namespace np1k;
    if (m == 0) return 0;
    if (mode == 0) {
        const int32_t k = kc_find_regions(code, flag, L, F, m, gap, con, ext, with_ext != 0, out, cap);
        return k < 0 ? k : kc_merge_regions(out, k);
    }
    std::vector<uint32_t> rs;
    for (uint32_t k = 0; k < m; ++k)
        if (k == 0 || F[k] - F[k - 1] - 1u > gap) rs.push_back(k);
    const uint32_t n_runs = (uint32_t)rs.size();
    std::vector<KcRun> runs(n_runs);
    for (uint32_t r = 0; r < n_runs; ++r)
        kc_run_region(F, rs[r], (r + 1 < n_runs ? rs[r + 1] : m) - 1, code, flag, L, gap, con, ext, with_ext != 0, &runs[r]);
    std::vector<uint32_t> bad;
    for (uint32_t r = 1; r < n_runs; ++r)   // (on the snapshot before any chain ran, like the kernel)
        if (runs[r - 1].emit == 2u || (int64_t)runs[r].first_pos < kc_reach(runs[r - 1])) bad.push_back(r);
    uint32_t next_q = 0;
    bool ended = false;
    for (size_t b = 0; b < bad.size() && !ended; ++b) {
        uint32_t q = bad[b];
        if (q < next_q) continue;
        q = kc_chain(F, runs.data(), n_runs, q, code, flag, L, gap, con, ext, with_ext != 0, true, &ended);
        next_q = q + 1;
    }
    int32_t n = 0;
    for (uint32_t r = 0; r < n_runs; ++r)
        if (runs[r].emit) {
            if (n + 2 > cap) return -1;
            out[n++] = runs[r].s;
            out[n++] = runs[r].e;
        }
    bool overlap = false;
    for (int32_t i = 1; i < n / 2; ++i) overlap = overlap || out[2 * i] < out[2 * (i - 1) + 1];
    if (n && !(out[0] < out[1])) return kc_merge_regions(out, n);   // (needs cap >= n + 2, like the walk's own buffer)
    return overlap ? kc_merge_fast(out, n) : n;
}
}

// ---------------------------------------------------------------------------------------------------------------
// snp_phase (task 3): the stage bodies of np1_phase.h driven the way np1_batch_snp_phase drives the kernels
#include "../../nextpolish_amd/csrc/np1_phase.h"
#include "../../nextpolish_amd/csrc/np1_phase_host.h"

extern "C" int np1m_snp_phase(const np1_stream_view* vs, const np1_stream_view* vl, const Configure* cfg, char** out, uint32_t* bounds) {
    using namespace np1p;
    const uint32_t nc = (uint32_t)vs->n_contigs;
    const uint64_t G = (uint64_t)vs->draft_len;
    if ((uint32_t)vl->n_contigs != nc || (uint64_t)vl->draft_len != G) return -20;
    if ((vs->qual_len == 0 && vs->n_reads > 0) || (vl->qual_len == 0 && vl->n_reads > 0)) return -10;
    uint32_t err = 0;
    SpParams P;
    P.min_depth_snp = cfg->min_depth_snp; P.min_count_snp = cfg->min_count_snp; P.min_count_snp_link = cfg->min_count_snp_link;
    P.max_variant_count_lgs = cfg->max_variant_count_lgs; P.read_len = cfg->read_len; P.ext_len_edge = cfg->ext_len_edge;
    P.min_snp_factor_sgs = cfg->min_snp_factor_sgs; P.max_clip_ratio_lgs = cfg->max_clip_ratio_lgs; P.rate_lgs = cfg->indel_balance_factor_lgs;
    P.max_indel_factor_lgs = cfg->max_indel_factor_lgs; P.max_snp_factor_lgs = cfg->max_snp_factor_lgs; P.ploidy = cfg->ploidy;
    auto make_ctx = [&](const np1_stream_view* v, bool lr, std::vector<uint8_t>& level, std::vector<int32_t>& endpos) {
        KcCtx c;
        memset(&c, 0, sizeof(c));
        c.R = ReadsDev{v->pos, v->ctg, v->flag, v->n_cigar, v->l_qseq, v->cigar_off, v->seq_off, v->cigar, v->seq};
        c.mapq = v->mapq; c.isize = v->isize; c.qual_off = v->qual_off; c.qual = v->qual;
        c.ctg_off = vs->ctg_off; c.read_begin = v->read_begin;
        c.trim = cfg->trim_len_edge; c.ext_len_edge = cfg->ext_len_edge;
        c.min_map_quality = cfg->min_map_quality; c.read_tlen = cfg->read_tlen;
        c.max_clip_ratio_sgs = cfg->max_clip_ratio_sgs; c.min_count_ratio_skip = cfg->min_count_ratio_skip;
        c.K = -1; c.rate = cfg->indel_balance_factor_lgs;
        c.third_rule = 1; c.max_indel_factor_lgs = cfg->max_indel_factor_lgs; c.max_snp_factor_lgs = cfg->max_snp_factor_lgs;
        c.err = &err;
        const int64_t n = v->n_reads;
        level.assign(n ? n : 1, 0); endpos.assign(n ? n : 1, 0);
        int32_t max_span = 1;
        for (int64_t r = 0; r < n; ++r) {
            level[r] = lr ? (uint8_t)sp_lr_level(c.R, r, P.max_clip_ratio_lgs)
                          : (uint8_t)kc_filter_level(c.R, r, c.mapq, c.isize, c.read_tlen, c.max_clip_ratio_sgs, c.min_map_quality);
            endpos[r] = kc_endpos(c.R, r);
            if (endpos[r] - v->pos[r] > max_span) max_span = endpos[r] - v->pos[r];
        }
        c.level = level.data(); c.endpos = endpos.data(); c.max_span = max_span;
        return c;
    };
    std::vector<uint8_t> lev_s, lev_l;
    std::vector<int32_t> end_s, end_l;
    KcCtx cs = make_ctx(vs, false, lev_s, end_s), cl = make_ctx(vl, true, lev_l, end_l);
    cl.keep_zero_marks = 1;
    // P1: short-read columns, first slot space
    std::vector<uint32_t> ins(G + 1, 0);
    for (int64_t r = 0; r < vs->n_reads; ++r) if (lev_s[r] >= 1) sp_insert_record(cs.R, r, vs->ctg_off, ins.data(), 0, nullptr, nullptr);
    std::vector<uint32_t> soff1(G + 2);
    uint64_t acc = 0;
    for (uint64_t g = 0; g < G; ++g) { soff1[g] = (uint32_t)acc; acc += 1 + ins[g]; }
    soff1[G] = soff1[G + 1] = (uint32_t)acc;
    const uint32_t S1 = (uint32_t)acc;
    std::vector<uint8_t> info1(S1 + 64, 0), sbase1(S1 + 64), sflag1(S1 + 64), dec(S1 + 64, 0), top(S1 + 64, 0);
    std::vector<uint32_t> sown1(S1 + 64, 0);
    for (uint32_t ct = 0; ct < nc; ++ct)
        for (uint32_t g = vs->ctg_off[ct]; g < vs->ctg_off[ct + 1]; ++g)
            slotinfo_base((const uint8_t*)vs->draft, g, vs->ctg_off[ct], vs->ctg_off[ct + 1], soff1.data(), info1.data(), sown1.data());
    for (uint32_t s = 0; s < S1; ++s) { sbase1[s] = info1[s] & 0xf; sflag1[s] = (info1[s] & SI_LOWER) ? 1 : 0; }
    std::vector<uint16_t> scount1(S1 + 64, 0);
    // P2: histogram
    std::vector<uint32_t> cnt((size_t)S1 * 16 + 16, 0), first((size_t)S1 * 16 + 16, 0xffffffffu);
    cs.soff = soff1.data();
    for (int64_t r = 0; r < vs->n_reads; ++r) sp_hist_record(cs, r, cnt.data(), first.data());
    // P3: slot verdicts (the walk of ts_find_snps stops on the main slot of each contig's last base)
    for (uint32_t ct = 0; ct < nc; ++ct) {
        const uint32_t g0 = vs->ctg_off[ct], g1 = vs->ctg_off[ct + 1];
        if (g1 == g0) continue;
        for (uint32_t s = soff1[g0]; s <= soff1[g1 - 1]; ++s)
            sp_slot_decide(P, s, cnt.data(), first.data(), sbase1.data(), sflag1.data(), scount1.data(), dec.data(), top.data(), &err);
    }
    if (err) return (int)err;
    // P4: sites
    std::vector<uint8_t> dirty(G + 1, 0), alle(G + 1, 0);
    std::vector<uint32_t> site_g, site_ctg, site_first(nc + 1, 0);
    for (uint32_t ct = 0; ct < nc; ++ct) {
        const uint32_t g0 = vs->ctg_off[ct], g1 = vs->ctg_off[ct + 1];
        site_first[ct] = (uint32_t)site_g.size();
        for (uint32_t g = g0; g < g1; ++g) {
            dirty[g] = (uint8_t)sp_base_site(g, (int32_t)(g - g0), (int32_t)(g1 - g0), soff1.data(), dec.data(), top.data(), &alle[g]);
            if (dirty[g]) { site_g.push_back(g); site_ctg.push_back(ct); sflag1[soff1[g]] |= F_SNP; }
        }
    }
    site_first[nc] = (uint32_t)site_g.size();
    const uint32_t NS = (uint32_t)site_g.size();
    std::vector<int32_t> site_left(NS + 1), site_right(NS + 1), site_len(NS + 1, 1), site_pos(NS + 1);
    for (uint32_t k = 0; k < NS; ++k) {
        const uint32_t g0 = vs->ctg_off[site_ctg[k]];
        site_pos[k] = (int32_t)(site_g[k] - g0);
        sp_site_anchors(dirty.data() + g0, site_pos[k], (int32_t)(vs->ctg_off[site_ctg[k] + 1] - g0), &site_left[k], &site_right[k]);
    }
    // P5: low-depth regions, INSERT marks
    std::vector<std::vector<int32_t>> nodepth(nc);
    for (uint32_t ct = 0; ct < nc; ++ct) {
        const uint32_t g0 = vs->ctg_off[ct], g1 = vs->ctg_off[ct + 1];
        if (g1 == g0) continue;
        std::vector<uint32_t> F;
        for (uint32_t s = soff1[g0]; s <= soff1[g1 - 1]; ++s) if (sflag1[s] & F_DEPTH) F.push_back(s);
        std::vector<int32_t> buf(2 * F.size() + 4);
        int32_t n = sp_depth_regions(F.data(), (uint32_t)F.size(), soff1.data(), sown1.data(), g0, (int32_t)(g1 - g0), (uint32_t)P.ext_len_edge, P.ext_len_edge,
                                     buf.data(), (int32_t)buf.size());
        if (n < 0) return -21;
        n = kc_merge_regions(buf.data(), n);
        nodepth[ct].assign(buf.begin(), buf.begin() + n);
        for (int32_t i = 0; i + 1 < n; i += 2)
            for (uint32_t s = soff1[g0 + (uint32_t)buf[i]]; s <= soff1[g0 + (uint32_t)buf[i + 1]]; ++s) sflag1[s] |= F_INSERT;
    }
    // P6: long-read columns behind marked bases, second slot space
    for (int64_t r = 0; r < vl->n_reads; ++r)
        if (lev_l[r] >= 1) sp_insert_record(cl.R, r, vs->ctg_off, ins.data(), F_INSERT | F_SNP, soff1.data(), sflag1.data());
    std::vector<uint32_t> soff(G + 2);
    acc = 0;
    for (uint64_t g = 0; g < G; ++g) { soff[g] = (uint32_t)acc; acc += 1 + ins[g]; }
    soff[G] = soff[G + 1] = (uint32_t)acc;
    const uint32_t S = (uint32_t)acc;
    std::vector<uint8_t> sbase(S + 64), sflag(S + 64), slot_info(S + 64, 0);
    std::vector<uint16_t> scount(S + 64, 0), srefk(S + 64, 0);
    std::vector<uint32_t> sown(S + 64, 0);
    for (uint64_t g = 0; g < G; ++g)
        sp_reslot_base((uint32_t)g, soff1.data(), soff.data(), sbase1.data(), sflag1.data(), scount1.data(), sbase.data(), sflag.data(), scount.data(), sown.data());
    for (uint32_t ct = 0; ct < nc; ++ct)
        for (uint32_t g = vs->ctg_off[ct]; g < vs->ctg_off[ct + 1]; ++g)
            slotinfo_base((const uint8_t*)vs->draft, g, vs->ctg_off[ct], vs->ctg_off[ct + 1], soff.data(), slot_info.data());
    // shared slot state of both contexts
    std::vector<uint32_t> lhead(S + 64, 0), lpool(2ull * (1u << 22));
    uint32_t lcount = 0, stcount = 0, hcount = 0;
    const uint32_t stcap = 1u << 20;
    std::vector<long long> stsc(16ull * stcap);
    std::vector<uint16_t> stkm(16ull * stcap);
    std::vector<uint8_t> strk(16ull * stcap);
    std::vector<uint8_t> hpool(64u << 20);
    for (KcCtx* c : {&cs, &cl}) {
        c->soff = soff.data(); c->sbase = sbase.data(); c->sflag = sflag.data(); c->srefk = srefk.data(); c->scount = scount.data();
        c->lhead = lhead.data(); c->lpool = lpool.data(); c->lcap = 1u << 22; c->lcount = &lcount;
        c->st_score = stsc.data(); c->st_kmer = stkm.data(); c->st_rank = strk.data(); c->st_cap = stcap; c->st_count = &stcount;
        c->hpool = hpool.data(); c->hcap = (uint32_t)hpool.size(); c->hcount = &hcount;
        c->sown = sown.data();
    }
    // P7: site verdicts
    std::vector<uint32_t> roff(NS + 1, 0), rstride(NS + 1, 0);
    uint32_t racc = 0;
    for (uint32_t k = 0; k < NS; ++k) {
        rstride[k] = soff[site_g[k] + 1] - soff[site_g[k]] + 1 + 8;
        roff[k] = racc;
        racc += 2 * rstride[k];
    }
    std::vector<uint8_t> rpool(racc + 16, 0), keep(NS + 1, 0);
    for (uint32_t k = 0; k < NS; ++k) { rpool[roff[k]] = alle[site_g[k]] & 0xf; rpool[roff[k] + rstride[k]] = alle[site_g[k]] >> 4; }
    SpSites SS{site_g.data(), site_ctg.data(), site_left.data(), site_right.data(), site_len.data(), keep.data(), roff.data(), rstride.data(), rpool.data()};
    const char* dbg_stop = getenv("NP1O_SP_STOP");
    const int stop = dbg_stop ? atoi(dbg_stop) : 99;
    for (uint32_t k = 0; k < NS && stop >= 2; ++k) {
        hcount = 0;
        sp_site_verdict(cs, cl, P, SS, k, soff1.data(), cnt.data(), first.data());
    }
    if (err) return (int)err;
    // P9: low-depth regions, touching ones as one group
    {
        std::vector<uint32_t> reg_ctg;
        std::vector<int32_t> reg_se;
        for (uint32_t ct = 0; ct < nc; ++ct)
            for (size_t i = 0; i + 1 < nodepth[ct].size(); i += 2) { reg_ctg.push_back(ct); reg_se.push_back(nodepth[ct][i]); reg_se.push_back(nodepth[ct][i + 1]); }
        const std::vector<uint32_t> grp = sp_region_groups(reg_ctg, reg_se);
        for (size_t k = 0; k + 1 < grp.size() && stop >= 3; ++k) {
            stcount = 0;
            sp_lowdepth_group(cs, cl, reg_ctg.data(), reg_se.data(), grp[k], grp[k + 1]);
        }
    }
    if (err) return (int)err;
    // kept sites, per contig
    std::vector<uint32_t> k_first(nc + 1, 0), k_roff, k_rstride;
    std::vector<int32_t> k_pos, k_len;
    std::vector<uint32_t> k_g;
    for (uint32_t ct = 0; ct < nc; ++ct) {
        k_first[ct] = (uint32_t)k_pos.size();
        for (uint32_t k = site_first[ct]; k < site_first[ct + 1]; ++k)
            if (keep[k]) { k_pos.push_back(site_pos[k]); k_len.push_back((int32_t)(int16_t)site_len[k]); k_roff.push_back(roff[k]); k_rstride.push_back(rstride[k]); k_g.push_back(site_g[k]); }
    }
    k_first[nc] = (uint32_t)k_pos.size();
    const uint32_t NK = (uint32_t)k_pos.size();
    std::vector<int32_t> lk_num(4ull * NK + 4, 0), lk_mq(4ull * NK + 4, 0), lk_q(4ull * NK + 4, 0), lk_total(NK + 1, 0);
    std::vector<unsigned long long> lk_first(4ull * NK + 4, ~0ull);
    SpLinks LK{k_first.data(), k_pos.data(), k_len.data(), k_roff.data(), k_rstride.data(), rpool.data(), lk_num.data(), lk_mq.data(), lk_q.data(), lk_first.data(), lk_total.data()};
    const int32_t bcap = P.max_variant_count_lgs + 4096;
    std::vector<uint16_t> bmark(G + 1, 0);
    std::vector<unsigned long long> bbits(G / 64 + 2, 0);
    auto refresh_marks = [&](uint32_t mask) {
        for (uint64_t g = 0; g < G; ++g) bmark[g] = sp_base_mark((uint32_t)g, soff.data(), sflag.data());
        for (uint64_t w = 0; w < G / 64 + 1; ++w) bbits[w] = sp_base_bits_word(w, G, bmark.data(), mask);
    };
    cs.bmark = bmark.data(); cl.bmark = bmark.data(); cs.bbits = bbits.data(); cl.bbits = bbits.data();
    std::vector<uint8_t> lbytes((size_t)bcap + 16);
    std::vector<SpEntry> lents(1u << 16);
    // the remembered orientation of every kept site
    std::vector<std::vector<SpHostSite>> hs(nc);
    auto host_sites = [&](uint32_t ct) {
        std::vector<SpHostSite> v;
        for (uint32_t k = site_first[ct], kk = k_first[ct]; k < site_first[ct + 1]; ++k) {
            if (!keep[k]) continue;
            SpHostSite h;
            memset(&h, 0, sizeof(h));
            h.pos = site_pos[k]; h.left = site_left[k]; h.right = site_right[k]; h.len = k_len[kk];
            h.flag = sflag[soff[site_g[k]]];
            h.total = lk_total[kk];
            for (int t = 0; t < 4; ++t) { h.num[t] = lk_num[4ull * kk + t]; h.mapqual[t] = lk_mq[4ull * kk + t]; h.qual[t] = lk_q[4ull * kk + t]; h.first[t] = lk_first[4ull * kk + t]; }
            v.push_back(h);
            ++kk;
        }
        return v;
    };
    for (uint32_t ct = 0; ct < nc && stop >= 4; ++ct) {
        if (k_first[ct + 1] - k_first[ct] <= 1) continue;
        const uint32_t g0 = vs->ctg_off[ct];
        // P10a: short-read links
        refresh_marks(F_SNP);
        std::vector<SpHostSite> h = host_sites(ct);
        std::vector<int32_t> reg = sp_link_regions(h, P.read_len, F_SNP);
        for (size_t i = 0; i + 1 < reg.size(); i += 2) {
            const int64_t rb = (int64_t)vs->read_begin[ct], re = (int64_t)vs->read_begin[ct + 1];
            for (int64_t r = kc_lower_bound_pos(cs.R, rb, re, reg[i] - cs.max_span); r < re; ++r) {
                if (cs.R.pos[r] >= reg[i + 1] + 1) break;
                if (end_s[r] <= reg[i] || lev_s[r] != 2) continue;
                sp_link_record(cs, P, LK, r, ct, reg[i], reg[i + 1], 0, (unsigned long long)(i / 2) << 32 | (unsigned long long)(r - rb), lents.data(), (uint32_t)lents.size(),
                               lbytes.data(), bcap);
            }
        }
        if (err) return (int)err;
        // marks, long-read regions
        h = host_sites(ct);
        for (auto& m : sp_link_marks(h, P.min_count_snp_link)) sflag[soff[g0 + (uint32_t)m.first]] |= m.second;
        refresh_marks(F_LEFT | F_RIGHT);
        h = host_sites(ct);
        reg = sp_link_regions(h, P.max_variant_count_lgs, 0);
        for (size_t i = 0; i + 1 < reg.size(); i += 2) {
            const int64_t rb = (int64_t)vl->read_begin[ct], re = (int64_t)vl->read_begin[ct + 1];
            for (int64_t r = kc_lower_bound_pos(cl.R, rb, re, reg[i] - cl.max_span); r < re; ++r) {
                if (cl.R.pos[r] >= reg[i + 1] + 1) break;
                if (end_l[r] <= reg[i] || lev_l[r] != 1) continue;
                sp_link_record(cl, P, LK, r, ct, reg[i], reg[i + 1], 1, 1ull << 63 | (unsigned long long)(i / 2) << 32 | (unsigned long long)(r - rb), lents.data(),
                               (uint32_t)lents.size(), lbytes.data(), bcap);
            }
        }
        if (err) return (int)err;
        // the chain, then the writes
        h = host_sites(ct);
        std::vector<int8_t> choice;
        if (!sp_chain(h, P.ploidy, &choice)) return (int)ERR_SP_UNDEFINED;
        for (uint32_t t = 0; t < (uint32_t)h.size(); ++t) {
            if (choice[t] < 0) continue;
            const uint32_t kk = k_first[ct] + t;
            const uint8_t* reg_b = rpool.data() + k_roff[kk] + (uint32_t)choice[t] * k_rstride[kk];
            const uint32_t sb = soff[k_g[kk]];
            if (k_len[kk] == 1) sbase[sb] = reg_b[0];
            else for (uint32_t s = sb; s < soff[k_g[kk] + 1]; ++s) sbase[s] = reg_b[s - sb];
        }
    }
    if (err) return (int)err;
    std::vector<uint16_t> slot_res(S + 64);
    for (uint32_t s = 0; s < S; ++s) slot_res[s] = (uint16_t)(sbase[s] | sflag[s] << 8);
    std::vector<uint32_t> opos(S + 1);
    uint32_t o = 0;
    for (uint32_t s = 0; s < S; ++s) { opos[s] = o; o += (slot_res[s] & 0xff) != 3; }
    opos[S] = o;
    char* buf = (char*)calloc(1, (size_t)o + 1);
    for (uint32_t s = 0; s < S; ++s) emit_slot(s, slot_res.data(), slot_info.data(), opos.data(), F_THIRD, (uint8_t*)buf);
    for (uint32_t ct = 0; ct <= nc; ++ct) bounds[ct] = opos[soff[vs->ctg_off[ct]]];
    *out = buf;
    return 0;
}
