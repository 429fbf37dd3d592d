// Per-lane bodies of the score_chain kernels, shared between the HIP kernels (np1_kernels.hip)
// and the host-side lockstep model the CPU test-suite uses to check the staged algorithm against
// the oracle without a GPU (tests/model/np1_model.cpp -- test infrastructure, never shipped/loaded
// by the product).  Everything here is integer/byte logic; citations point at the reference lines
// each body restates.
#pragma once
#include <string.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NP1_HD __host__ __device__ __forceinline__
#else
#define NP1_HD inline
struct uint4 { uint32_t x, y, z, w; };
#endif

namespace np1k {

// Decoded record stream in HBM (mirror of np::ReadStream; 32 B of fixed fields per record).
struct ReadsDev {
    const int32_t* pos;
    const uint32_t* ctg;
    const uint16_t* flag;
    const uint32_t* n_cigar;
    const int32_t* l_qseq;
    const uint64_t* cigar_off;
    const uint64_t* seq_off;
    const uint32_t* cigar;
    const uint8_t* seq;
};

// slot_info bits: low nibble = draft symbol (nt16 code; 3 = DEL for insertion columns)
constexpr uint32_t SI_INSERT = 0x10, SI_LOWER = 0x20, SI_LAST = 0x40, SI_FIRST = 0x80;
// k_vote geometry: a wave owns 62 consecutive slots; lanes 0,1 are left-context halo
constexpr uint32_t VOTE_CH = 62;
// context lists: 16 / 64 / 160 entries per slot in LDS, then every possible one (3 symbols of 4 bits) in an HBM scratch list
constexpr int VOTE_E_ALL = 4096;
// DP record: [slot][n<<16|total][refk | hdr<<16][n x (kmer<<16|count)][8 words state kmers][fmax base | mask<<8]
constexpr uint32_t FLAG_ALL_RECORDS = 0x100;   // or-ed into the tile kernel's flag_single argument: every slot spills a DP record
constexpr uint32_t REC_SINGLE = 1, REC_CTG_LAST = 2, REC_CTG_FIRST = 4;   // hdr bits; hdr bits 4..7 = previous slot's draft symbol
constexpr uint32_t REC_FIXED_WORDS = 12;
enum { CNT_POOL = 0, CNT_HEADS = 1, CNT_REDO = 2, CNT_ERR = 3, CNT_REDO2 = 4, CNT_OVFDESC = 5,
       CNT_STAT_EVENTS = 6, CNT_STAT_FALLBACK = 7,   // statistics of the event kernels
       CNT_POOL_S0 = 8, CNT_HEADS_S0 = 16, CNT_REDO3 = 24, CNT_WORDS = 32 };   // fused pipeline: pool / run-head counters sharded 8 ways
constexpr uint32_t POOL_SHARDS = 8;
constexpr uint32_t ERR_DOUBLE_INS = 1, ERR_BAD_RECORD = 2, ERR_CTX_OVERFLOW = 4, ERR_POOL_OVERFLOW = 8,
                   ERR_DP_INCONSISTENT = 16, ERR_DESC_OVERFLOW = 32;

NP1_HD uint32_t cig_op(uint32_t c) { return c & 0xf; }
NP1_HD int32_t cig_len(uint32_t c) { return (int32_t)(c >> 4); }
NP1_HD uint32_t seq_nib(const uint8_t* s, int32_t i) { return (s[i >> 1] >> ((~i & 1) << 2)) & 0xf; }   // bam_seqi

NP1_HD void np1_atomic_max(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(p, v);
#else
    if (*p < v) *p = v;
#endif
}
NP1_HD void np1_atomic_min(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (*p > v) *p = v;
#endif
}
NP1_HD void np1_atomic_or(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}

// the reference's strtobase table has 100 entries (source/lib/base.c:6-15)
NP1_HD uint32_t draft_code(uint32_t ch_upper) {
    switch (ch_upper) {
        case '=': return 0;  case 'A': return 1;  case 'C': return 2;  case 'M': return 3;
        case 'G': return 4;  case 'R': return 5;  case 'S': return 6;  case 'V': return 7;
        case 'T': return 8;  case 'W': return 9;  case 'Y': return 10; case 'H': return 11;
        case 'K': return 12; case 'D': return 13; case 'B': return 14;
        default: return 15;
    }
}

// ---------------------------------------------------------------------------------------------
// prep: filter (contig.c:667-677), trimmed query window (contig.c:333-358), insertion-column
// max-reduce (contig.c:202-245).  Output window qs > qe means "votes on nothing".
NP1_HD void prep_record(const ReadsDev& R, int64_t r, const uint32_t* ctg_off, int trim, int32_t* qs_out,
                        int32_t* qe_out, int32_t* span_out, uint32_t* ins, uint32_t* counters) {
    uint32_t flag = R.flag[r], ncig = R.n_cigar[r];
    int32_t qs = 1, qe = 0, span = -1;   // span -1: filtered record
    if ((flag & 0xC04) == 0 && ncig > 0) {
        const uint32_t* cg = R.cigar + R.cigar_off[r];
        const uint8_t* seq = R.seq + R.seq_off[r];
        int32_t lq = R.l_qseq[r];
        uint32_t c0 = cg[0], cl = cg[ncig - 1];
        qs = trim + (cig_op(c0) == 4 ? cig_len(c0) : 0);
        qe = lq - trim - (cig_op(cl) == 4 ? cig_len(cl) : 0) - 1;
        if (trim > 0) {   // homopolymer trim; the unbounded reference loops can only empty the window when they leave the read
            bool dead = false;
            for (;;) {
                if (qs >= lq) { dead = true; break; }
                if (seq_nib(seq, qs) != seq_nib(seq, qs - 1)) break;
                ++qs;
            }
            while (!dead) {
                if (qe < 0 || qe + 1 >= lq) { dead = qe < qs; break; }
                if (seq_nib(seq, qe) != seq_nib(seq, qe + 1)) break;
                --qe;
            }
            if (dead) { qs = 1; qe = 0; }
        }
        if (qs > qe) { qs = 1; qe = 0; }
        uint32_t c = R.ctg[r];
        uint32_t g0 = ctg_off[c];
        int32_t L = (int32_t)(ctg_off[c + 1] - g0);
        int32_t pos0 = R.pos[r], pos = pos0, last_ins_pos = -1;
        for (uint32_t i = 0; i < ncig; ++i) {
            uint32_t op = cig_op(cg[i]);
            int32_t len = cig_len(cg[i]);
            if (op == 0 || op == 2) {
                pos += len;
            } else if (op == 1) {
                if (pos > 0 && pos <= L - 1) np1_atomic_max(&ins[g0 + pos - 1], (uint32_t)len);
                if (pos == last_ins_pos) np1_atomic_or(&counters[CNT_ERR], ERR_DOUBLE_INS);
                last_ins_pos = pos;
            }
        }
        span = pos - pos0;
        if (pos > L || pos0 < 0) np1_atomic_or(&counters[CNT_ERR], ERR_BAD_RECORD);
    }
    qs_out[r] = qs;
    qe_out[r] = qe;
    span_out[r] = span;
}

// per draft base: slot_info for the base and its insertion columns (contig.c:81-102: lowercase input
// sets FLAG_ZERO; contig.c:230-234: new insertion columns copy the flag of their base)
NP1_HD void slotinfo_base(const uint8_t* draft, uint32_t g, uint32_t g_first, uint32_t g_end, const uint32_t* soff,
                          uint8_t* slot_info, uint32_t* slot_g = nullptr) {
    uint32_t ch = draft[g];
    uint32_t lower = 0;
    if (ch >= 97 && ch <= 122) { ch -= 32; lower = SI_LOWER; }
    uint32_t s0 = soff[g], s1 = soff[g + 1];
    uint32_t info = draft_code(ch) | lower;
    if (g == g_first) info |= SI_FIRST;
    if (g + 1 == g_end) info |= SI_LAST;
    slot_info[s0] = (uint8_t)info;
    for (uint32_t s = s0 + 1; s < s1; ++s) slot_info[s] = (uint8_t)(3u | SI_INSERT | lower);
    if (slot_g)
        for (uint32_t s = s0; s < s1; ++s) slot_g[s] = g;   // draft index of every slot (fused pipeline)
}

// where a record's symbol row starts in slot space, and its byte capacity
NP1_HD void rowcap_record(const ReadsDev& R, int64_t r, const uint32_t* ctg_off, const uint32_t* soff, const int32_t* qs,
                          const int32_t* qe, const int32_t* span, uint32_t* rbase, uint32_t* cap_bytes) {
    uint32_t rb = 0, cb = 0;
    if (span[r] >= 0 && qs[r] <= qe[r]) {
        uint32_t c = R.ctg[r];
        uint32_t g0 = ctg_off[c], g1 = ctg_off[c + 1];
        int32_t pos = R.pos[r];
        uint32_t g = g0 + (uint32_t)pos;
        rb = pos > 0 ? soff[g - 1] + 1 : soff[g];
        uint32_t gend = g + (uint32_t)span[r];
        if (gend > g1) gend = g1;
        uint32_t cap = soff[gend] - rb;
        cb = (((cap + 1) >> 1) + 3u) & ~3u;   // 4-bit symbols, rows 4-byte aligned
    }
    rbase[r] = rb;
    cap_bytes[r] = cb;
}

struct RowWriter {
    uint32_t* roww;   // the record's row, 4-byte words (global memory in the staged pipeline, LDS in the fused one)
    uint32_t rbase, cur_word, w, sfirst, slast;
    bool any;
    NP1_HD void init(uint32_t* row_words, uint32_t row_base_slot) {
        roww = row_words; rbase = row_base_slot; any = false; sfirst = 1; slast = 0; cur_word = 0; w = 0;
    }
    NP1_HD void emit(uint32_t slot, uint32_t sym) {
        uint32_t n = slot - rbase;
        uint32_t wi = n >> 3;
        if (!any) { any = true; sfirst = slot; cur_word = wi; w = 0; }
        else if (wi != cur_word) {
            roww[cur_word] = w;
            cur_word = wi;
            w = 0;
        }
        w |= sym << ((n & 7) * 4);
        slast = slot;
    }
    NP1_HD void flush() {
        if (any) roww[cur_word] = w;
    }
};

// plain accessors used by the staged pipeline and the host model
struct SoGlobal {
    const uint32_t* soff;
    NP1_HD uint32_t operator()(uint32_t gi) const { return soff[gi]; }
};
struct SeqBytes {
    const uint8_t* seq;
    NP1_HD uint32_t operator()(int32_t q) { return seq_nib(seq, q); }
};

// pass 2 (contig.c:247-331 with start = 0, end = L-1): emits the record's symbol for every slot it votes on
// (always a contiguous slot run).  So(gi) = slot offset of global draft index gi; Sq(q) = 4-bit base q of the record.
template <class So, class Sq>
NP1_HD void walk_record(const uint32_t* cg, uint32_t ncig, int32_t pos0, uint32_t g0, int32_t L, int32_t qs, int32_t qe,
                        So so, Sq sq, RowWriter& wr) {
    int32_t pos = pos0, qpos = 0;
    uint32_t last = 1;   // BAM_CINS
    for (uint32_t i = 0; i < ncig; ++i) {
        const uint32_t op = cig_op(cg[i]);
        const int32_t len = cig_len(cg[i]);
        if (op == 0 || op == 2) {
            // columns j of this op that lie inside the contig and inside the trimmed query window
            int32_t jlo = 0, jhi = len - 1;
            if (-pos > jlo) jlo = -pos;
            if (L - 1 - pos < jhi) jhi = L - 1 - pos;
            if (op == 0) {
                if (qs - qpos > jlo) jlo = qs - qpos;
                if (qe - qpos < jhi) jhi = qe - qpos;
            } else if (qpos < qs || qpos > qe) {
                jhi = -1;
            }
            if (jlo <= jhi) {
                uint32_t sprev = 0;
                for (int32_t j = jlo; j <= jhi; ++j) {
                    const int32_t p = pos + j, q = op == 0 ? qpos + j : qpos;
                    const uint32_t scur = so(g0 + (uint32_t)p);
                    bool pad;
                    if (j > jlo) {
                        pad = true;   // inside a run the previous column was emitted by this very op
                    } else {
                        const uint32_t lastj = j > 0 ? op : last;
                        pad = lastj != 1 && p > 0 && (q > qs || (q == qs && lastj == 2));
                        if (pad) sprev = so(g0 + (uint32_t)p - 1);
                    }
                    if (pad)
                        for (uint32_t s = sprev + 1; s < scur; ++s) wr.emit(s, 3u);   // unused insertion columns vote DEL
                    wr.emit(scur, op == 2 ? 3u : sq(q));
                    sprev = scur;
                }
            }
            if (len > 0) last = op;
            pos += len;
            if (op == 0) qpos += len;
        } else if (op == 1) {
            if (pos != 0) {
                const bool inr = pos > 0 && pos <= L - 1;
                if (inr) {
                    const uint32_t sprev = so(g0 + (uint32_t)pos - 1), scur = so(g0 + (uint32_t)pos);
                    int32_t jlo = 0, jhi = len - 1;
                    if (qs - qpos > jlo) jlo = qs - qpos;
                    if (qe - qpos < jhi) jhi = qe - qpos;
                    for (int32_t j = jlo; j <= jhi; ++j) wr.emit(sprev + 1 + (uint32_t)j, sq(qpos + j));
                    const int32_t qafter = qpos + len;
                    if (qafter > qs && qafter <= qe + 1)
                        for (uint32_t s = sprev + 1 + (uint32_t)len; s < scur; ++s) wr.emit(s, 3u);
                }
                qpos += len;
                last = 1;
            } else {   // insertion before the first base of the contig: skipped, the window shifts (contig.c:315-319)
                qpos += len;
                qs += len;
                last = 1;
            }
        } else if (op == 4 || op == 5) {
            qpos += len;   // hard clips advance the query cursor too (contig.c:321-324)
        }
        if (pos > L - 1) break;
    }
    wr.flush();
}

// staged pipeline: one record -> its row in the global row pool + vote-chunk bounds; returns the number of votes
NP1_HD uint32_t rows_record(const ReadsDev& R, int64_t r, const uint32_t* ctg_off, const uint32_t* soff,
                            const int32_t* qs_in, const int32_t* qe_in, const uint32_t* rbase, const uint64_t* rowoff,
                            uint8_t* rows, uint4* meta, uint32_t* chunk_first, uint32_t* chunk_last) {
    int32_t qs = qs_in[r], qe = qe_in[r];
    uint4 m;
    m.x = 1; m.y = 0; m.z = 0; m.w = 0;
    uint32_t votes = 0;
    if (qs <= qe) {
        uint32_t c = R.ctg[r];
        uint32_t g0 = ctg_off[c];
        RowWriter wr;
        wr.init(reinterpret_cast<uint32_t*>(rows + rowoff[r]), rbase[r]);
        walk_record(R.cigar + R.cigar_off[r], R.n_cigar[r], R.pos[r], g0, (int32_t)(ctg_off[c + 1] - g0), qs, qe,
                    SoGlobal{soff}, SeqBytes{R.seq + R.seq_off[r]}, wr);
        if (wr.any) {
            m.x = wr.sfirst;
            m.y = wr.slast;
            m.z = wr.rbase;
            m.w = (uint32_t)(rowoff[r] >> 2);
            votes = wr.slast - wr.sfirst + 1;
            uint32_t c0 = wr.sfirst / VOTE_CH, c1 = (wr.slast + 2) / VOTE_CH;
            for (uint32_t cc = c0; cc <= c1; ++cc) {
                np1_atomic_min(&chunk_first[cc], (uint32_t)r);
                np1_atomic_max(&chunk_last[cc], (uint32_t)r);
            }
        }
    }
    meta[r] = m;
    return votes;
}

// ---------------------------------------------------------------------------------------------
// vote: one lane's context histogram.  Entries 0/1 live in registers, the rest in a lane-strided
// scratch list (LDS on the device).  First-seen order is the insertion order (base.c:60-71).
template <int E>
struct VoteLane {
    uint32_t k0, c0, k1, c1, n;
    bool ovf;
    NP1_HD void init(uint32_t refk) { k0 = refk; c0 = 1; k1 = 0xffffu; c1 = 0; n = 1; ovf = false; }
    NP1_HD void tally(uint32_t k, uint32_t* L, int lane) {
        if (k == k0) ++c0;
        else if (k == k1) ++c1;
        else if (n == 1) { k1 = k; c1 = 1; n = 2; }
        else {
            bool found = false;
            for (uint32_t e = 2; e < n; ++e) {
                uint32_t v = L[(e - 2) * 64 + lane];
                if ((v >> 16) == k) {
                    L[(e - 2) * 64 + lane] = (v & 0xffff0000u) | ((v + 1) & 0xffffu);
                    found = true;
                    break;
                }
            }
            if (!found) {
                if (n < (uint32_t)E) { L[(n - 2) * 64 + lane] = k << 16 | 1u; ++n; }
                else ovf = true;
            }
        }
    }
    NP1_HD uint32_t total(const uint32_t* L, int lane) const {
        uint32_t t = c0 + c1;
        for (uint32_t e = 2; e < n; ++e) t += L[(e - 2) * 64 + lane] & 0xffffu;
        return t & 0xffffu;   // uint16 counter of the reference (base.h:44)
    }
    NP1_HD void write_record(uint32_t* rec, uint32_t slot, uint32_t tot, uint32_t hdr, const uint32_t* L, int lane) const {
        rec[0] = slot;
        rec[1] = n << 16 | tot;
        rec[2] = k0 | hdr << 16;
        rec[3] = k0 << 16 | (c0 & 0xffffu);
        if (n > 1) rec[4] = k1 << 16 | (c1 & 0xffffu);
        for (uint32_t e = 2; e < n; ++e) rec[3 + e] = L[(e - 2) * 64 + lane];
    }
};

// ---------------------------------------------------------------------------------------------
// chain DP over one multi-state run (contig.c:424-496).  Scores are exact integers: value * 2^K with
// rate = Rfix / 2^K.  `St` provides sc(buf,b), km(buf,b), rk(buf,b) lvalues for the two 16-entry state
// buffers (LDS on the device, plain arrays in the host model).
//
// FP = true is the general-rate path: `rate` is any double (indel_balance_factor_sgs = 0.3, 0.55 ...), scores are the
// reference's own doubles evaluated in its order -- score = S0; score += count - total * rate (contig.c:448, no contraction) --
// and the "run" is the WHOLE CONTIG from its first slot (every slot carries a record): with rounding in play a run can no
// longer start from 0 independently of what precedes it.  The 64-bit state cells then hold the bit patterns of the doubles.
template <bool FP> struct DpScore;
template <> struct DpScore<false> {
    typedef long long T;
    static NP1_HD T get(long long cell) { return cell; }
    static NP1_HD long long put(T v) { return v; }
    static NP1_HD T step(T s0, uint32_t cnt, uint32_t tot, int K, long long Rfix, double) { return s0 + ((long long)cnt << K) - (long long)tot * Rfix; }
};
template <> struct DpScore<true> {
    typedef double T;
    static NP1_HD T get(long long cell) { T v; memcpy(&v, &cell, 8); return v; }
    static NP1_HD long long put(T v) { long long c; memcpy(&c, &v, 8); return c; }
    static NP1_HD T step(T s0, uint32_t cnt, uint32_t tot, int, long long, double rate) {
        const double prod = (double)(int)tot * rate;      // total * rate
        const double inc = (double)(int)cnt - prod;       // count - total * rate
        return s0 + inc;                                  // score += ...
    }
};

template <bool FP, class St>
NP1_HD bool dp_run(uint32_t head_off, uint32_t* pool, const uint32_t* slot_rec, uint16_t* slot_res, int K,
                   long long Rfix, double rate, double min_ratio, St& st) {
    typedef DpScore<FP> SC;
    typedef typename SC::T score_t;
    uint32_t* rec = pool + head_off;
    const uint32_t s_head = rec[0];
    uint32_t hdr = rec[2] >> 16;
    int cur = 0;
    // predecessor: all-zero seed at a contig start (contig.c:459-464), else the single state of slot-1 at score 0
    bool seed = (hdr & REC_CTG_FIRST) != 0;
    uint32_t pmask = 0;
    score_t pfmax = 0;
    if (!seed) {
        uint32_t pb = (hdr >> 4) & 0xf;
        pmask = 1u << pb;
        st.sc(1, pb) = SC::put((score_t)0);
    }
    uint32_t s = s_head;
    bool ok = true;
    for (uint32_t guard = 0;; ++guard) {   // ---- forward
        if (guard > (FP ? 0xfffffff0u : (1u << 26))) { ok = false; break; }
        const uint32_t n = rec[1] >> 16, total = rec[1] & 0xffffu, refk = rec[2] & 0xffffu;
        hdr = rec[2] >> 16;
        const uint32_t tot = total > 1 ? total - 1 : total;
        uint32_t cmask = 0, ncur = 0;
        const int prv = cur ^ 1;
        for (uint32_t e = 0; e < n; ++e) {
            const uint32_t ent = rec[3 + e];
            const uint32_t k = ent >> 16;
            uint32_t cnt = ent & 0xffffu;
            const uint32_t p = (k >> 4) & 0xf;
            score_t S0 = 0;
            if (!seed) {
                if (p == 0) S0 = pfmax;
                else if (pmask >> p & 1u) S0 = SC::get(st.sc(prv, p));
                else ok = false;   // the reference would dereference NULL here
            }
            if (k == refk && total > 1) cnt = (cnt - 1) & 0xffffu;
            const score_t v = SC::step(S0, cnt, tot, K, Rfix, rate);
            const uint32_t b = k & 0xf;
            if (k != 0) {
                if (!(cmask >> b & 1u)) {
                    cmask |= 1u << b;
                    st.sc(cur, b) = SC::put(v); st.km(cur, b) = (uint16_t)k; st.rk(cur, b) = (uint8_t)ncur++;
                } else if (SC::get(st.sc(cur, b)) < v) {
                    st.sc(cur, b) = SC::put(v); st.km(cur, b) = (uint16_t)k;
                }
            } else {   // base_get_score(cur, 0) is base_max_score(cur) (base.c:171-178)
                bool take = ncur == 0;
                if (!take) {
                    score_t best = 0; uint32_t br = 0xff;
                    for (uint32_t bb = 0; bb < 16; ++bb)
                        if (cmask >> bb & 1u) {
                            score_t x = SC::get(st.sc(cur, bb)); uint32_t rr = st.rk(cur, bb);
                            if (br == 0xff || x > best || (x == best && rr < br)) { best = x; br = rr; }
                        }
                    take = best < v;
                }
                if (take) {
                    if (!(cmask & 1u)) { cmask |= 1u; st.rk(cur, 0) = (uint8_t)ncur++; }
                    st.sc(cur, 0) = SC::put(v); st.km(cur, 0) = 0;
                }
            }
        }
        // first strict maximum in insertion order (base.c:185-197)
        score_t best = 0; uint32_t br = 0xff, bbase = 0;
        for (uint32_t bb = 0; bb < 16; ++bb)
            if (cmask >> bb & 1u) {
                score_t x = SC::get(st.sc(cur, bb)); uint32_t rr = st.rk(cur, bb);
                if (br == 0xff || x > best || (x == best && rr < br)) { best = x; br = rr; bbase = bb; }
            }
        uint32_t* scr = rec + 3 + n;   // per-slot backtrace table
        for (uint32_t w = 0; w < 8; ++w) {
            uint32_t lo = (cmask >> (2 * w) & 1u) ? st.km(cur, 2 * w) : 0u;
            uint32_t hi2 = (cmask >> (2 * w + 1) & 1u) ? st.km(cur, 2 * w + 1) : 0u;
            scr[w] = lo | hi2 << 16;
        }
        scr[8] = bbase | cmask << 8;
        if ((!FP && (hdr & REC_SINGLE)) || (hdr & REC_CTG_LAST)) break;   // FP: the run is the whole contig
        uint32_t noff = slot_rec[s + 1];
        if (noff == 0xffffffffu) { ok = false; break; }
        seed = false;
        pmask = cmask;
        pfmax = best;
        cur ^= 1;
        ++s;
        rec = pool + noff;
    }
    if (!ok) return false;
    // ---- backward (contig.c:473-496)
    uint32_t b = rec[3 + (rec[1] >> 16) + 8] & 0xffu;   // terminator: its only state / contig end: first maximum
    for (;;) {
        const uint32_t n = rec[1] >> 16, total = rec[1] & 0xffffu;
        hdr = rec[2] >> 16;
        const uint32_t* scr = rec + 3 + n;
        const uint32_t kk = (scr[b >> 1] >> ((b & 1) * 16)) & 0xffffu;
        if (!(hdr & REC_SINGLE)) {
            uint32_t cntb = 0;
            for (uint32_t e = 0; e < n; ++e) {
                uint32_t ent = rec[3 + e];
                if (((ent >> 16) & 0xf) == b) cntb += ent & 0xffffu;
            }
            uint32_t fl = total == 1 ? 1u : 0u;                        // FLAG_ZERO
            if ((double)cntb / (double)total < min_ratio) fl |= 2u;    // FLAG_COVERAGE (base.c:79-89)
            slot_res[s] = (uint16_t)(b | fl << 8);
        }
        if (s == s_head) break;
        --s;
        rec = pool + slot_rec[s];
        const uint32_t arg = kk >> 4;
        if (arg) b = arg & 0xf;
        else b = rec[3 + (rec[1] >> 16) + 8] & 0xffu;
    }
    return true;
}

// contig_region_correct never revisits base `start` when it owns insertion columns (contig.c:473-496 with
// contig_data_pre, contig.c:402-422): those slots keep their initial base and flag.
NP1_HD void fixfirst_contig(uint32_t g0, uint32_t g1, const uint32_t* soff, const uint8_t* slot_info, uint16_t* slot_res) {
    if (g1 - g0 < 2) return;
    uint32_t s0 = soff[g0], s1 = soff[g0 + 1];
    if (s1 - s0 < 2) return;
    for (uint32_t s = s0; s < s1; ++s) {
        uint32_t info = slot_info[s];
        slot_res[s] = (uint16_t)((info & 0xf) | ((info & SI_LOWER) ? 1u << 8 : 0u));
    }
}

// one output character (contig.c:736-786): lowercase when flagged, or when a flagged deleted slot directly
// precedes it ("sign")
NP1_HD void emit_slot(uint32_t s, const uint16_t* slot_res, const uint8_t* slot_info, const uint32_t* opos, uint32_t mask,
                      uint8_t* out) {
    uint32_t r = slot_res[s];
    uint32_t b = r & 0xff;
    if (b == 3) return;
    bool lower = ((r >> 8) & mask) != 0;
    if (!lower) {
        uint32_t t = s;
        while (t > 0 && !(slot_info[t] & SI_FIRST)) {
            --t;
            uint32_t rr = slot_res[t];
            if ((rr & 0xff) != 3) break;
            if ((rr >> 8) & mask) { lower = true; break; }
        }
    }
    const char* tbl = "=ACMGRSVTWYHKDBN";
    char ch = tbl[b & 0xf];
    out[opos[s]] = (uint8_t)(lower ? ch + 32 : ch);
}

}  // namespace np1k
