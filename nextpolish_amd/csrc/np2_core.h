// Long-read consensus (nextpolish2) -- per-lane bodies shared by the HIP kernels (np2_kernels.hip) and the host
// lockstep model used by the tests (tests/model/np2_model.cpp).  Nothing here allocates or does I/O.
//
// Vocabulary (reference: source/lib/ctg_cns.c / ctg_cns.h):
//   window   [s, e) slice of one contig that is polished in one go (ctg_cns.c:3455-3460)
//   column   one alignment column of a read against the window: (t_pos, delta) with delta = 0 for a draft base and
//            1, 2, ... for the insertion columns after it
//   tag      4-bit code of one column of one read: 3-bit base (A0 T1 G2 C3 -4 N5 M6) + 8 when the column is an
//            insertion column; streams end with nibble 15 (get_align_tags, ctg_cns.c:1213-1256)
//   node     (t_pos, delta, base); entry = one distinct predecessor pair (pp, ppp) of a node with its link count,
//            kept in first-seen order (update_msa, ctg_cns.c:324-365)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NP2_HD __host__ __device__ __forceinline__
#else
#define NP2_HD inline
#endif

namespace np2k {

constexpr int READS_ONT = 1, READS_CLR = 2, READS_HIFI = 3, READS_RS = 4;   // ctg_cns.c:23-26

// ---- character tables (ctg_cns.c:47-66, htslib seq_nt16_str)
NP2_HD char nt16_char(uint32_t c) { return "=ACMGRSVTWYHKDBN"[c & 15]; }
NP2_HD uint32_t base_to_int(unsigned char c) {   // A0 T1 G2 C3, N5, M6 (upper case only for M/N), everything else 4
    switch (c) {
        case 'A': case 'a': return 0;
        case 'T': case 't': return 1;
        case 'G': case 'g': return 2;
        case 'C': case 'c': return 3;
        case 'N': return 5;
        case 'M': return 6;
        default: return 4;
    }
}
NP2_HD char int_to_base(uint32_t b) { return "ATGC-NM"[b]; }

// ---- one read of a window: BAM core fields the path touches
struct ReadView {
    int32_t pos;             // 0-based leftmost contig coordinate
    uint32_t n_cigar;
    const uint32_t* cigar;   // BAM-encoded ops
    const uint8_t* seq;      // 4-bit packed bases
};

// Iterator over the alignment columns the reference's bam2aln materialises (ctg_cns.c:2403-2456): M, I, D ops give
// columns; S/H advance the query; N advances the reference (and, as in the reference, the reported start).
// t = draft character or '-', q = read character or '-'.
struct ColIter {
    const ReadView* r;
    const char* rf;        // contig characters (window base pointer minus window start: rf[contig coordinate])
    uint32_t op_i, in_op;  // current op and offset inside it
    uint32_t rfi, rdi;     // contig / query cursor of the next column
    NP2_HD void begin(const ReadView* rv, const char* rfseq) {
        r = rv; rf = rfseq; op_i = 0; in_op = 0; rfi = (uint32_t)rv->pos; rdi = 0;
        skip_noncolumn_fwd();
    }
    NP2_HD void skip_noncolumn_fwd() {
        while (op_i < r->n_cigar) {
            const uint32_t c = r->cigar[op_i] & 0xf, n = r->cigar[op_i] >> 4;
            if (c == 0 || c == 1 || c == 2) { if (n) return; }
            else if (c == 4 || c == 5) rdi += n;
            else if (c == 3) rfi += n;
            ++op_i;
        }
    }
    NP2_HD bool done() const { return op_i >= r->n_cigar; }
    // current column
    NP2_HD void get(char* t, char* q) const {
        const uint32_t c = r->cigar[op_i] & 0xf;
        *t = c == 1 ? '-' : rf[rfi];
        *q = c == 2 ? '-' : nt16_char((uint32_t)(r->seq[rdi >> 1] >> ((~rdi & 1) << 2)));
    }
    NP2_HD void next() {
        const uint32_t c = r->cigar[op_i] & 0xf, n = r->cigar[op_i] >> 4;
        if (c != 1) ++rfi;
        if (c != 2) ++rdi;
        if (++in_op >= n) { in_op = 0; ++op_i; skip_noncolumn_fwd(); }
    }
};

// the same columns from the last one backwards
struct ColIterRev {
    const ReadView* r;
    const char* rf;
    int32_t op_i;          // current op (-1 = before the first)
    uint32_t left;         // columns left in the current op (counting the current one)
    uint32_t rfi, rdi;     // contig / query coordinate of the CURRENT column's base (valid when the op consumes it)
    NP2_HD void begin(const ReadView* rv, const char* rfseq, uint32_t rf_end, uint32_t rd_end) {
        r = rv; rf = rfseq; op_i = (int32_t)rv->n_cigar - 1; left = 0; rfi = rf_end; rdi = rd_end;
        enter();
    }
    NP2_HD void enter() {   // position on the last column of op_i, skipping ops without columns
        while (op_i >= 0) {
            const uint32_t c = r->cigar[op_i] & 0xf, n = r->cigar[op_i] >> 4;
            if ((c == 0 || c == 1 || c == 2) && n) {
                left = n;
                if (c != 1) --rfi;
                if (c != 2) --rdi;
                return;
            }
            if (c == 4 || c == 5) rdi -= n;
            else if (c == 3) rfi -= n;
            --op_i;
        }
    }
    NP2_HD bool done() const { return op_i < 0; }
    NP2_HD void get(char* t, char* q) const {
        const uint32_t c = r->cigar[op_i] & 0xf;
        *t = c == 1 ? '-' : rf[rfi];
        *q = c == 2 ? '-' : nt16_char((uint32_t)(r->seq[rdi >> 1] >> ((~rdi & 1) << 2)));
    }
    NP2_HD void prev() {
        const uint32_t c = r->cigar[op_i] & 0xf;
        if (--left == 0) { --op_i; enter(); return; }
        if (c != 1) --rfi;
        if (c != 2) --rdi;
    }
};

// totals of a CIGAR: alignment columns, contig and query bases consumed (clips included in the query total)
NP2_HD void cigar_totals(const ReadView& r, uint32_t* n_cols, uint32_t* rf_len, uint32_t* rd_len, bool* bad_op) {
    uint32_t cols = 0, rf = 0, rd = 0;
    bool bad = false;
    for (uint32_t i = 0; i < r.n_cigar; ++i) {
        const uint32_t c = r.cigar[i] & 0xf, n = r.cigar[i] >> 4;
        if (c == 0) { cols += n; rf += n; rd += n; }
        else if (c == 1) { cols += n; rd += n; }
        else if (c == 2) { cols += n; rf += n; }
        else if (c == 3) rf += n;
        else if (c == 4 || c == 5) rd += n;
        else bad = true;   // '=', 'X', 'P', 'B': the reference aborts ("unexpected cigar", ctg_cns.c:2450-2452)
    }
    *n_cols = cols; *rf_len = rf; *rd_len = rd; *bad_op = bad;
}

// Result of clip_aln + get_align_shift(k = 8) (ctg_cns.c:2809-2826,139-201) expressed on column indices of the
// unclipped alignment: the kept columns are [col0, col0 + aln_len).
struct AlnSpan {
    uint32_t col0, aln_len;
    uint32_t aln_t_s, aln_t_e;   // contig coordinates: first kept draft base, one past the last
    uint32_t aln_q_s;            // query coordinate of the first kept column (the reference tracks it for split reads only)
};

// s, e: window [s, e) in contig coordinates.  Literal restatement incl. the unsigned arithmetic and the
// "aln_len = 10" outcome of a clip that leaves 500 columns or fewer.
NP2_HD AlnSpan align_span(const ReadView& r, const char* rfseq, int32_t s, int32_t e, uint32_t q0 = 0) {
    uint32_t N, rf_len, rd_len;
    bool bad;
    cigar_totals(r, &N, &rf_len, &rd_len, &bad);
    AlnSpan a;
    a.col0 = 0;
    a.aln_len = N;
    // aln_t_s as bam2aln leaves it: N ops before/inside shift the reported start too (ctg_cns.c:2416-2419)
    uint32_t nskip = 0;
    for (uint32_t i = 0; i < r.n_cigar; ++i)
        if ((r.cigar[i] & 0xf) == 3) nskip += r.cigar[i] >> 4;
    a.aln_t_s = (uint32_t)r.pos + nskip;
    a.aln_t_e = (uint32_t)r.pos + rf_len;
    a.aln_q_s = q0;
    char t, q;
    if ((int64_t)a.aln_t_s < s || (int64_t)a.aln_t_e > e) {   // clip_aln (signed compare: the reference compares uint with int32 -> unsigned; positions are < 2^31)
        uint32_t s_ = 0;
        ColIter f;
        f.begin(&r, rfseq);
        while ((int64_t)a.aln_t_s < s && !f.done()) {
            f.get(&t, &q);
            if (t != '-') ++a.aln_t_s;
            if (q != '-') ++a.aln_q_s;
            f.next();
            ++s_;
        }
        while (!f.done()) {
            f.get(&t, &q);
            if (t != '-') break;
            f.next();
            ++s_;
        }
        int64_t e_ = (int64_t)N - 1;
        ColIterRev b;
        b.begin(&r, rfseq, (uint32_t)r.pos + rf_len, rd_len);
        while ((int64_t)a.aln_t_e > e && !b.done()) {
            b.get(&t, &q);
            if (t != '-') --a.aln_t_e;
            b.prev();
            --e_;
        }
        if (e_ > (int64_t)s_ + 500) {
            a.col0 = s_;
            a.aln_len = (uint32_t)(e_ - s_ + 1);
        } else {
            a.col0 = 0;
            a.aln_len = 10 < N ? 10 : N;   // (the reference reads 10 columns of its buffer; shorter alignments cannot pass the 500 bp test anyway)
        }
    }
    // get_align_shift(aln, 8, l): first and last run of eight matching columns
    const uint32_t k = 8;
    {
        ColIter f;
        f.begin(&r, rfseq);
        for (uint32_t i = 0; i < a.col0; ++i) f.next();
        uint32_t i = 0, j = 0;
        bool found = false;
        const uint32_t len0 = a.aln_len;
        while (i < len0) {
            f.get(&t, &q);
            if (t == q) ++j; else j = 0;
            if (t != '-') ++a.aln_t_s;
            if (q != '-') ++a.aln_q_s;
            if (j == k) {
                a.aln_t_s -= k;
                a.aln_q_s -= k;
                a.col0 += i - k + 1;
                a.aln_len = len0 - i + k - 1;
                found = true;
                break;
            }
            f.next();
            ++i;
        }
        if (!found) { a.aln_len = 0; return a; }
    }
    {
        // backwards from the last kept column; may run through the whole alignment (ctg_cns.c:170-197)
        uint32_t cols_after = N - (a.col0 + a.aln_len);   // columns behind the kept range
        ColIterRev b;
        b.begin(&r, rfseq, (uint32_t)r.pos + rf_len, rd_len);
        for (uint32_t i = 0; i < cols_after; ++i) b.prev();
        int64_t i = (int64_t)a.aln_len - 1;   // index relative to col0
        uint32_t j = 0, tcount = 0;
        while (i >= 0) {
            b.get(&t, &q);
            if (t == q) ++j; else j = 0;
            if (t != '-') --a.aln_t_e;
            if (j == k) {
                a.aln_t_e += k;
                a.aln_len = a.aln_len - tcount + k - 1;
                break;
            }
            b.prev();
            --i;
            ++tcount;
        }
    }
    return a;
}

// per-column pileup statistics (msa_p, ctg_cns.h:142-148)
struct ColStat {
    uint16_t max_size, coverage, l_del, l_ins;
};

// columns of an already gapped pair of strings (the concatenated low-quality regions, ctg_cns.c:1287-1414)
struct StrColIter {
    const char* t;
    const char* q;
    uint32_t i;
    NP2_HD void get(char* tc, char* qc) const { *tc = t[i]; *qc = q[i]; }
    NP2_HD void next() { ++i; }
};

// get_align_tags (ctg_cns.c:1213-1256) over aln_len columns delivered by `f`.  tags: nibble stream (first column in
// the high nibble), zero-initialised by the caller with (aln_len + 1) / 2 + 1 bytes.  Adds the stream to the column
// statistics through St (atomic on the device).  tpos: window-relative position of the first draft base;
// gap_min_len: 3 ONT, 5 else.  Returns the window-relative exclusive end.
// state of the tag emission between two columns (checkpoints let lanes own chunks of a stream)
struct EmitState {
    uint32_t te;      // window-relative position of the last draft column seen (tpos - 1 before the first)
    uint32_t delta;   // length of the open insertion run
    uint32_t l;       // the open insertion run was already counted in l_ins
};
// columns [p0, p0 + n) of a stream (p = column index inside the stream = nibble index); `last`: this call ends the
// stream and writes the terminator
template <class It, class St>
NP2_HD void emit_tags_range(It& f, uint32_t p0, uint32_t n, EmitState* es, bool last, uint32_t gap_min_len, uint8_t* tags, St& st) {
    uint32_t te = es->te, delta = es->delta, l = es->l, p = p0;
    char t, q;
    for (; p < p0 + n; ++p) {
        f.get(&t, &q);
        uint32_t b = base_to_int((unsigned char)q);
        if (t == '-') { b |= 8; ++delta; }
        else { ++te; l = 0; delta = 0; }
        tags[p >> 1] |= (uint8_t)((p & 1) ? b : b << 4);
        if (delta == 0 && q != 'M') st.coverage(te);
        st.max_size(te, delta);                     // max_size = max(max_size, delta + 1) (16-bit in the reference)
        if (delta >= gap_min_len && !l) { st.l_ins(te); l = 1; }
        if (delta == 0 && q == '-') st.l_del(te);
        f.next();
    }
    if (last) {
        if ((p - 1) & 1) tags[p >> 1] |= 255;
        else tags[p >> 1] |= 15;
    }
    es->te = te; es->delta = delta; es->l = l;
}
template <class It, class St>
NP2_HD uint32_t emit_tags_from(It& f, uint32_t aln_len, uint32_t tpos, uint32_t gap_min_len, uint8_t* tags, St& st) {
    EmitState es{tpos - 1, 0, 0};
    emit_tags_range(f, 0u, aln_len, &es, true, gap_min_len, tags, st);
    return es.te + 1;
}

// the kept columns of one record (win_s: window start in contig coordinates)
template <class St>
NP2_HD uint32_t emit_tags(const ReadView& r, const char* rfseq, const AlnSpan& a, int32_t win_s, uint32_t gap_min_len,
                          uint8_t* tags, St& st) {
    ColIter f;
    f.begin(&r, rfseq);
    for (uint32_t i = 0; i < a.col0; ++i) f.next();
    return emit_tags_from(f, a.aln_len, a.aln_t_s - (uint32_t)win_s, gap_min_len, tags, st);
}

// ---- tag stream walker (get_align_tag, ctg_cns.c:304-322)
struct Tag {
    int32_t t_pos;
    uint32_t delta;   // uint16 in the reference
    uint32_t q_base;
};
NP2_HD bool next_tag(const uint8_t* tags, uint32_t aln_t_s, uint32_t* p, Tag* tag) {
    uint32_t t = tags[*p >> 1];
    if (!(*p & 1)) t >>= 4;
    if ((t & 15) == 15) return false;
    tag->q_base = t & 7;
    if ((*p)++) {
        if (t & 8) tag->delta = (tag->delta + 1) & 0xffffu;
        else { tag->delta = 0; ++tag->t_pos; }
    } else {
        tag->t_pos = (int32_t)aln_t_s;
        tag->delta = 0;
    }
    return true;
}

// packed identity of a node / predecessor: t_pos (32, -1 = stream head) | delta (16) | base (8)
NP2_HD uint64_t node_key(int32_t t_pos, uint32_t delta, uint32_t base) {
    return (uint64_t)(uint32_t)t_pos << 24 | (uint64_t)(delta & 0xffffu) << 8 | (base & 0xffu);
}
constexpr uint64_t KEY_HEAD = ((uint64_t)0xffffffffu << 24);   // align_tag_head: t_pos -1, delta 0, base 0 (ctg_cns.c:52-56)
NP2_HD int32_t key_tpos(uint64_t k) { return (int32_t)(uint32_t)(k >> 24); }
NP2_HD uint32_t key_delta(uint64_t k) { return (uint32_t)(k >> 8) & 0xffffu; }
NP2_HD uint32_t key_base(uint64_t k) { return (uint32_t)k & 0xffu; }

// one link observation: read `rd` visits node (t_pos, delta, base) coming from pp, ppp
struct LinkObs {
    uint64_t pp, ppp;
    uint32_t rd;
    uint16_t delta;
    uint8_t base, pad;
};

// entry of a node's predecessor list (msa_p_d_b_pp_ppp, ctg_cns.h:118-124); score is int64:48 in the reference
struct Entry {
    uint64_t pp, ppp;
    long long score;
    uint32_t link;       // uint16 in the reference (wraps at 65536)
    uint32_t node;       // delta << 8 | base : which node of the column the entry belongs to
};
// node of a column: its entries are entries[start .. start + len), `best` = index of the best predecessor
// (the reference re-uses msa_p_d_b.max_size for it, ctg_cns.c:2070)
struct Node {
    uint32_t key;        // delta << 8 | base
    uint32_t start, len, best;
};

// Builds the nodes and entries of ONE column from its link observations, which must be ordered by (rd, delta)
// (= the order update_msa sees them).  entries/nodes: output regions with room for n items each.
// Returns the number of nodes; nodes come out ordered by (delta, base), entries grouped by node in first-seen order.
NP2_HD uint32_t build_column(const LinkObs* obs, uint32_t n, Entry* entries, Node* nodes) {
    // pass 1: distinct nodes in (delta, base) order (insertion into a small sorted table)
    uint32_t nn = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t key = (uint32_t)obs[i].delta << 8 | obs[i].base;
        uint32_t j = 0;
        while (j < nn && nodes[j].key < key) ++j;
        if (j < nn && nodes[j].key == key) { ++nodes[j].len; continue; }
        for (uint32_t m = nn; m > j; --m) nodes[m] = nodes[m - 1];
        nodes[j].key = key; nodes[j].len = 1; nodes[j].start = 0; nodes[j].best = 0;
        ++nn;
    }
    // node regions sized by their observation counts (an upper bound of the distinct entries)
    uint32_t off = 0;
    for (uint32_t j = 0; j < nn; ++j) { nodes[j].start = off; off += nodes[j].len; nodes[j].len = 0; }
    // pass 2: first-seen lists with link counts
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t key = (uint32_t)obs[i].delta << 8 | obs[i].base;
        uint32_t j = 0;
        while (nodes[j].key != key) ++j;
        Entry* e = entries + nodes[j].start;
        uint32_t m = 0;
        for (; m < nodes[j].len; ++m)
            if (e[m].pp == obs[i].pp && e[m].ppp == obs[i].ppp) { e[m].link = (e[m].link + 1) & 0xffffu; break; }
        if (m == nodes[j].len) {
            e[m].pp = obs[i].pp; e[m].ppp = obs[i].ppp; e[m].link = 1; e[m].score = 0; e[m].node = key;
            ++nodes[j].len;
        }
    }
    return nn;
}

// ---- link graph of a window as the kernels keep it: per column p the nodes nodes[col_off[p] .. + col_nn[p]) in
// (delta, base) order and their entries entries[col_off[p] + node.start .. + node.len)
// Predecessor entries of an entry, resolved once (the keys never change, only the scores do): the entries en of the
// node named by pp with en.pp == ppp, as offsets from that node's first entry, in list order.  n > MATCH_INLINE means
// "more than fit": the consumer scans the node's list like the reference does.
constexpr uint32_t MATCH_INLINE = 4;
struct EMatch {
    uint32_t pe0;        // global index of the predecessor node's first entry
    uint16_t n;          // number of matching entries (0: stream head, missing node or no match)
    uint16_t ps0;        // state index (live entries before it in its column) of the predecessor node's first entry
    uint64_t m4;         // the offsets of up to MATCH_INLINE matching entries, 16 bits each (no indexed array: stays in registers)
    NP2_HD uint32_t at(uint32_t k) const { return (uint32_t)(m4 >> (16 * k)) & 0xffffu; }
};
struct MsaView {
    const uint32_t* col_off;   // len + 2 offsets (column bucket starts; shared by nodes[] and entries[])
    const uint32_t* col_nn;    // nodes per column
    Node* nodes;
    Entry* entries;
    const ColStat* stat;
    const EMatch* match = nullptr;   // per entry (same index as entries[]), optional: without it every lookup searches
};
NP2_HD Node* find_node(const MsaView& m, int32_t t_pos, uint32_t key) {
    Node* nd = m.nodes + m.col_off[t_pos];
    const uint32_t nn = m.col_nn[t_pos];
    if (nn > 16) {   // a column under a long insertion holds thousands of nodes, sorted by key = (delta, base)
        uint32_t lo = 0, hi = nn;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (nd[mid].key < key) lo = mid + 1;
            else hi = mid;
        }
        return lo < nn && nd[lo].key == key ? nd + lo : nullptr;
    }
    for (uint32_t j = 0; j < nn; ++j)
        if (nd[j].key == key) return nd + j;
    return nullptr;
}

// the predecessor entries of entry `em` (global index g): from the resolved match list when there is one, else the whole
// list of the predecessor node (the caller then tests en.pp == em.ppp itself)
struct PredList {
    const Entry* PE = nullptr;
    uint32_t cnt = 0;
    bool listed = false;
    EMatch mt;
    NP2_HD void open(const MsaView& m, const Entry& em, uint32_t g) {
        if (m.match) {
            mt = m.match[g];
            if (mt.n <= MATCH_INLINE) { listed = true; cnt = mt.n; PE = m.entries + mt.pe0; return; }
        }
        Node* ppn = find_node(m, key_tpos(em.pp), key_delta(em.pp) << 8 | key_base(em.pp));
        cnt = ppn ? ppn->len : 0u;
        PE = ppn ? m.entries + m.col_off[key_tpos(em.pp)] + ppn->start : nullptr;
    }
};

// One column of the chain DP of get_cns_from_align_tags (ctg_cns.c:1876-2125), literal per read type: entry score
// = max(0, best matching predecessor entry + 10 * link - C * coverage) (entries start at 0 and are only raised;
// stream heads are assigned directly), per node the index of the best entry by the read-type specific rules, and
// on the last column the global best node (">=": the last one in (delta, base) order wins ties).
template <int TYPE>
NP2_HD void dp_column(const MsaView& m, int32_t p, int32_t len, long long* gbest_score, uint64_t* gbest_key) {
    constexpr long long C = TYPE == READS_HIFI ? 4 : 3;
    const long long cov = m.stat[p].coverage;
    Node* nd = m.nodes + m.col_off[p];
    const uint32_t nn = m.col_nn[p];
    for (uint32_t j = 0; j < nn; ++j) {
        Node& pb = nd[j];
        Entry* E = m.entries + m.col_off[p] + pb.start;
        const uint32_t b = pb.key & 0xffu;
        pb.best = 0;
        long long p_pp_score_ = INT64_MIN, p_pp_score = INT64_MIN;
        int tmp = 0;
        if (TYPE == READS_ONT)
            for (uint32_t mi = 0; mi < pb.len; ++mi)
                if ((int)E[mi].link > tmp) tmp = (int)E[mi].link;
        for (uint32_t mi = 0; mi < pb.len; ++mi) {
            Entry& em = E[mi];
            if (key_tpos(em.pp) == -1) {
                em.score = 10 * (long long)em.link - C * cov;
            } else {
                PredList pr;
                pr.open(m, em, m.col_off[p] + pb.start + mi);
                for (uint32_t it = 0; it < pr.cnt; ++it) {
                    const Entry& en = pr.PE[pr.listed ? pr.mt.at(it) : it];
                    if (!pr.listed && en.pp != em.ppp) continue;
                    const long long cand = en.score + 10 * (long long)em.link - C * cov;
                    if (cand > em.score) {
                        em.score = cand;
                        p_pp_score_ = en.score;
                    }
                    if (TYPE == READS_CLR || TYPE == READS_HIFI) {
                        if (en.score > p_pp_score || (en.score == p_pp_score && key_base(em.pp) != 4)) {
                            pb.best = mi;
                            p_pp_score = en.score;
                        }
                    } else if (TYPE == READS_ONT) {
                        const uint32_t ppb = key_base(em.pp), pppb = key_base(em.ppp);
                        if (((key_delta(em.ppp) > 1 || key_delta(em.pp) > 0) &&
                             ((double)em.link > (double)cov * 0.2 || (int)em.link > tmp / 2)) ||
                            ((int)em.link > (int)E[pb.best].link / 2 && en.score > p_pp_score &&
                             (ppb == 4 || ppb == b || pppb == b || ppb == pppb))) {
                            pb.best = mi;
                            p_pp_score = en.score;
                        }
                    }
                }
            }
            if (TYPE == READS_RS) {
                if (em.score >= E[pb.best].score) { pb.best = mi; p_pp_score = p_pp_score_; }
            } else if (em.score > E[pb.best].score || (em.score == E[pb.best].score && key_base(em.pp) != 4)) {
                pb.best = mi;
                p_pp_score = p_pp_score_;
            }
        }
        if (pb.len && p == len - 1 && E[pb.best].score >= *gbest_score) {
            *gbest_key = node_key(p, pb.key >> 8, b);
            if (E[pb.best].score > *gbest_score) *gbest_score = E[pb.best].score;
        }
    }
}

// Column of the DP variant used on the concatenated low-quality regions (get_lqseqs_from_align_tags, non-HiFi branch,
// ctg_cns.c:1043-1094): coefficient 2, its own best-predecessor rule, no global best (the caller starts the backtrace
// at the last node of the last column).
template <bool kHifi>
NP2_HD void dp_column_lq(const MsaView& m, int32_t p) {
    constexpr long long C = kHifi ? 4 : 2;   // HiFi branch: ctg_cns.c:998-1042
    const long long cov = m.stat[p].coverage;
    Node* nd = m.nodes + m.col_off[p];
    const uint32_t nn = m.col_nn[p];
    for (uint32_t j = 0; j < nn; ++j) {
        Node& pb = nd[j];
        Entry* E = m.entries + m.col_off[p] + pb.start;
        const uint32_t b = pb.key & 0xffu;
        pb.best = 0;
        long long p_pp_score_ = INT64_MIN, p_pp_score = INT64_MIN;
        for (uint32_t mi = 0; mi < pb.len; ++mi) {
            Entry& em = E[mi];
            if (key_tpos(em.pp) == -1) {
                em.score = 10 * (long long)em.link - C * cov;
            } else {
                PredList pr;
                pr.open(m, em, m.col_off[p] + pb.start + mi);
                for (uint32_t it = 0; it < pr.cnt; ++it) {
                    const Entry& en = pr.PE[pr.listed ? pr.mt.at(it) : it];
                    if (!pr.listed && en.pp != em.ppp) continue;
                    const long long cand = en.score + 10 * (long long)em.link - C * cov;
                    if (cand > em.score) {
                        em.score = cand;
                        p_pp_score_ = en.score;
                    }
                    const uint32_t ppb = key_base(em.pp), pppb = key_base(em.ppp);
                    if (kHifi) {
                        if (en.score > p_pp_score || (en.score == p_pp_score && ppb != 4)) {
                            pb.best = mi;
                            p_pp_score = en.score;
                        }
                    } else if ((int)em.link > (int)E[pb.best].link / 2 && en.score > p_pp_score &&
                               (ppb == 4 || ppb == b || pppb == b || ppb == pppb)) {
                        pb.best = mi;
                        p_pp_score = en.score;
                    }
                }
            }
            if (em.score > E[pb.best].score || (em.score == E[pb.best].score && key_base(em.pp) != 4)) {
                pb.best = mi;
                p_pp_score = p_pp_score_;
            }
        }
    }
}

}  // namespace np2k
