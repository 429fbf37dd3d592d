// Pseudo-seed of a low-quality region: partial-order alignment of a few candidate strings and the heaviest path through the
// resulting graph -- the HOST version.  The product makes the pseudo-seeds on the device (np2_poa_dev.h: one wave per region, the
// same procedure with the score-table rows spread over the lanes); this file is what the device hands a region to when its graph
// does not fit the kernel's LDS arrays, what NP2_POA_CHECK=1 compares every device result with, and what the tests' host executor
// runs.  It is pinned to the reference's poa_to_consensus by 60 known answers and a fuzz against oracle/_ref.
//
// The result has to equal the reference's poa_to_consensus (source/lib/dag.c:658-694) character for character, and that is
// decided by details of its procedure: which of equal-scoring alignment moves wins (dag.c:261-300), which sink the alignment
// ends in (:302-314), when a node is reused / joins an aligned group / is created (:345-405), the depth-first order in which
// groups are emitted (:469-508; it decides ties of the two scans that walk "the first best in order"), 16-bit node ids, and
// the path weight `support - 0.5 * indegree` in double (:555-595).  Those decisions are kept; the data layout is this file's
// own: structure-of-arrays graph, edge support as one 64-bit mask (<= 50 strings), aligned groups as explicit id lists, a score
// table in two planes (scores, packed back-references) whose per-predecessor part is plain array arithmetic.
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>
#include <vector>

#include "np2_lq.h"

namespace np2 {
namespace {

constexpr int MAX_STRINGS = 50;
constexpr int32_t W_MATCH = 1, W_MISMATCH = -2, W_GAP = -2;

// ---- the alignment graph ----------------------------------------------------------------------------------------------
struct PoGraph {
    // nodes
    std::vector<uint8_t> base;
    std::vector<std::vector<uint32_t>> in, out;       // edge ids in the order they were attached
    std::vector<std::vector<uint16_t>> peers;         // nodes aligned to this one (same column, other base), in joining order
    // edges
    std::vector<uint16_t> src, dst;
    std::vector<uint64_t> support;                    // bit q: string q runs along this edge
    // emission order of the nodes (sources first) and its inverse
    std::vector<uint16_t> order, rank;

    size_t used = 0;                                   // nodes in use (the per-node vectors are kept and reused from region to region)
    uint16_t n_nodes() const { return (uint16_t)used; }
    void clear() { used = 0; src.clear(); dst.clear(); support.clear(); order.clear(); rank.clear(); }
    uint16_t new_node(char b) {
        if (used == base.size()) { base.push_back(0); in.emplace_back(); out.emplace_back(); peers.emplace_back(); }
        base[used] = (uint8_t)b;
        in[used].clear(); out[used].clear(); peers[used].clear();
        return (uint16_t)used++;
    }
    void connect(uint16_t a, uint16_t b, int q) {
        const uint32_t e = (uint32_t)src.size();
        src.push_back(a); dst.push_back(b); support.push_back(1ull << q);
        out[a].push_back(e);
        in[b].push_back(e);
    }
    // a run of fresh nodes for characters that align with nothing: returns through first / last the ends of the run
    void chain(int q, const char* s, size_t n, int32_t* first, int32_t* last) {
        for (size_t i = 0; i < n; ++i) {
            const uint16_t v = new_node(s[i]);
            if (*first == -1) *first = v;
            else connect((uint16_t)*last, v, q);
            *last = v;
        }
    }
    void reorder();
};

// Emission order: aligned groups are the units; a group is a source when its representative (or, only if the representative has
// no incoming edge at all, its peers) has no incoming edge; sources are taken in ascending group number, and from each one a
// depth-first walk (successor edges pushed in attachment order, the representative's before its peers') emits groups when they
// are finished, filling the order from the back.
void PoGraph::reorder() {
    const uint16_t n = n_nodes();
    static thread_local std::vector<int32_t> group_of;      // scratch kept per host thread: this runs after every string of every region
    static thread_local std::vector<uint16_t> head, stack;
    static thread_local std::vector<int8_t> finished, open;
    group_of.assign((size_t)n, -1);
    head.clear();                                      // representative node of every group = its lowest-numbered member seen first
    for (uint16_t v = 0; v < n; ++v) {
        if (group_of[v] != -1) continue;
        const int32_t g = (int32_t)head.size();
        head.push_back(v);
        group_of[v] = g;
        for (uint16_t p : peers[v]) group_of[p] = g;
    }
    const size_t n_groups = head.size();
    finished.assign(n_groups, 0);
    open.assign(n_groups, 0);
    order.assign(n, 0);
    int32_t slot = (int32_t)n - 1;                      // next free place, from the back
    auto blocked = [&](uint16_t v) {                    // does anything still have to come before this group?
        size_t c = in[v].size();
        for (size_t j = 0; j < peers[v].size() && c == 0; ++j) c += in[peers[v][j]].size();
        return c != 0;
    };
    while (slot >= 0) {
        int32_t start = -1;
        for (size_t g = 0; g < n_groups; ++g)
            if (!finished[g] && !blocked(head[g])) { start = (int32_t)g; break; }
        assert(start != -1);
        std::fill(open.begin(), open.end(), 0);
        stack.assign(1, (uint16_t)start);
        while (!stack.empty()) {
            const uint16_t g = stack.back();
            stack.pop_back();
            if (finished[g]) continue;
            const uint16_t v = head[g];
            if (open[g]) {                              // second visit: everything below is out, emit the group
                finished[g] = 1;
                order[(size_t)slot--] = v;
                for (uint16_t p : peers[v]) order[(size_t)slot--] = p;
                open[g] = 0;
                continue;
            }
            open[g] = 1;
            stack.push_back(g);
            for (uint32_t e : out[v]) stack.push_back((uint16_t)group_of[dst[e]]);
            for (uint16_t p : peers[v])
                for (uint32_t e : out[p]) stack.push_back((uint16_t)group_of[dst[e]]);
        }
    }
    rank.assign(n, 0);
    for (uint16_t i = 0; i < n; ++i) rank[order[i]] = i;
}

// ---- string against graph ----------------------------------------------------------------------------------------------
// Score table: row 0 = before any node, row r + 1 = node order[r]; column c = c characters of the string consumed.  A cell keeps
// its score and the cell it was reached from (row << 16 | column; the two border lines point at the origin), in two planes so that
// the part of a row that does not depend on the row itself -- the best move out of the predecessor rows -- is plain array
// arithmetic over the columns.
struct Step { int32_t node, chr; };     // one column of the alignment: graph node and / or string position (-1 = none)

// Best move into (row, c + 1) out of ONE predecessor row P, for every c: skipping this node (gap) or pairing the character with it;
// the gap goes first when the two tie.  first = this is the first predecessor (assign), else it replaces the current offer only
// where it is strictly better.
__attribute__((target_clones("avx2", "default")))
void offers_from(const int32_t* __restrict__ P, const int32_t* __restrict__ w, uint32_t pr, uint32_t len, bool first,
                 int32_t* __restrict__ O, uint32_t* __restrict__ F) {
    const uint32_t tag = pr << 16;
    if (first) {
        for (uint32_t c = 0; c < len; ++c) {
            const int32_t skip = P[c + 1] + W_GAP, pair = P[c] + w[c];
            const bool sk = skip >= pair;
            O[c] = sk ? skip : pair;
            F[c] = tag | (c + (sk ? 1u : 0u));
        }
    } else {
        for (uint32_t c = 0; c < len; ++c) {
            const int32_t skip = P[c + 1] + W_GAP, pair = P[c] + w[c];
            const bool sk = skip >= pair;
            const int32_t cand = sk ? skip : pair;
            const bool better = cand > O[c];
            O[c] = better ? cand : O[c];
            F[c] = better ? (tag | (c + (sk ? 1u : 0u))) : F[c];
        }
    }
}

void add_string(PoGraph& g, int q, const char* s, uint16_t len) {
    const uint16_t n = g.n_nodes();
    const size_t width = (size_t)len + 1;
    static thread_local std::vector<int32_t> tabS, O, W;
    static thread_local std::vector<uint32_t> tabF, F;
    tabS.resize(((size_t)n + 1) * width);          // every cell is written below: no fill
    tabF.resize(((size_t)n + 1) * width);
    O.resize(width); F.resize(width); W.resize(width);
    int32_t* const TS = tabS.data();
    uint32_t* const TF = tabF.data();
    for (size_t c = 0; c < width; ++c) { TS[c] = (int32_t)c * W_GAP; TF[c] = 0; }
    // the rows a row draws from (its in-edges in attachment order, row 0 for a source), resolved once per row
    static thread_local std::vector<uint32_t> prow_beg, prow;
    prow_beg.assign((size_t)n + 1, 0);
    prow.clear();
    for (uint16_t r = 0; r < n; ++r) {
        const uint16_t v = g.order[r];
        prow_beg[r] = (uint32_t)prow.size();
        for (uint32_t e : g.in[v]) prow.push_back((uint32_t)g.rank[g.src[e]] + 1u);
    }
    prow_beg[n] = (uint32_t)prow.size();
    // left border: the best predecessor's border value plus a gap (sources start from 0)
    for (uint16_t r = 0; r < n; ++r) {
        int32_t best = 0;
        bool any = false;
        for (uint32_t k = prow_beg[r]; k < prow_beg[(size_t)r + 1]; ++k) {
            const int32_t x = TS[(size_t)prow[k] * width];
            if (!any || x > best) { best = x; any = true; }
        }
        TS[((size_t)r + 1) * width] = best + W_GAP;
        TF[((size_t)r + 1) * width] = 0;
    }
    // interior.  Moves into (row, c + 1): stay on the node and take a character (the default), or come from a predecessor row
    // either skipping this node's character pairing (gap) or pairing the character with the node.  A move replaces the current
    // choice only if it is strictly better, and of a predecessor's two moves the gap goes first when they tie: so the winner is the
    // first predecessor (in edge order) with the best move, and it beats staying on the node only when it is strictly better.
    for (uint16_t r = 0; r < n; ++r) {
        const char vb = (char)g.base[g.order[r]];
        const uint32_t row = (uint32_t)r + 1;
        int32_t* const RS = TS + (size_t)row * width;
        uint32_t* const RF = TF + (size_t)row * width;
        const uint32_t k0 = prow_beg[r], k1 = prow_beg[(size_t)r + 1];
        for (uint32_t c = 0; c < len; ++c) W[c] = s[c] == vb ? W_MATCH : W_MISMATCH;
        if (k0 == k1) offers_from(TS, W.data(), 0u, len, true, O.data(), F.data());          // a source draws from row 0
        for (uint32_t k = k0; k < k1; ++k) offers_from(TS + (size_t)prow[k] * width, W.data(), prow[k], len, k == k0, O.data(), F.data());
        const uint32_t stay = row << 16;
        for (uint32_t c = 0; c < len; ++c) {
            const int32_t h = RS[c] + W_GAP;
            const bool take = O[c] > h;
            RS[c + 1] = take ? O[c] : h;
            RF[c + 1] = take ? F[c] : (stay | c);
        }
    }
    // the alignment ends in a sink: the first one in order with the best full-length score
    size_t row = 0;
    {
        int32_t best = 0;
        bool any = false;
        for (uint16_t r = 0; r < n; ++r)
            if (g.out[g.order[r]].empty()) {
                const int32_t x = TS[((size_t)r + 1) * width + len];
                if (!any || x > best) { row = (size_t)r + 1; best = x; any = true; }
            }
    }
    // walk back to the origin; every move contributes a column (node, character, or both)
    static thread_local std::vector<Step> path;
    path.clear();
    int64_t lowest_chr = -1, highest_chr = -1;
    for (size_t col = len; row != 0 || col != 0;) {
        const uint32_t from = TF[row * width + col];
        const size_t from_row = from >> 16, from_col = from & 0xffffu;
        Step st{-1, -1};
        if (from_row != row) st.node = g.order[row - 1];
        if (from_col != col) {
            st.chr = (int32_t)col - 1;
            lowest_chr = st.chr;
            if (highest_chr == -1) highest_chr = st.chr;
        }
        path.push_back(st);
        row = from_row;
        col = from_col;
    }
    std::reverse(path.begin(), path.end());
    // ---- thread the string through the graph
    int32_t first = -1, prev = -1, tail_first = -1, cur = -1;
    bool cur_is_new = true, prev_is_new = true;
    if (lowest_chr > 0) g.chain(q, s, (size_t)lowest_chr, &first, &prev);                  // characters before the first aligned one
    if (highest_chr < (int64_t)len - 1)                                                     // and behind the last one (the run takes the
        g.chain(q, s + highest_chr + 1, (size_t)((int64_t)len - highest_chr), &tail_first, &cur);   // terminator along, like the reference)
    for (const Step& st : path) {
        if (st.chr == -1) continue;
        cur_is_new = false;
        const char b = s[st.chr];
        if (st.node == -1) { cur = g.new_node(b); cur_is_new = true; }
        else if ((char)g.base[(size_t)st.node] == b) cur = st.node;
        else {
            int32_t same = -1;
            for (uint16_t p : g.peers[(size_t)st.node])
                if ((char)g.base[p] == b) cur = same = p;                                   // the last peer with this base
            if (same == -1) {                                                               // a new member of the column
                cur = g.new_node(b);
                cur_is_new = true;
                std::vector<uint16_t> col{(uint16_t)st.node};
                col.insert(col.end(), g.peers[(size_t)st.node].begin(), g.peers[(size_t)st.node].end());
                g.peers[(size_t)cur] = col;
                for (uint16_t p : col) g.peers[p].push_back((uint16_t)cur);
            }
        }
        if (prev != -1) {
            bool joined = false;
            if (!cur_is_new && !prev_is_new)
                for (uint32_t e : g.out[(size_t)prev])
                    if (g.dst[e] == (uint16_t)cur) { g.support[e] |= 1ull << q; joined = true; }
            if (!joined) g.connect((uint16_t)prev, (uint16_t)cur, q);
        }
        prev = cur;
        prev_is_new = cur_is_new;
        if (first == -1) first = prev;
    }
    if (tail_first != -1) g.connect((uint16_t)prev, (uint16_t)tail_first, q);
    g.reorder();
}

}  // namespace

std::string poa_consensus(const std::vector<std::string>& seqs) {
    static thread_local PoGraph g;      // kept per host thread, its vectors keep their capacity
    g.clear();
    assert((int)seqs.size() <= MAX_STRINGS);
    for (size_t q = 0; q < seqs.size(); ++q) {
        const std::string& s = seqs[q];
        if (q == 0) {
            int32_t first = -1, last = -1;
            g.chain(0, s.c_str(), s.size(), &first, &last);
            g.order.resize(g.n_nodes());
            g.rank.resize(g.n_nodes());
            for (uint16_t v = 0; v < g.n_nodes(); ++v) g.order[v] = g.rank[v] = v;
        } else {
            add_string(g, (int)q, s.c_str(), (uint16_t)s.size());
        }
    }
    // heaviest path: weight of entering a node over an edge = strings on the edge - half the node's indegree; the running
    // best is carried from node to node in emission order (a node's own best starts from the previous node's value, which is
    // what the reference's shared variable does), the overall best is the first strict maximum
    const uint16_t n = g.n_nodes();
    static thread_local std::vector<double> best_at;
    static thread_local std::vector<int32_t> came_from;
    best_at.assign((size_t)n, 0.0);
    came_from.assign((size_t)n, -1);
    int32_t top = -1;
    double carried = -1, top_score = -1;
    for (uint16_t r = 0; r < n; ++r) {
        const uint16_t v = g.order[r];
        int32_t from = -1;
        if (!g.in[v].empty()) {
            const double toll = 0.5 * (double)(uint8_t)g.in[v].size();
            for (uint32_t e : g.in[v]) {
                const double x = best_at[g.src[e]] + (double)__builtin_popcountll(g.support[e]) - toll;
                if (x > carried || from == -1) { carried = x; from = g.src[e]; }
            }
        } else {
            carried = 0;
        }
        best_at[v] = carried;
        came_from[v] = from;
        if (carried > top_score) { top_score = carried; top = v; }
    }
    std::string out;
    for (int32_t v = top; v != -1; v = came_from[(size_t)v]) out.push_back((char)g.base[(size_t)v]);
    std::reverse(out.begin(), out.end());
    // the reference returns a C string: an embedded terminator (a tail node built from the NUL of a candidate) ends it
    const size_t z = out.find('\0');
    if (z != std::string::npos) out.resize(z);
    return out;
}

}  // namespace np2
