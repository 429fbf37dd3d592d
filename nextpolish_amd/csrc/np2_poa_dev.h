// Pseudo-seed of a low-quality region on the device: partial-order alignment of a few candidate strings and the heaviest path
// through the graph, ONE WAVE PER REGION (reference: source/lib/dag.c:261-405,469-508,555-595,658-694 poa_to_consensus; the
// decisions that fix its output are listed at the top of np2_poa.cpp, whose host version this follows step by step).
//
// What is parallel and what is not:
//   * a row of the string-against-graph score table depends on the rows of the node's predecessors and, inside the row, on the cell
//     to its left.  The lanes are the columns: the best move out of the predecessor rows is independent per column, and "stay on the
//     node and take a character" is a running maximum: with gap cost g, H(j) = max(H(j-1) + g, O(j)) <=> H(j) - g j = prefix-max of
//     (O(k) - g k), and the offer O(j) is taken exactly where it raises that prefix maximum (ties stay on the node, as the
//     reference's strict comparison does).  One wave scan per 64 columns;
//   * the end node is the first sink with the best full-length score: a wave maximum + the lowest set bit of a ballot;
//   * walking back, threading the string through the graph, re-ordering the graph (depth-first over aligned groups) and the
//     heaviest path are pointer work with a strict visiting order: every lane runs them redundantly on wave-uniform state that
//     lives in LDS (no divergence; reads come back through a scalar broadcast), a few thousand dependent LDS operations per string.
// The score table (rows x columns, 8 bytes a cell) lives in a slice of HBM scratch owned by the resident wave.
// A region whose graph does not fit the LDS arrays is flagged and done by the host version.
#pragma once
#ifndef NP2_POA_HOST_EMU      // tests/model/np2_poa_emu.cpp runs this header on 64 host threads in lockstep and supplies the wave intrinsics itself
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace np2poa {

constexpr int32_t W_MATCH = 1, W_MISMATCH = -2, W_GAP = -2;

// Two size classes of one algorithm (round 5).  A job = the best six candidates of a region at most (np2_lq.cpp: rank_and_seed), typically
// 35-45 characters each, so its graph has a few dozen nodes and its score table a few thousand cells:
//   Small  indices in one byte (127 nodes, 254 edges, strings up to 126 characters, 8 strings), the SCORE TABLE IN LDS as 16-bit scores +
//          16-bit back pointers (3072 cells): 16.9 KB of LDS a wave, 9 regions a CU, and a row of the table costs LDS round trips where the
//          first version (one class, table in a slice of HBM scratch) paid three dependent global round trips and a fence per row;
//   Big    the first version as it was (255 nodes, 511 edges, 255 characters, 64 strings, table in HBM scratch): what a Small wave gives
//          back (graph or table too large) runs here, what this one gives back goes to the host version.
// Scores fit 16 bits: |score| <= 2 (rows + columns) <= 2 (128 + 127); the weights of the heaviest path are multiples of 0.5 below 2^11,
// exact in a float; so both classes compute the same numbers and take the same decisions.
struct Big {
    using ix = uint16_t; using sup_t = unsigned long long; using best_t = double; using path_t = int16_t; using ts_t = int32_t; using tf_t = uint32_t;
    static constexpr uint32_t MAXN = 256, MAXE = 512, MAXLEN = 255, MAXP = MAXN + MAXLEN + 1, MAXS = 1024, NONE = 0xffffu, TAB_LDS = 0, FSHIFT = 16, MAXSTR = 64;
};
struct Small {
    using ix = uint8_t; using sup_t = uint8_t; using best_t = float; using path_t = int8_t; using ts_t = int16_t; using tf_t = uint16_t;
    static constexpr uint32_t MAXN = 127, MAXE = 254, MAXLEN = 126, MAXP = MAXN + MAXLEN + 1, MAXS = 512, NONE = 0xffu, TAB_LDS = 3072, FSHIFT = 8, MAXSTR = 8;
};

template <class C> struct PoaLdsT {
    using ix = typename C::ix;
    uint8_t base[C::MAXN];
    ix in_head[C::MAXN], in_tail[C::MAXN], out_head[C::MAXN], out_tail[C::MAXN];   // edge lists in attachment order (linked through the edges)
    ix col_of[C::MAXN], col_next[C::MAXN];          // aligned group ("column") of a node, next member in joining order
    ix col_head[C::MAXN], col_tail[C::MAXN];        // per group: first (= lowest-numbered) and last member
    ix order[C::MAXN], rank[C::MAXN];
    ix e_src[C::MAXE], e_dst[C::MAXE], e_nin[C::MAXE], e_nout[C::MAXE];
    typename C::sup_t e_sup[C::MAXE];               // bit q: string q runs along this edge
    typename C::path_t path_node[C::MAXP], path_chr[C::MAXP];        // the alignment columns in walk-back order
    ix stack[C::MAXS];
    uint8_t finished[C::MAXN], open[C::MAXN];
    typename C::best_t best_at[C::MAXN];
    typename C::path_t came_from[C::MAXN];
    typename C::ts_t ts[C::TAB_LDS ? C::TAB_LDS : 1];      // Small: the score table and its back pointers
    typename C::tf_t tf[C::TAB_LDS ? C::TAB_LDS : 1];
    // Big: the scores of the last two rows.  Graphs are mostly chains, so the row a new row reads is nearly always the one just made: it is
    // read here, and the wave waits for its stores to HBM only when a row further back is needed (and once before walking back)
    int32_t prev[2][C::TAB_LDS ? 1 : C::MAXLEN + 2];
};
using PoaLds = PoaLdsT<Big>;

struct Job { uint32_t first_str, n_str; unsigned long long out_off; uint32_t out_cap, small; };      // small: 1 = the host found nothing that rules the Small class out

// wave-uniform broadcast of a value every lane has just read from LDS (one scalar copy instead of 64).  EVERY read of wave-uniform LDS
// state goes through uni() / unis(), and always from wave-uniform control flow: that is what lets the host emulation
// (tests/model/np2_poa_emu.cpp) keep its 64 threads in step -- there it is a barrier behind the read.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void glb_sync() {      // the wave's own table rows: written by some lanes, read by others later
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <class C> struct Graph {          // wave-uniform counters; the arrays are in LDS
    static constexpr uint32_t MAXN = C::MAXN, MAXE = C::MAXE, MAXS = C::MAXS, NONE16 = C::NONE;
    using ix_t = typename C::ix;
    using sup_t = typename C::sup_t;
    PoaLdsT<C>* L;
    uint32_t n, ne, ncols;
    bool fail;
    __device__ __forceinline__ uint32_t new_node(uint32_t b, uint32_t col) {      // col == NONE16: a group of its own
        if (n >= MAXN) { fail = true; return 0; }
        const uint32_t v = n++;
        L->base[v] = (uint8_t)b;
        L->in_head[v] = L->in_tail[v] = L->out_head[v] = L->out_tail[v] = (ix_t)NONE16;
        L->col_next[v] = (ix_t)NONE16;
        if (col == NONE16) {
            col = ncols++;
            L->col_head[col] = L->col_tail[col] = (ix_t)v;
        } else {
            const uint32_t t = uni(L->col_tail[col]);
            L->col_next[t] = (ix_t)v;
            L->col_tail[col] = (ix_t)v;
        }
        L->col_of[v] = (ix_t)col;
        lds_sync();
        return v;
    }
    __device__ __forceinline__ void connect(uint32_t a, uint32_t b, uint32_t q) {
        if (ne >= MAXE) { fail = true; return; }
        const uint32_t e = ne++;
        L->e_src[e] = (ix_t)a; L->e_dst[e] = (ix_t)b; L->e_sup[e] = (sup_t)(1ull << q);
        L->e_nin[e] = L->e_nout[e] = (ix_t)NONE16;
        const uint32_t ot = uni(L->out_tail[a]);
        if (ot == NONE16) L->out_head[a] = (ix_t)e; else L->e_nout[ot] = (ix_t)e;
        L->out_tail[a] = (ix_t)e;
        lds_sync();          // a == b cannot happen, but in-list and out-list updates may touch the same edge record
        const uint32_t it = uni(L->in_tail[b]);
        if (it == NONE16) L->in_head[b] = (ix_t)e; else L->e_nin[it] = (ix_t)e;
        L->in_tail[b] = (ix_t)e;
        lds_sync();
    }
    // a run of fresh nodes for characters that align with nothing
    __device__ __forceinline__ void chain(uint32_t q, const char* s, uint32_t cnt, int32_t* first, int32_t* last) {
        for (uint32_t i = 0; i < cnt && !fail; ++i) {
            const uint32_t v = new_node((uint8_t)s[i], NONE16);
            if (*first == -1) *first = (int32_t)v;
            else connect((uint32_t)*last, v, q);
            *last = (int32_t)v;
        }
    }
    __device__ __forceinline__ bool blocked(uint32_t head) const {      // does anything still have to come before this group?
        if (uni(L->in_head[head]) != NONE16) return true;
        for (uint32_t p = uni(L->col_next[head]); p != NONE16; p = uni(L->col_next[p]))
            if (uni(L->in_head[p]) != NONE16) return true;
        return false;
    }
    // emission order (np2_poa.cpp PoGraph::reorder): groups are the units; sources in ascending group number; from each a
    // depth-first walk (successor edges pushed in attachment order, the first member's before the others') emits a group when it
    // is finished, filling the order from the back
    __device__ void reorder() {
        const uint32_t lane = __lane_id();
        for (uint32_t g = lane; g < ncols; g += 64) { L->finished[g] = 0; L->open[g] = 0; }
        lds_sync();
        int32_t slot = (int32_t)n - 1;
        while (slot >= 0 && !fail) {
            int32_t start = -1;
            for (uint32_t g = 0; g < ncols; ++g)
                if (!uni(L->finished[g]) && !blocked(uni(L->col_head[g]))) { start = (int32_t)g; break; }
            if (start < 0) { fail = true; break; }
            for (uint32_t g = lane; g < ncols; g += 64) L->open[g] = 0;
            lds_sync();
            uint32_t sp = 0;
            L->stack[sp++] = (ix_t)start;
            lds_sync();
            while (sp) {
                const uint32_t g = uni(L->stack[--sp]);
                if (uni(L->finished[g])) continue;
                const uint32_t v = uni(L->col_head[g]);
                if (uni(L->open[g])) {          // second visit: everything below is out, emit the group
                    L->finished[g] = 1;
                    L->open[g] = 0;
                    for (uint32_t p = v; p != NONE16; p = uni(L->col_next[p])) {
                        if (slot < 0) { fail = true; break; }
                        L->order[slot--] = (ix_t)p;
                    }
                    lds_sync();
                    continue;
                }
                L->open[g] = 1;
                if (sp >= MAXS) { fail = true; break; }
                L->stack[sp++] = (ix_t)g;
                for (uint32_t p = v; p != NONE16 && !fail; p = uni(L->col_next[p]))
                    for (uint32_t e = uni(L->out_head[p]); e != NONE16; e = uni(L->e_nout[e])) {
                        if (sp >= MAXS) { fail = true; break; }
                        L->stack[sp++] = L->col_of[uni(L->e_dst[e])];
                    }
                lds_sync();
            }
        }
        for (uint32_t i = lane; i < n; i += 64) L->rank[L->order[i]] = (ix_t)i;
        lds_sync();
    }
};

// signed wave-uniform broadcast (path and came_from entries are -1 or an index)
__device__ __forceinline__ int32_t unis(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// The score table of one string against the graph, (nodes + 1) rows x (length + 1) columns, a score and a back pointer (row << FSHIFT |
// column) per cell.  Big: a slice of HBM scratch owned by the resident wave; Small: the wave's LDS.
template <class C> struct Table {
    PoaLdsT<C>* L;
    int32_t* TS;
    uint32_t* TF;
    uint32_t cap;
    uint32_t* dbg;      // debugging (NP2_POA_DEBUG): where this wave is -- {stage, counter} -- readable from the host while the kernel runs
    __device__ __forceinline__ void mark(uint32_t stage, uint32_t v) const {
        if (dbg && __lane_id() == 0) { __atomic_store_n(dbg, stage, __ATOMIC_RELAXED); __atomic_store_n(dbg + 1, v, __ATOMIC_RELAXED); }
    }
    __device__ __forceinline__ uint32_t capacity() const { if constexpr (C::TAB_LDS > 0) return C::TAB_LDS; else return cap; }
    __device__ __forceinline__ int32_t score(uint32_t i) const { if constexpr (C::TAB_LDS > 0) return (int32_t)L->ts[i]; else return TS[i]; }
    __device__ __forceinline__ uint32_t from(uint32_t i) const { if constexpr (C::TAB_LDS > 0) return (uint32_t)L->tf[i]; else return TF[i]; }
    __device__ __forceinline__ void set(uint32_t i, int32_t sc, uint32_t fr) const {
        if constexpr (C::TAB_LDS > 0) { L->ts[i] = (typename C::ts_t)sc; L->tf[i] = (typename C::tf_t)fr; }
        else { TS[i] = sc; TF[i] = fr; }
    }
    __device__ __forceinline__ void sync() const {      // the wave's own rows: written by some lanes, read by others later
        if constexpr (C::TAB_LDS > 0) lds_sync(); else glb_sync();
    }
};

// one string against the graph, then through it (np2_poa.cpp add_string)
template <class C> __device__ void add_string(Graph<C>& G, uint32_t q, const char* s, uint32_t len, const Table<C>& T) {
    constexpr uint32_t NONE16 = C::NONE, MAXP = C::MAXP, FS = C::FSHIFT, FMASK = (1u << C::FSHIFT) - 1u;
    using path_t = typename C::path_t;
    using sup_t = typename C::sup_t;
    PoaLdsT<C>* L = G.L;
    const uint32_t lane = __lane_id();
    const uint32_t n = G.n, width = len + 1;
    if ((unsigned long long)(n + 1) * width > T.capacity()) { G.fail = true; return; }
    constexpr bool BIGC = C::TAB_LDS == 0;
    for (uint32_t c = lane; c < width; c += 64) {
        T.set(c, (int32_t)c * W_GAP, 0u);
        if constexpr (BIGC) L->prev[0][c] = (int32_t)c * W_GAP;
    }
    if constexpr (BIGC) lds_sync(); else T.sync();
    // rows in emission order
    for (uint32_t r = 0; r < n; ++r) {
        T.mark(3, q << 16 | r);
        const uint32_t v = uni(L->order[r]);
        const char vb = (char)uni(L->base[v]);
        const uint32_t row = r + 1;
        const uint32_t R0 = row * width;
        const uint32_t e0 = uni(L->in_head[v]);
        // The rows this row reads (its predecessors' rows; a source reads row 0), found ONCE per row and kept in scalars -- the border,
        // the "anything but the row just made?" test and every 64-column chunk below go over them (round 5: each of those walked the edge
        // list through LDS again, three dependent reads per edge and chunk).  More than four predecessors: the walk is repeated as before.
        uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, n_pred = 0;
        if (e0 == NONE16) { n_pred = 1; }
        else for (uint32_t e = e0; e != NONE16; e = uni(L->e_nin[e])) {
            const uint32_t pr = uni(L->rank[uni(L->e_src[e])]) + 1u;
            if (n_pred == 0) p0 = pr; else if (n_pred == 1) p1 = pr; else if (n_pred == 2) p2 = pr; else if (n_pred == 3) p3 = pr;
            ++n_pred;
        }
        auto for_each_pred = [&](auto&& f) {
            if (n_pred <= 4) {
                f(p0);
                if (n_pred > 1) f(p1);
                if (n_pred > 2) f(p2);
                if (n_pred > 3) f(p3);
            } else {
                for (uint32_t e = e0; e != NONE16; e = uni(L->e_nin[e])) f(uni(L->rank[uni(L->e_src[e])]) + 1u);
            }
        };
        if constexpr (BIGC) {      // does this row read anything but the row just made?  then the earlier rows' stores have to have landed
            bool far = false;
            for_each_pred([&](uint32_t pr) { if (pr != row - 1u) far = true; });
            if (far) glb_sync();
        }
        auto rd = [&](uint32_t pr, uint32_t col) -> int32_t {      // (pr is wave-uniform)
            if constexpr (BIGC) { if (pr == row - 1u) return L->prev[(row - 1u) & 1u][col]; }
            return T.score(pr * width + col);
        };
        // left border: the best predecessor's border value plus a gap (sources start from 0)
        int32_t border = 0;
        if (e0 != NONE16) {
            bool any = false;
            for_each_pred([&](uint32_t pr) {
                const int32_t x = unis(rd(pr, 0));
                if (!any || x > border) { border = x; any = true; }
            });
        }
        border += W_GAP;
        if (lane == 0) {
            T.set(R0, border, 0u);
            if constexpr (BIGC) L->prev[row & 1u][0] = border;
        }
        int32_t run = border;                     // max over k < j of (H(k) + 2 k) = of A(k); A(0) = H(0)
        for (uint32_t cb = 0; cb < len; cb += 64) {
            const uint32_t c = cb + lane;         // string position; the cell is column j = c + 1
            const bool valid = c < len;
            int32_t O = INT32_MIN;
            uint32_t F = 0;
            {
                const int32_t w = (valid && s[c] == vb) ? W_MATCH : W_MISMATCH;
                bool first = true;
                for_each_pred([&](uint32_t pr) {
                    if (!valid) return;
                    const int32_t skip = rd(pr, c + 1) + W_GAP, pair = rd(pr, c) + w;
                    const bool sk = skip >= pair;
                    const int32_t cand = sk ? skip : pair;
                    if (first || cand > O) { O = cand; F = pr << FS | (c + (sk ? 1u : 0u)); }
                    first = false;
                });
            }
            const int32_t j = (int32_t)c + 1;
            const int32_t A = valid ? O - W_GAP * j : INT32_MIN;
            int32_t inc = A;                      // inclusive prefix maximum over the lanes
            for (int d = 1; d < 64; d <<= 1) { const int32_t y = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d && y > inc) inc = y; }
            int32_t exc = __shfl_up(inc, 1, 64);
            if (lane == 0 || exc < run) exc = run;
            if (valid) {
                const bool take = A > exc;
                const int32_t M = take ? A : exc;
                T.set(R0 + (uint32_t)j, M + W_GAP * j, take ? F : (row << FS | (uint32_t)(j - 1)));
                if constexpr (BIGC) L->prev[row & 1u][j] = M + W_GAP * j;
            }
            const int32_t last = __shfl(inc, 63, 64);      // invalid lanes carry INT32_MIN: the maximum of the valid ones
            if (last > run) run = last;
        }
        if constexpr (BIGC) lds_sync(); else T.sync();
    }
    if constexpr (BIGC) glb_sync();      // the sinks' last cells and the back pointers are read from HBM below
    // the alignment ends in a sink: the first one in order with the best full-length score
    uint32_t row = 0;
    {
        int32_t best = 0;
        bool any = false;
        for (uint32_t rb = 0; rb < n; rb += 64) {
            const uint32_t r = rb + lane;
            const bool cand = r < n && L->out_head[L->order[r]] == NONE16;
            const int32_t x = cand ? T.score((r + 1) * width + len) : INT32_MIN;
            int32_t m = x;
            for (int o = 32; o > 0; o >>= 1) { const int32_t y = __shfl_xor(m, o, 64); m = y > m ? y : m; }
            const unsigned long long hit = __ballot(cand && x == m);
            if (hit && (!any || m > best)) { best = m; any = true; row = rb + (uint32_t)__ffsll((long long)hit); }   // = r + 1 of the first lane
        }
    }
    // walk back to the origin; every move contributes a column (node, character, or both)
    uint32_t np = 0;
    int32_t lowest_chr = -1, highest_chr = -1;
    for (uint32_t col = len; row != 0 || col != 0;) {
        T.mark(4, row << 16 | col);
        const uint32_t from = uni(T.from(row * width + col));
        const uint32_t from_row = from >> FS, from_col = from & FMASK;
        int32_t node = -1, chr = -1;
        if (from_row != row) node = (int32_t)uni(L->order[row - 1]);
        if (from_col != col) {
            chr = (int32_t)col - 1;
            lowest_chr = chr;
            if (highest_chr == -1) highest_chr = chr;
        }
        if (np >= MAXP) { G.fail = true; return; }
        L->path_node[np] = (path_t)node;
        L->path_chr[np] = (path_t)chr;
        ++np;
        row = from_row;
        col = from_col;
    }
    lds_sync();
    // ---- thread the string through the graph
    int32_t first = -1, prev = -1, tail_first = -1, cur = -1;
    bool cur_is_new = true, prev_is_new = true;
    if (lowest_chr > 0) G.chain(q, s, (uint32_t)lowest_chr, &first, &prev);                      // characters before the first aligned one
    if (highest_chr < (int32_t)len - 1)                                                         // and behind the last one (the run takes the
        G.chain(q, s + highest_chr + 1, (uint32_t)((int32_t)len - highest_chr), &tail_first, &cur);   // terminator along, like the reference)
    for (uint32_t k = np; k-- > 0 && !G.fail;) {
        T.mark(5, q << 16 | k);
        const int32_t chr = unis((int32_t)L->path_chr[k]);
        if (chr == -1) continue;
        const int32_t node = unis((int32_t)L->path_node[k]);
        cur_is_new = false;
        const uint8_t b = (uint8_t)s[chr];
        if (node == -1) { cur = (int32_t)G.new_node(b, NONE16); cur_is_new = true; }
        else if ((uint8_t)uni(L->base[node]) == b) cur = node;
        else {
            const uint32_t col = uni(L->col_of[node]);
            int32_t same = -1;
            for (uint32_t p = uni(L->col_head[col]); p != NONE16; p = uni(L->col_next[p]))
                if ((int32_t)p != node && (uint8_t)uni(L->base[p]) == b) same = (int32_t)p;
            if (same != -1) cur = same;
            else { cur = (int32_t)G.new_node(b, col); cur_is_new = true; }                      // a new member of the column
        }
        if (prev != -1) {
            bool joined = false;
            if (!cur_is_new && !prev_is_new)
                for (uint32_t e = uni(L->out_head[prev]); e != NONE16; e = uni(L->e_nout[e]))
                    if (uni(L->e_dst[e]) == (uint32_t)cur) { L->e_sup[e] |= (sup_t)(1ull << q); joined = true; }
            if (joined) lds_sync();
            else G.connect((uint32_t)prev, (uint32_t)cur, q);
        }
        prev = cur;
        prev_is_new = cur_is_new;
        if (first == -1) first = prev;
    }
    if (tail_first != -1 && !G.fail) G.connect((uint32_t)prev, (uint32_t)tail_first, q);
    T.mark(6, q << 16 | G.n);
    if (!G.fail) G.reorder();
    T.mark(7, q);
}

// the whole job: strings of one region -> consensus characters; returns false when the region has to go to the host version
template <class C> __device__ bool poa_region(const char* pool, const uint32_t* __restrict__ str_off, const uint32_t* __restrict__ str_len, const Job& J, int32_t* TS, uint32_t* TF,
                                              uint32_t tab_cap, char* out_pool, uint32_t* out_len, PoaLdsT<C>* L, uint32_t* dbg = nullptr) {
    constexpr uint32_t NONE16 = C::NONE, MAXLEN = C::MAXLEN, MAXN = C::MAXN;
    using ix_t = typename C::ix;
    using path_t = typename C::path_t;
    using sup_t = typename C::sup_t;
    using best_t = typename C::best_t;
    const uint32_t lane = __lane_id();
    if (J.n_str > C::MAXSTR) return false;
    Graph<C> G{L, 0u, 0u, 0u, false};
    const Table<C> T{L, TS, TF, tab_cap, dbg};
    T.mark(1, J.first_str);
    // the first string is the graph: a chain, every node a group of its own, emission order = string order
    const uint32_t len0 = str_len[J.first_str];
    const char* s0 = pool + str_off[J.first_str];
    if (len0 == 0 || len0 > MAXLEN || len0 > MAXN) return false;
    for (uint32_t i = lane; i < len0; i += 64) {
        L->base[i] = (uint8_t)s0[i];
        L->in_head[i] = L->in_tail[i] = (ix_t)(i > 0 ? i - 1 : NONE16);
        L->out_head[i] = L->out_tail[i] = (ix_t)(i + 1 < len0 ? i : NONE16);
        L->col_of[i] = (ix_t)i; L->col_next[i] = (ix_t)NONE16;
        L->col_head[i] = L->col_tail[i] = (ix_t)i;
        L->order[i] = L->rank[i] = (ix_t)i;
        if (i + 1 < len0) { L->e_src[i] = (ix_t)i; L->e_dst[i] = (ix_t)(i + 1); L->e_nin[i] = L->e_nout[i] = (ix_t)NONE16; L->e_sup[i] = (sup_t)1; }
    }
    G.n = len0; G.ne = len0 - 1; G.ncols = len0;
    lds_sync();
    for (uint32_t q = 1; q < J.n_str && !G.fail; ++q) {
        const uint32_t len = str_len[J.first_str + q];
        if (len == 0 || len > MAXLEN) return false;
        add_string<C>(G, q, pool + str_off[J.first_str + q], len, T);
    }
    if (G.fail) return false;
    // heaviest path: weight of entering a node over an edge = strings on the edge - half the node's indegree (as uint8); the
    // running best is carried from node to node in emission order, the overall best is the first strict maximum
    int32_t top = -1;
    double carried = -1, top_score = -1;
    for (uint32_t r = 0; r < G.n; ++r) {
        T.mark(8, r);
        const uint32_t v = uni(L->order[r]);
        int32_t from = -1;
        const uint32_t e0 = uni(L->in_head[v]);
        if (e0 != NONE16) {
            uint32_t indeg = 0;
            for (uint32_t e = e0; e != NONE16; e = uni(L->e_nin[e])) ++indeg;
            const double toll = 0.5 * (double)(uint8_t)indeg;
            for (uint32_t e = e0; e != NONE16; e = uni(L->e_nin[e])) {
                const uint32_t src = uni(L->e_src[e]);
                const unsigned long long sup = (unsigned long long)L->e_sup[e];
                const double x = (double)L->best_at[src] + (double)__popcll(sup) - toll;
                if (x > carried || from == -1) { carried = x; from = (int32_t)src; }
            }
        } else {
            carried = 0;
        }
        L->best_at[v] = (best_t)carried;
        L->came_from[v] = (path_t)from;
        lds_sync();
        if (carried > top_score) { top_score = carried; top = (int32_t)v; }
    }
    // characters from the end of the path backwards, then turned over; an embedded terminator (a tail node built from the NUL of
    // a candidate) ends the string
    uint32_t m = 0;
    T.mark(9, (uint32_t)top);
    for (int32_t v = top; v != -1; v = unis((int32_t)L->came_from[v])) ++m;
    T.mark(10, m);
    if (m > J.out_cap) return false;
    char* out = out_pool + J.out_off;
    uint32_t k = m;
    for (int32_t v = top; v != -1; v = unis((int32_t)L->came_from[v])) {
        --k;
        if (lane == 0) out[k] = (char)L->base[v];
    }
    glb_sync();
    uint32_t z = m;
    for (uint32_t cb = 0; cb < m; cb += 64) {
        const unsigned long long nul = __ballot(cb + lane < m && out[cb + lane] == 0);
        if (nul) { z = cb + (uint32_t)__ffsll((long long)nul) - 1u; break; }
    }
    if (lane == 0) *out_len = z;
    T.mark(11, z);
    return true;
}

}  // namespace np2poa
