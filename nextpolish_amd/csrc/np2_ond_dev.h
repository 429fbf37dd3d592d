// Candidate-to-seed alignment of the low-quality regions on the device: ONE WAVE PER (candidate, seed) PAIR.
//
// What is computed is the greedy furthest-reaching-path search of Myers' O(ND) difference algorithm with the three rules the
// reference adds to it (source/lib/align.c:39-177): at most 0.4 (|q| + |t|) differences, diagonals whose progress falls more
// than 150 behind the best one are dropped after every round, and a traceback that meets a gap run longer than 250 gives the
// alignment up.  The reference walks the diagonals of a round one after the other; here
//   * the diagonals of a round are the lanes of the wave: every furthest-reaching point of round d depends only on points of
//     round d - 1 on the two neighbouring diagonals, so they are independent; "the first diagonal (ascending) that reaches both
//     ends" is the lowest set bit of a ballot, the pruning bounds are the first / last set bit of another;
//   * both strings sit in LDS; the choice bits of a round (came from the left or from the right diagonal) are one ballot word per
//     64 diagonals, rows indexed by position inside the round's band (the band's first diagonal is kept per round);
//   * the traceback follows its single path, but each diagonal run of matches is measured by the whole wave at once (compare
//     64 positions, count the leading agreements) and written out by the whole wave.
// The gapped strings are stored in traceback order (last column first); the kernel that assembles the concatenated alignments
// reads them backwards.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace np2ond {

constexpr uint32_t STR_CAP = 2048;   // bytes of LDS per string and wave; longer strings are read from HBM

struct Pair {                 // one alignment job
    uint64_t q_off, t_off;    // candidate / seed characters in the string pool
    uint32_t q_len, t_len;
    uint64_t out_off;         // gapped strings (traceback order): t at out_off, q at out_off + out_cap
    uint32_t out_cap;         // q_len + t_len + 2
    uint32_t pad;
};
struct PairResult { int32_t aln_len, aln_t_len, aln_q_len, status; };   // status: 0 ok / no alignment, 1 = scratch too small (never expected)

struct WaveScratch {          // HBM scratch of one resident wave, sized for the largest pair of the launch
    int32_t* V;               // 2 * max_d_cap + 4 furthest-reaching x per diagonal (index k + offset)
    int32_t* band_lo;         // first diagonal of every round
    uint64_t* choice;         // round d: words [d * row_words, ...)
    uint32_t max_d_cap, row_words;
};

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ int32_t bcast(int32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, (int)l); }

struct Strs {                 // the two strings, in LDS when they fit
    const uint8_t* q; const uint8_t* t;
    __device__ __forceinline__ uint8_t qc(int32_t i) const { return q[i]; }
    __device__ __forceinline__ uint8_t tc(int32_t i) const { return t[i]; }
};

// All 64 lanes call it with the same arguments.
__device__ __forceinline__ void align_pair_wave(const uint8_t* pool, const Pair& P, uint8_t* out_pool, PairResult* res, const WaveScratch& W,
                                                uint8_t* lds_q, uint8_t* lds_t) {
    const uint32_t lane = lane_id();
    const int32_t q_len = (int32_t)P.q_len, t_len = (int32_t)P.t_len;
    int32_t max_d = (int32_t)(0.4 * (double)(q_len + t_len));
    const float band_factor = q_len + t_len > 5000 ? 0.1f : 1.0f;
    const int32_t band_size = (int32_t)(band_factor * (float)(q_len + t_len));
    const int32_t koff = max_d;
    PairResult r{0, 0, 0, 0};
    if ((uint32_t)max_d > W.max_d_cap) { r.status = 1; if (lane == 0) *res = r; return; }
    // strings
    Strs S;
    const uint8_t* gq = pool + P.q_off;
    const uint8_t* gt = pool + P.t_off;
    if (P.q_len <= STR_CAP && P.t_len <= STR_CAP) {
        for (uint32_t i = lane; i < P.q_len; i += 64) lds_q[i] = gq[i];
        for (uint32_t i = lane; i < P.t_len; i += 64) lds_t[i] = gt[i];
        S.q = lds_q; S.t = lds_t;
    } else { S.q = gq; S.t = gt; }
    for (int32_t i = (int32_t)lane; i < 2 * max_d + 4; i += 64) W.V[i] = 0;     // the reference clears V before every call (ctg_cns.c:1354)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int32_t min_k = 0, max_k = 0, best_m = -1;
    bool aligned = false;
    int32_t ax = 0, ay = 0, ak = 0, ad = 0;
    for (int32_t d = 0; d < max_d && max_k - min_k <= band_size; ++d) {
        if (lane == 0) W.band_lo[d] = min_k;
        const int32_t nk = max_k >= min_k ? (max_k - min_k) / 2 + 1 : 0;
        int32_t first_ok = 0x7fffffff, last_ok = -0x7fffffff;      // pruning: lowest / highest diagonal that keeps up
        int32_t round_best = best_m;
        // pass 1: the new furthest-reaching points (reads only the other parity of V), kept in registers per 64 diagonals
        for (int32_t base = 0; base < nk && !aligned; base += 64) {
            const int32_t k = min_k + 2 * (base + (int32_t)lane);
            const bool act = base + (int32_t)lane < nk;
            int32_t x = 0;
            bool from_left = false;
            if (act) {
                const int32_t vl = W.V[k - 1 + koff], vr = W.V[k + 1 + koff];
                if (k == min_k || (k != max_k && vl < vr)) x = vr;
                else { x = vl + 1; from_left = true; }
                int32_t y = x - k;
                while (x < q_len && y < t_len && S.qc(x) == S.tc(y)) { ++x; ++y; }
            }
            const int32_t y = x - k;
            const uint64_t done = __ballot(act && x >= q_len && y >= t_len);
            const uint64_t word = __ballot(act && from_left);
            if (lane == 0) W.choice[(size_t)d * W.row_words + (size_t)(base >> 6)] = word;
            if (done) {        // the reference stops the round at the first (lowest) diagonal that reaches both ends
                const uint32_t l0 = (uint32_t)__ffsll((long long)done) - 1u;
                aligned = true;
                ax = bcast(x, l0); ay = bcast(y, l0); ak = min_k + 2 * (base + (int32_t)l0); ad = d;
                break;
            }
            // x + y of this chunk
            int32_t m = act ? x + y : -0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int32_t v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
            round_best = m > round_best ? m : round_best;
            // V of this parity is not read again in this round: store now, test the pruning condition after the round's best is known
            if (act) W.V[k + koff] = x;
        }
        if (aligned) break;
        best_m = round_best;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // pass 2: pruning bounds (align.c:98-113): lowest / highest diagonal of the round with 2 x - k >= best - 150
        for (int32_t base = 0; base < nk; base += 64) {
            const int32_t k = min_k + 2 * (base + (int32_t)lane);
            const bool act = base + (int32_t)lane < nk;
            const bool ok = act && (W.V[k + koff] * 2 - k >= best_m - 150);
            const uint64_t b = __ballot(ok);
            if (b) {
                const int32_t lo = min_k + 2 * (base + (int32_t)__ffsll((long long)b) - 1);
                const int32_t hi = min_k + 2 * (base + 63 - (int32_t)__clzll((long long)b));
                if (lo < first_ok) first_ok = lo;
                if (hi > last_ok) last_ok = hi;
            }
        }
        const int32_t new_min_k = first_ok != 0x7fffffff ? first_ok : max_k;     // the scans start from the far end's value
        const int32_t new_max_k = last_ok != -0x7fffffff ? last_ok : min_k;
        max_k = new_max_k + 1;
        min_k = new_min_k - 1;
    }
    if (!aligned) { if (lane == 0) *res = r; return; }
    // ---- traceback (align.c:115-170)
    uint8_t* out_t = out_pool + P.out_off;
    uint8_t* out_q = out_t + P.out_cap;
    int32_t x = ax - 1, k = ak, d = ad;
    r.aln_t_len = ay;
    r.aln_q_len = x + 1;
    int32_t gap = 0, pos = 0;
    bool bad = false;
    for (;;) {
        // diagonal run of matches ending at x: the whole wave measures and writes it
        for (;;) {
            const int32_t xi = x - (int32_t)lane;
            const bool m = xi >= 0 && xi >= k && S.qc(xi) == S.tc(xi - k);
            const uint64_t mm = __ballot(m);
            const uint32_t run = mm == ~0ull ? 64u : (uint32_t)__ffsll((long long)~mm) - 1u;
            if (lane < run) {
                if ((uint32_t)pos + lane < P.out_cap) { const uint8_t c = S.qc(xi); out_t[pos + (int32_t)lane] = c; out_q[pos + (int32_t)lane] = c; }
            }
            if (run) gap = 0;
            x -= (int32_t)run;
            pos += (int32_t)run;
            if (run < 64) break;
        }
        if (x < 0 && x - k < 0) break;
        if (d < 0 || (uint32_t)pos >= P.out_cap) { bad = true; break; }
        const int32_t idx = (k - W.band_lo[d]) / 2;
        const bool from_left = (W.choice[(size_t)d * W.row_words + (size_t)(idx >> 6)] >> (idx & 63)) & 1ull;
        int32_t pre_k, pre_x;
        if (from_left) { pre_k = k - 1; pre_x = x - 1; }
        else { pre_k = k + 1; pre_x = x; }
        if (!from_left) {          // the path came down the right diagonal: a target character against a gap
            if (x - k < 0) gap = 260;
            else if (lane == 0) { out_q[pos] = '-'; out_t[pos] = S.tc(x - k); }
            if (x - k >= 0) ++pos;
        } else {                   // from the left diagonal: a query character against a gap
            if (x < 0) gap = 260;
            else if (lane == 0) { out_q[pos] = S.qc(x); out_t[pos] = '-'; }
            if (x >= 0) ++pos;
        }
        if (gap++ > 250) { pos = 2; break; }   // "only allow the max length of a gap = 250": the caller treats 2 columns as no alignment
        --d;
        k = pre_k;
        x = pre_x;
    }
    r.aln_len = bad ? 0 : pos;
    if (lane == 0) *res = r;
}

}  // namespace np2ond
