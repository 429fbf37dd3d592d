// The forms a record stream crosses PCIe in: built once per stream on the host, undone on the device by the
// small kernels of np1_kernels.hip (k_unpack_seq2, k_unpack_draft4, k_patch_seq, k_expand_*).  Plain host code, no HIP: the product
// calls the builders from np1_device.hip:stream_facts, and the CPU test-suite runs them against the host restatements of the undo
// kernels at the end of this file (tests/model: np1m_upload_roundtrip) -- what comes back must be the stream's own arrays.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <array>
#include <vector>

#include "np1_priv.h"
#include "np_stream.h"
#include "np_threads.h"

namespace np1up {

// 2 bits per base (A C G T = 4-bit code 1 << k): seq2[j] = codes of seq[2j] (high 4 bits) and seq[2j + 1] (low 4 bits); every byte
// of seq holding anything else (N, ambiguity codes, the pad nibble of an odd-length record) is an exception (byte index, byte).
// false: more than 1 byte in `max_esc_ratio` would be an exception -- nothing is kept.
inline bool build_seq2(const std::vector<uint8_t>& seq, std::vector<uint8_t>* seq2, std::vector<uint64_t>* esc_at, std::vector<uint8_t>* esc_val,
                       size_t max_esc_ratio = 64) {
    seq2->clear(); esc_at->clear(); esc_val->clear();
    const size_t nb = seq.size();
    static const std::array<uint8_t, 256> lut = [] {      // byte of seq -> 4 bits of seq2, 0x80: an exception
        std::array<uint8_t, 256> t{};
        auto c2 = [](uint32_t nib) -> int { return nib == 1 ? 0 : nib == 2 ? 1 : nib == 4 ? 2 : nib == 8 ? 3 : -1; };
        for (uint32_t b = 0; b < 256; ++b) {
            const int h = c2(b >> 4), l = c2(b & 15u);
            t[b] = (h < 0 || l < 0) ? 0x80 : (uint8_t)(h << 2 | l);
        }
        return t;
    }();
    const size_t n2 = (nb + 1) / 2;
    seq2->resize(n2);
    const size_t grain = (size_t)1 << 20;                     // seq2 bytes per block
    const size_t blocks = (n2 + grain - 1) / grain;
    std::vector<std::vector<uint64_t>> e_at(blocks);
    std::vector<std::vector<uint8_t>> e_val(blocks);
    const uint8_t* src = seq.data();
    uint8_t* dst = seq2->data();
    np::parallel_for(blocks, 1, [&](size_t b0, size_t b1) {
        for (size_t blk = b0; blk < b1; ++blk) {
            const size_t j0 = blk * grain, j1 = std::min(n2, j0 + grain);
            for (size_t j = j0; j < j1; ++j) {
                const uint8_t a = lut[src[2 * j]], c = 2 * j + 1 < nb ? lut[src[2 * j + 1]] : 0;
                if (a & 0x80) { e_at[blk].push_back(2 * j); e_val[blk].push_back(src[2 * j]); }
                if (c & 0x80) { e_at[blk].push_back(2 * j + 1); e_val[blk].push_back(src[2 * j + 1]); }
                dst[j] = (uint8_t)((a & 15u) << 4 | (c & 15u));
            }
        }
    });
    size_t n_esc = 0;
    for (const auto& v : e_at) n_esc += v.size();
    if (n_esc * max_esc_ratio > nb) {                         // not worth it: the plain array goes up
        seq2->clear();
        seq2->shrink_to_fit();
        return false;
    }
    esc_at->reserve(n_esc); esc_val->reserve(n_esc);
    for (size_t blk = 0; blk < blocks; ++blk) {
        esc_at->insert(esc_at->end(), e_at[blk].begin(), e_at[blk].end());
        esc_val->insert(esc_val->end(), e_val[blk].begin(), e_val[blk].end());
    }
    return true;
}

// 4 bits per draft character: 2-bit base | lower-case << 2; every other character is an exception (index, character).
inline bool build_draft4(const std::string& draft, std::vector<uint8_t>* draft4, std::vector<uint64_t>* esc_at, std::vector<uint8_t>* esc_val,
                         size_t max_esc_ratio = 32) {
    draft4->clear(); esc_at->clear(); esc_val->clear();
    const size_t G = draft.size();
    const size_t g2 = (G + 1) / 2;
    draft4->resize(g2);
    auto code = [](uint8_t ch) -> int {      // 2-bit base | lower << 2, or -1
        switch (ch) {
            case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
            case 'a': return 4; case 'c': return 5; case 'g': return 6; case 't': return 7;
            default: return -1;
        }
    };
    const size_t grain = (size_t)1 << 20, blocks = (g2 + grain - 1) / grain;
    std::vector<std::vector<uint64_t>> e_at(blocks);
    std::vector<std::vector<uint8_t>> e_val(blocks);
    const uint8_t* src = reinterpret_cast<const uint8_t*>(draft.data());
    uint8_t* dst = draft4->data();
    np::parallel_for(blocks, 1, [&](size_t b0, size_t b1) {
        for (size_t blk = b0; blk < b1; ++blk)
            for (size_t j = blk * grain, j1 = std::min(g2, (blk + 1) * grain); j < j1; ++j) {
                int a = code(src[2 * j]), c = 2 * j + 1 < G ? code(src[2 * j + 1]) : 0;
                if (a < 0) { e_at[blk].push_back(2 * j); e_val[blk].push_back(src[2 * j]); a = 0; }
                if (c < 0) { e_at[blk].push_back(2 * j + 1); e_val[blk].push_back(src[2 * j + 1]); c = 0; }
                dst[j] = (uint8_t)(a << 4 | c);
            }
    });
    size_t n_esc = 0;
    for (const auto& v : e_at) n_esc += v.size();
    if (n_esc * max_esc_ratio > G) {
        draft4->clear();
        draft4->shrink_to_fit();
        return false;
    }
    for (size_t blk = 0; blk < blocks; ++blk) {
        esc_at->insert(esc_at->end(), e_at[blk].begin(), e_at[blk].end());
        esc_val->insert(esc_val->end(), e_val[blk].begin(), e_val[blk].end());
    }
    return true;
}

// The compact form of pos / n_cigar / l_qseq / the operation pool of a DENSE stream (np1_priv.h: np1_stream::Compact); C->on is left
// false -- the caller decides from the sizes whether it pays.
inline void build_compact(const np::ReadStream& s, np1_stream::Compact* Cp) {
    np1_stream::Compact& C = *Cp;
    C = np1_stream::Compact();
    const size_t n = s.n_reads();
    if (n == 0) return;
    {   // the usual read length: the most frequent one among the first records
        const size_t m = std::min<size_t>(n, 4096);
        std::vector<int32_t> sample(s.l_qseq.begin(), s.l_qseq.begin() + m);
        std::sort(sample.begin(), sample.end());
        size_t best = 0, run = 0;
        for (size_t i = 0; i < m; ++i) {
            run = (i > 0 && sample[i] == sample[i - 1]) ? run + 1 : 1;
            if (run > best && sample[i] > 0 && sample[i] < (1 << 28)) { best = run; C.common_lq = (uint32_t)sample[i]; }
        }
    }
    C.plain.assign((n + 31) / 32, 0u);
    C.dpos.resize(n);
    const uint32_t plain_op = C.common_lq << 4;      // <common_lq>M
    size_t ct = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t k = s.n_cigar[i];
        const uint32_t* cg = s.cigar.data() + s.cigar_off[i];
        if (C.common_lq && k == 1 && (uint32_t)s.l_qseq[i] == C.common_lq && cg[0] == plain_op) {
            C.plain[i >> 5] |= 1u << (i & 31u);
        } else {
            C.x_lq.push_back(s.l_qseq[i]);
            C.x_ncig.push_back(k);
            C.x_cigar.insert(C.x_cigar.end(), cg, cg + k);
        }
        while (ct + 1 < s.read_begin.size() && s.read_begin[ct + 1] <= i) ++ct;
        const int64_t d = i > 0 ? (int64_t)s.pos[i] - (int64_t)s.pos[i - 1] : -1;
        if (i == 0 || i == s.read_begin[ct] || d < 0 || d >= 255) { C.dpos[i] = 255; C.x_pos.push_back(s.pos[i]); }
        else C.dpos[i] = (uint8_t)d;
    }
    C.n_ops = s.cigar.size();
}
inline uint64_t compact_bytes(const np1_stream::Compact& C, size_t n) {
    return 4 * (uint64_t)C.plain.size() + n + 4 * (uint64_t)C.x_pos.size() + 8 * (uint64_t)C.x_lq.size() + 4 * (uint64_t)C.x_cigar.size();
}

// ---- host restatements of the kernels that undo the forms (np1_kernels.hip), for the CPU tests ------------------------------------
inline void undo_seq2(const std::vector<uint8_t>& seq2, const std::vector<uint64_t>& esc_at, const std::vector<uint8_t>& esc_val, std::vector<uint8_t>* seq) {
    seq->assign(2 * seq2.size(), 0);
    for (size_t j = 0; j < seq2.size(); ++j) {      // k_unpack_seq2
        const uint32_t b = seq2[j];
        (*seq)[2 * j] = (uint8_t)(((1u << ((b >> 6) & 3u)) << 4) | (1u << ((b >> 4) & 3u)));
        (*seq)[2 * j + 1] = (uint8_t)(((1u << ((b >> 2) & 3u)) << 4) | (1u << (b & 3u)));
    }
    for (size_t i = 0; i < esc_at.size(); ++i) (*seq)[esc_at[i]] = esc_val[i];      // k_patch_seq
}
inline void undo_draft4(const std::vector<uint8_t>& d4, size_t G, const std::vector<uint64_t>& esc_at, const std::vector<uint8_t>& esc_val, std::string* draft) {
    draft->assign(G, '\0');
    const char lut[4] = {'A', 'C', 'G', 'T'};
    for (size_t j = 0; 2 * j < G; ++j) {            // k_unpack_draft4
        const uint32_t v = d4[j];
        (*draft)[2 * j] = (char)(lut[(v >> 4) & 3u] | (((v >> 6) & 1u) << 5));
        if (2 * j + 1 < G) (*draft)[2 * j + 1] = (char)(lut[v & 3u] | (((v >> 2) & 1u) << 5));
    }
    for (size_t i = 0; i < esc_at.size(); ++i) (*draft)[esc_at[i]] = (char)esc_val[i];
}
// launch_expand_records + launch_record_offsets (the running sums) + launch_expand_cigars
inline void undo_compact(const np1_stream::Compact& C, size_t n, std::vector<int32_t>* pos, std::vector<uint32_t>* ncig, std::vector<int32_t>* lq,
                         std::vector<uint32_t>* cigar) {
    pos->assign(n, 0); ncig->assign(n, 0); lq->assign(n, 0);
    std::vector<uint64_t> xidx(n + 1), pidx(n + 1), steps(n + 1), esc_rec(C.x_pos.size());
    uint64_t a = 0, b = 0, c = 0;
    for (size_t i = 0; i < n; ++i) {                // the three exclusive scans
        xidx[i] = a; pidx[i] = b; steps[i] = c;
        a += ((C.plain[i >> 5] >> (i & 31u)) & 1u) ^ 1u;
        b += C.dpos[i] == 255 ? 1u : 0u;
        c += C.dpos[i] == 255 ? 0u : C.dpos[i];
    }
    for (size_t r = 0; r < n; ++r) {                // k_expand_a
        const bool plain = (C.plain[r >> 5] >> (r & 31u)) & 1u;
        if (plain) { (*ncig)[r] = 1u; (*lq)[r] = (int32_t)C.common_lq; }
        else { (*ncig)[r] = C.x_ncig[xidx[r]]; (*lq)[r] = C.x_lq[xidx[r]]; }
        if (C.dpos[r] == 255) esc_rec[pidx[r]] = r;
    }
    for (size_t r = 0; r < n; ++r) {                // k_expand_pos
        const bool esc = C.dpos[r] == 255;
        const uint64_t k = pidx[r] + (esc ? 1u : 0u) - 1u;
        const uint64_t e = esc_rec[k];
        const uint64_t upto_r = steps[r] + (esc ? 0ull : (uint64_t)C.dpos[r]);
        (*pos)[r] = C.x_pos[k] + (int32_t)(upto_r - steps[e]);
    }
    std::vector<uint64_t> cigoff(n + 1), x_cigoff(C.x_ncig.size() + 1);
    uint64_t t = 0;
    for (size_t r = 0; r < n; ++r) { cigoff[r] = t; t += (*ncig)[r]; }
    cigoff[n] = t;
    t = 0;
    for (size_t k = 0; k < C.x_ncig.size(); ++k) { x_cigoff[k] = t; t += C.x_ncig[k]; }
    cigar->assign(cigoff[n], 0);
    for (size_t r = 0; r < n; ++r) {                // k_expand_cigar
        const bool plain = (C.plain[r >> 5] >> (r & 31u)) & 1u;
        if (plain) { (*cigar)[cigoff[r]] = C.common_lq << 4; continue; }
        for (uint32_t j = 0; j < (*ncig)[r]; ++j) (*cigar)[cigoff[r] + j] = C.x_cigar[x_cigoff[xidx[r]] + j];
    }
}

}  // namespace np1up
