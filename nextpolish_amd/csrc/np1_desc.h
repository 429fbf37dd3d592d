// Record descriptors of the fused score_chain pipeline (k_desc / k_tile3), host+device.
//
// The symbol a record votes at a slot is a pure function of the slot's (draft index g, insertion column
// jj) and of a handful of per-record intervals, so no symbol row is ever materialised: the pass-2 walk of
// the reference (source/lib/contig.c:247-331) is condensed per record into
//   segments  [g_lo, g_lo+len) -> query index q_lo + (g - g_lo), or DEL for a deletion op
//   inserts   draft index p whose insertion columns 0..len-1 carry query bases q0.., later columns DEL
// plus the contiguous slot run [sfirst, slast] the record votes on; every other covered insertion column
// votes DEL (the padding rules of contig.c:273-284,305-313).  A record with more indel operations than one
// descriptor holds continues in a chain of "parts" (overflow pool), each covering the next slot run.
#pragma once
#include "np1_core.h"

namespace np1k {

constexpr int DESC_NSEG = 7, DESC_NINS = 2;
constexpr int DESC_SEG0 = 4, DESC_INS0 = DESC_SEG0 + 2 * DESC_NSEG, DESC_NEXT = DESC_INS0 + 2 * DESC_NINS;
constexpr int DESC_WORDS = DESC_NEXT + 2;   // 24 words = 96 B per record (seven segments: a read with up to three deletions, or two
                                            // deletions and two insertions, stays in one descriptor; five segments sent 4 % of the records of a
                                            // 0.5 %-indel draft to chained parts in HBM, which cost the tile kernels a tenth of their time)
constexpr uint32_t DESC_CHAIN = 1u << 16;   // d[2] flag of a head part that continues in the overflow pool
constexpr uint32_t DESC_SIMPLE = 1u << 17;  // d[2] flag: the whole record is ONE matched segment (no deletion, no insertion, no further part)
// d[0]=sfirst d[1]=slast (this part)  d[2]=nseg | nins<<8 | flags  d[3]=low word of the record's offset in the base pool
// seg k: d[SEG0+2k]=g_lo, +1: len | qcode<<16 (qcode = q_lo, or 0xffff for DEL)
// ins k: d[INS0+2k]=p,    +1: len | q0<<16
// d[NEXT]=index+1 of the next part in the overflow pool (0 = none)   d[NEXT+1]=slast of the whole record (head part only)

struct DescSink {
    uint32_t* head;        // this record's slot in the per-record descriptor array
    uint32_t* ovf_pool;    // overflow parts, DESC_WORDS each
    uint32_t ovf_cap;      // parts available
    uint32_t* ovf_count;   // parts handed out so far (device: atomic)
    uint32_t* err;         // counters[CNT_ERR]
};

NP1_HD uint32_t desc_alloc_part(const DescSink& k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(k.ovf_count, 1u);
#else
    return (*k.ovf_count)++;
#endif
}

struct DescBuilder {
    DescSink sink;
    uint32_t* d;
    uint32_t nseg, nins, sfirst, slast;
    bool any, failed;
    NP1_HD void begin(const DescSink& s) {
        sink = s; d = s.head; nseg = nins = 0; sfirst = 1; slast = 0; any = false; failed = false;
        d[DESC_NEXT] = 0; d[DESC_NEXT + 1] = 0; d[3] = 0;
    }
    // closes the current part and continues in a fresh one from the overflow pool
    NP1_HD void split() {
        if (failed) return;
        uint32_t idx = desc_alloc_part(sink);
        if (idx >= sink.ovf_cap) { failed = true; np1_atomic_or(sink.err, ERR_DESC_OVERFLOW); return; }
        d[0] = sfirst; d[1] = slast; d[2] = nseg | nins << 8 | DESC_CHAIN;
        d[DESC_NEXT] = idx + 1;
        d = sink.ovf_pool + (uint64_t)idx * DESC_WORDS;
        d[DESC_NEXT] = 0; d[DESC_NEXT + 1] = 0; d[3] = 0;
        nseg = nins = 0;
        sfirst = slast + 1;   // votes are contiguous: the next part starts right after this one
    }
    NP1_HD void add_seg(uint32_t g_lo, uint32_t n, uint32_t qcode, uint32_t new_slast) {
        while (n) {   // a segment longer than 16 bits of length is cut (never for short reads)
            if (nseg == (uint32_t)DESC_NSEG) split();
            if (failed) return;
            uint32_t take = n < 0xffffu ? n : 0xfffeu;
            d[DESC_SEG0 + 2 * nseg] = g_lo;
            d[DESC_SEG0 + 2 * nseg + 1] = take | qcode << 16;
            ++nseg;
            g_lo += take;
            if (qcode != 0xffffu) qcode += take;
            n -= take;
        }
        slast = new_slast;
    }
    NP1_HD void add_ins(uint32_t p, uint32_t len, uint32_t q0, uint32_t new_slast) {
        if (nins == (uint32_t)DESC_NINS) split();
        if (failed) return;
        d[DESC_INS0 + 2 * nins] = p;
        d[DESC_INS0 + 2 * nins + 1] = len | q0 << 16;
        ++nins;
        slast = new_slast;
    }
    NP1_HD void finish() {
        if (!any) { sfirst = 1; slast = 0; }
        d[0] = sfirst; d[1] = slast; d[2] = nseg | nins << 8;
        if (d == sink.head && any && nseg == 1 && nins == 0 && (d[DESC_SEG0 + 1] >> 16) != 0xffffu) d[2] |= DESC_SIMPLE;
        sink.head[DESC_NEXT + 1] = any ? slast : 0;   // whole-record slast for the candidate test
    }
};

// mirrors walk_record (np1_core.h) at op granularity
template <class So>
NP1_HD void build_desc(const uint32_t* cg, uint32_t ncig, int32_t pos0, uint32_t g0, int32_t L, int32_t qs, int32_t qe,
                       int32_t lq, So so, const DescSink& sink) {
    DescBuilder b;
    b.begin(sink);
    if (lq >= 0xffff) { b.failed = true; np1_atomic_or(sink.err, ERR_DESC_OVERFLOW); }   // query indices are stored in 16 bits
    int32_t pos = pos0, qpos = 0;
    uint32_t last = 1;   // BAM_CINS
    for (uint32_t i = 0; i < ncig && !b.failed; ++i) {
        const uint32_t op = cig_op(cg[i]);
        const int32_t len = cig_len(cg[i]);
        if (op == 0 || op == 2) {
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
                const int32_t p = pos + jlo, q = op == 0 ? qpos + jlo : qpos;
                if (!b.any) {   // where the run of votes starts: the first column may still pad the columns before it
                    b.any = true;
                    const uint32_t lastj = jlo > 0 ? op : last;
                    const bool pad = lastj != 1 && p > 0 && (q > qs || (q == qs && lastj == 2));
                    b.sfirst = pad ? so(g0 + (uint32_t)p - 1) + 1 : so(g0 + (uint32_t)p);
                }
                b.add_seg(g0 + (uint32_t)p, (uint32_t)(jhi - jlo + 1), op == 0 ? (uint32_t)q : 0xffffu,
                          so(g0 + (uint32_t)(pos + jhi)));
            }
            if (len > 0) last = op;
            pos += len;
            if (op == 0) qpos += len;
        } else if (op == 1) {
            if (pos != 0) {
                if (pos > 0 && pos <= L - 1) {
                    int32_t jlo = 0, jhi = len - 1;
                    if (qs - qpos > jlo) jlo = qs - qpos;
                    if (qe - qpos < jhi) jhi = qe - qpos;
                    const int32_t qafter = qpos + len;
                    const bool pad = qafter > qs && qafter <= qe + 1;
                    if (jlo <= jhi || pad) {
                        const uint32_t sprev = so(g0 + (uint32_t)pos - 1);
                        if (!b.any) { b.any = true; b.sfirst = sprev + 1 + (uint32_t)jlo; }
                        const uint32_t nsl = pad ? so(g0 + (uint32_t)pos) - 1 : sprev + 1 + (uint32_t)jhi;
                        if (len >= 0xffff) { b.failed = true; np1_atomic_or(sink.err, ERR_DESC_OVERFLOW); }
                        else b.add_ins(g0 + (uint32_t)pos - 1, (uint32_t)len, (uint32_t)qpos, nsl);
                    }
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
    b.finish();
}

// symbol one part (descriptor d, packed bases sq) votes at a covered slot with draft index g and insertion
// column jj (jj < 0: the base slot itself)
template <class Sq>
NP1_HD uint32_t desc_symbol(const uint32_t* d, uint32_t g, int32_t jj, Sq sq) {
    const uint32_t cnt = d[2];
    if (jj < 0) {
        const uint32_t nseg = cnt & 0xff;
        for (uint32_t k = 0; k < nseg; ++k) {
            const uint32_t off = g - d[DESC_SEG0 + 2 * k], w = d[DESC_SEG0 + 2 * k + 1];
            if (off < (w & 0xffffu)) {
                const uint32_t qc = w >> 16;
                return qc == 0xffffu ? 3u : sq((int32_t)(qc + off));
            }
        }
        return 3u;   // unreachable for a covered base slot (votes are contiguous)
    }
    const uint32_t nins = (cnt >> 8) & 0xff;
    for (uint32_t k = 0; k < nins; ++k) {
        const uint32_t w = d[DESC_INS0 + 2 * k + 1];
        if (d[DESC_INS0 + 2 * k] == g && (uint32_t)jj < (w & 0xffffu)) return sq((int32_t)((w >> 16) + (uint32_t)jj));
    }
    return 3u;   // an insertion column this record only passes through (or pads): DEL
}

// k_desc body: descriptor (+ overflow parts) and the vote chunks the record's votes can touch
// (d = where the record's head descriptor is built: its place in the descriptor array, or a staging slot in LDS that the kernel
// writes out with coalesced stores afterwards)
NP1_HD void desc_record_at(uint32_t* d, const ReadsDev& R, int64_t r, const uint32_t* ctg_off, const uint32_t* soff, const int32_t* qs,
                           const int32_t* qe, uint32_t* ovf_pool, uint32_t ovf_cap, uint32_t* counters, uint32_t* c0_out, uint32_t* c1_out) {
    *c0_out = 1;
    *c1_out = 0;
    if (qs[r] <= qe[r]) {
        const uint32_t c = R.ctg[r];
        const uint32_t g0 = ctg_off[c];
        DescSink sink{d, ovf_pool, ovf_cap, &counters[CNT_OVFDESC], &counters[CNT_ERR]};
        build_desc(R.cigar + R.cigar_off[r], R.n_cigar[r], R.pos[r], g0, (int32_t)(ctg_off[c + 1] - g0), qs[r], qe[r],
                   R.l_qseq[r], SoGlobal{soff}, sink);
        d[3] = (uint32_t)R.seq_off[r];
        if (d[0] <= d[DESC_NEXT + 1] && d[0] <= d[1]) {
            *c0_out = d[0] / VOTE_CH;
            *c1_out = (d[DESC_NEXT + 1] + 2) / VOTE_CH;
        }
    } else {
        d[0] = 1; d[1] = 0; d[2] = 0; d[3] = 0; d[DESC_NEXT] = 0; d[DESC_NEXT + 1] = 0;
    }
}
NP1_HD void desc_record(const ReadsDev& R, int64_t r, const uint32_t* ctg_off, const uint32_t* soff, const int32_t* qs,
                        const int32_t* qe, uint32_t* desc, uint32_t* ovf_pool, uint32_t ovf_cap, uint32_t* counters,
                        uint32_t* c0_out, uint32_t* c1_out) {
    desc_record_at(desc + (uint64_t)r * DESC_WORDS, R, r, ctg_off, soff, qs, qe, ovf_pool, ovf_cap, counters, c0_out, c1_out);
}

}  // namespace np1k
