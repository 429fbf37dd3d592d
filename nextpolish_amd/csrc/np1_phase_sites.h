// snp_phase, second half: everything that works on the sites (P7), the low-depth regions (P9) and the links (P10).
// Included by np1_phase.h.
#pragma once

namespace np1p {

// ---- spanning-read haplotypes of one site ------------------------------------------------------------------------
// candidate table in the haplotype pool: header + `length` bytes each, first-seen order
struct SpCand { int32_t num, mapqual, qual, len; };

struct SpHapSink {   // KcHapSink with atomic mark updates (neighbouring sites run at the same time)
    const KcCtx* c;
    uint32_t g0;
    uint8_t* buf;
    int32_t length, cap, qual, del;
    const uint8_t* q;
    bool keep_marks;
    NP1_HD void vote(int32_t pos, uint32_t col, uint32_t sym, int32_t qpos, bool pad) {
        if (col >= c->soff[g0 + (uint32_t)pos + 1] - c->soff[g0 + (uint32_t)pos]) { np1_atomic_or(c->err, ERR_SP_UNDEFINED); return; }
        if (length < cap) buf[length] = (uint8_t)sym;
        ++length;
        if (qpos >= 0) qual += q[qpos];
        if (pad) ++del;
        if (!keep_marks) sp_atomic_and8(&c->sflag[c->soff[g0 + (uint32_t)pos] + col], 0xffu & ~F_ZERO);
    }
};

// how often the record agrees with the draft at the two anchor columns (kmercount.c:416-420): every M / D step is tested, also
// outside the window and on deletions (with the query index of the next base), until the operation that passes `end` is done
NP1_HD int32_t sp_anchor_hits(const KcCtx& c, int64_t r, uint32_t g0, int32_t left, int32_t right, int32_t end) {
    const uint32_t ncig = c.R.n_cigar[r];
    const uint32_t* cg = c.R.cigar + c.R.cigar_off[r];
    const uint8_t* seq = c.R.seq + c.R.seq_off[r];
    const int32_t lq = c.R.l_qseq[r];
    int32_t pos = c.R.pos[r], qpos = 0, result = 0;
    for (uint32_t i = 0; i < ncig; ++i) {
        const uint32_t op = cig_op(cg[i]);
        const int32_t len = cig_len(cg[i]);
        if (op == 0 || op == 2) {
            for (int t = 0; t < 2; ++t) {
                const int32_t a = t == 0 ? left : right;
                if (t == 1 && right == left) break;
                if (a >= pos && a < pos + len) {
                    const int32_t qq = op == 0 ? qpos + (a - pos) : qpos;
                    if (qq < lq && seq_nib(seq, qq) == c.sbase[c.soff[g0 + (uint32_t)a]]) ++result;
                }
            }
            pos += len;
            if (op == 0) qpos += len;
        } else if (op == 1 || op == 4 || op == 5) {
            qpos += len;
        }
        if (pos > end) break;
    }
    return result;
}

struct SpSites {   // per site, batch-wide, ordered by (contig, position)
    const uint32_t* g;        // global base index
    const uint32_t* ctg;
    const int32_t* left;      // anchors, local positions
    const int32_t* right;
    int32_t* len;             // Snps.length (int16 in the reference)
    uint8_t* keep;
    const uint32_t* roff;     // two allele strings of `rstride` bytes each at rpool + roff
    const uint32_t* rstride;
    uint8_t* rpool;
};

// ts_fliter_snps for one site (snpphase.c:229-334).  soff1 / cnt1 / first1: the slot space and histogram of P2 (the main slot's
// symbol list in first-seen order is the reference's base->data).
NP1_HD void sp_site_verdict(const KcCtx& sr, const KcCtx& lr, const SpParams& P, const SpSites& S, uint32_t k, const uint32_t* soff1,
                            const uint32_t* cnt1, const uint32_t* first1) {
    const KcCtx& c = sr;
    const uint32_t ct = S.ctg[k], g = S.g[k], g0 = c.ctg_off[ct];
    const int32_t i = (int32_t)(g - g0);
    const uint32_t sb = c.soff[g];
    const int32_t nins = (int32_t)(c.soff[g + 1] - sb - 1);
    int32_t length = 1, start = i, end = i, total = 0;
    bool flag = false, have_ks = false;
    if (nins <= 0 && (int32_t)c.scount[sb] > P.min_count_snp) { S.keep[k] = 1; return; }   // the common site: covered well, no columns -- kept as found
    // spanning records of both streams (swapped-interval query, contig.c:1130-1135): pos < start, endpos > end + 1
    const int32_t end_q = nins > 0 ? i + 1 : i;
    const int64_t srb = (int64_t)sr.read_begin[ct], sre = (int64_t)sr.read_begin[ct + 1];
    const int64_t lrb = (int64_t)lr.read_begin[ct], lre = (int64_t)lr.read_begin[ct + 1];
    const int64_t s_r0 = kc_lower_bound_pos(sr.R, srb, sre, start - sr.max_span), s_stop = kc_lower_bound_pos(sr.R, srb, sre, start);
    const int64_t l_r0 = kc_lower_bound_pos(lr.R, lrb, lre, start - lr.max_span), l_stop = kc_lower_bound_pos(lr.R, lrb, lre, start);
    uint32_t n_span = 0;
    if (nins > 0) for (int64_t r = s_r0; r < s_stop; ++r) n_span += sr.endpos[r] > end_q + 1 && sr.level[r] == 2;
    for (int64_t r = l_r0; r < l_stop; ++r) n_span += lr.endpos[r] > end_q + 1 && lr.level[r] == 1;
    if (nins > 0) { end = i + 1; length += nins + 1; }
    const uint32_t max_cand = n_span + 17;
    const uint32_t stride = ((uint32_t)length + (uint32_t)sizeof(SpCand) + 3u) & ~3u;
    const uint32_t bytes = (max_cand + 2) * stride;
    const uint32_t off = kc_bump(c.hcount, (bytes + 15u) & ~15u);
    if ((uint64_t)off + bytes > c.hcap) { np1_atomic_or(c.err, ERR_SP_POOL); return; }
    uint8_t* pool = c.hpool + off;
    uint8_t* work = pool + (uint64_t)max_cand * stride + sizeof(SpCand);        // the KmerScore being filled
    uint8_t* probe = pool + (uint64_t)(max_cand + 1) * stride + sizeof(SpCand);  // the comparison strings built with memset
    uint32_t ncand = 0;
    auto cand = [&](uint32_t t) -> SpCand* { return reinterpret_cast<SpCand*>(pool + (uint64_t)t * stride); };
    auto cbytes = [&](uint32_t t) -> uint8_t* { return pool + (uint64_t)t * stride + sizeof(SpCand); };
    auto find = [&](const uint8_t* item) -> int32_t {   // seqlist_get_index + ks_compare_region: the list element's length decides
        for (uint32_t t = 0; t < ncand; ++t) {
            const uint8_t* h = cbytes(t);
            const int32_t n = cand(t)->len;
            bool same = true;
            for (int32_t u = 0; u < n; ++u) if (h[u] != item[u]) { same = false; break; }
            if (same) return (int32_t)t;
        }
        return -1;
    };
    auto get_region = [&](const KcCtx& cc, int64_t r, bool anchors, bool keep_marks) {   // ss_kmer_get_region (kmercount.c:332-363), flag -1
        SpHapSink sink{&cc, g0, work, 0, length, 0, 0, cc.qual + cc.qual_off[r], keep_marks};
        int32_t mq = 0, hits = 0;
        if (cc.R.n_cigar[r]) {
            mq = cc.mapq[r];
            kc_walk(cc, r, g0, start, end, sink);
            if (sink.length > 0 && sink.length != sink.del) sink.qual /= sink.length - sink.del; else sink.qual = 0;
            if (anchors) hits = sp_anchor_hits(cc, r, g0, S.left[k], S.right[k], end);
        }
        if (sink.length == length && hits >= (anchors ? 2 : 0)) {
            const int32_t hit = find(work);
            if (hit < 0) {
                if (ncand < max_cand) {
                    SpCand* cd = cand(ncand);
                    cd->num = 1; cd->mapqual = mq; cd->qual = sink.qual; cd->len = length - 1;
                    uint8_t* dst = cbytes(ncand);
                    for (int32_t t = 0; t < length; ++t) dst[t] = work[t];
                    ++ncand;
                }
            } else {
                SpCand* cd = cand((uint32_t)hit);
                cd->num++; cd->mapqual += mq; cd->qual += sink.qual;
            }
            ++total;
        }
    };
    if (nins > 0) {
        for (int64_t r = s_r0; r < s_stop; ++r) {
            if (!(sr.endpos[r] > end + 1) || sr.level[r] != 2) continue;
            have_ks = true;
            get_region(sr, r, false, false);
        }
        flag = true;
    } else {
        have_ks = true;
        total = c.scount[sb];
    }
    if (total <= P.min_count_snp) {
        if (!have_ks) { np1_atomic_or(c.err, ERR_SP_UNDEFINED); return; }   // snpphase.c:269 writes through a null KmerScore here
        if (length == 1) {   // the base's own symbols, in first-seen order, as zero-length candidates (snpphase.c:259-268)
            const uint32_t* cn = cnt1 + (uint64_t)soff1[g] * 16;
            const uint32_t* fs = first1 + (uint64_t)soff1[g] * 16;
            uint32_t last = 0;
            bool any = false;
            for (;;) {
                uint32_t best = 16, bf = 0xffffffffu;
                for (uint32_t s = 0; s < 16; ++s)
                    if (cn[s] && (!any || fs[s] > last) && fs[s] < bf) { bf = fs[s]; best = s; }
                if (best == 16) break;
                any = true; last = bf;
                SpCand* cd = cand(ncand);
                cd->num = (int32_t)(cn[best] & 0xffffu); cd->mapqual = 60 * cd->num; cd->qual = 41 * cd->num; cd->len = 0;
                cbytes(ncand)[0] = (uint8_t)best;
                ++ncand;
            }
        }
        for (int32_t t = 0; t < length; ++t) probe[t] = (uint8_t)SYM_DEL;
        int32_t del_idx = find(probe);
        for (int64_t r = l_r0; r < l_stop; ++r) {
            if (!(lr.endpos[r] > end + 1) || lr.level[r] != 1) continue;
            get_region(lr, r, true, true);
        }
        flag = true;
        if (del_idx == -1) {   // an all-deletion haplotype only the long reads brought is taken out again (snpphase.c:283-291)
            del_idx = find(probe);
            if (del_idx != -1) {
                for (uint32_t t = (uint32_t)del_idx; t + 1 < ncand; ++t) {
                    *cand(t) = *cand(t + 1);
                    uint8_t *d = cbytes(t), *s = cbytes(t + 1);
                    for (int32_t u = 0; u < length; ++u) d[u] = s[u];
                }
                --ncand;
            }
        }
    }
    if (!flag) { S.keep[k] = 1; return; }
    if (ncand == 0) { np1_atomic_or(c.err, ERR_SP_UNDEFINED); return; }   // ts_get_nlargest of an empty list reads an unset pointer
    // stable top two by (num, mapqual, qual) (snpphase.c:873-903)
    auto better = [&](uint32_t a, uint32_t b) -> bool {   // ks_compare(a, b) > 0
        const SpCand *x = cand(a), *y = cand(b);
        if (x->num != y->num) return x->num > y->num;
        if (x->mapqual != y->mapqual) return x->mapqual > y->mapqual;
        return x->qual > y->qual;
    };
    uint32_t m0 = 0, m1 = 0, nm = 1;
    for (uint32_t t = 1; t < ncand; ++t) {
        if (nm == 1) {
            if (better(t, m0)) { m1 = m0; m0 = t; } else m1 = t;
            nm = 2;
        } else if (better(t, m1)) {
            if (better(t, m0)) { m1 = m0; m0 = t; } else m1 = t;
        }
    }
    const double rate = nm == 1 ? 0 : cand(m1)->num / (double)cand(m0)->num;
    for (int32_t t = 0; t < length; ++t) probe[t] = (uint8_t)SYM_DEL;
    probe[0] = c.sbase[sb];
    bool same = true;
    for (int32_t t = 0; t < cand(m0)->len; ++t) if (cbytes(m0)[t] != probe[t]) { same = false; break; }
    const int32_t v = sp_check(P, total, rate, same);
    if (v == 1) {
        const int32_t ql = cand(m0)->len;
        S.len[k] = ql;
        uint8_t* r0 = S.rpool + S.roff[k];
        uint8_t* r1 = r0 + S.rstride[k];
        for (int32_t t = 0; t < ql; ++t) r0[t] = cbytes(m0)[t];
        if (nm == 2) for (int32_t t = 0; t < ql; ++t) r1[t] = cbytes(m1)[t];
        else for (int32_t t = 0; t < ql; ++t) r1[t] = probe[t];
        S.keep[k] = 1;
    } else {
        if (v == 2) {
            c.sbase[sb] = cbytes(m0)[0];
            if (end > start) for (int32_t t = 0; t <= nins; ++t) c.sbase[sb + (uint32_t)t] = cbytes(m0)[t];   // contig_update_contig(start, end, region, -1)
        }
        sp_atomic_and8(&c.sflag[sb], 0xf7u);
        S.keep[k] = 0;
    }
}

// ---- P9: ts_correct_lower_depth (snpphase.c:797-871): both streams vote, long-read rate, FLAG_THIRD rule ---------------------------
// The reference runs three loops over ALL regions: clean + draft vote + short reads, then the long reads, then the chains.  Merged
// regions can touch (one ends on the base the next starts on), and then the later region's clean wipes the earlier one's votes on
// that base before anything is scored.  Regions that touch form a group; one lane does the three loops over its group, in order.
NP1_HD void sp_lowdepth_group(const KcCtx& sr, const KcCtx& lr, const uint32_t* reg_ctg, const int32_t* reg_se, uint32_t first, uint32_t last) {
    for (uint32_t k = first; k < last; ++k) {
        const uint32_t ct = reg_ctg[k], g0 = sr.ctg_off[ct];
        const int32_t start = reg_se[2 * k], end = reg_se[2 * k + 1];
        const uint32_t s0 = sr.soff[g0 + (uint32_t)start], s1 = sr.soff[g0 + (uint32_t)end];
        for (uint32_t s = s0; s <= s1; ++s) { sr.lhead[s] = 0; sr.scount[s] = 0; }   // contig_clean_region
        kc_as_read(sr, g0, start, end);
        kc_parse_region(sr, ct, start, end, 2);
    }
    for (uint32_t k = first; k < last; ++k) kc_parse_region(lr, reg_ctg[k], reg_se[2 * k], reg_se[2 * k + 1], 1);
    for (uint32_t k = first; k < last; ++k)   // lr carries rate = indel_balance_factor_lgs and the third-generation rule
        if (!kc_region_solve(lr, lr.ctg_off[reg_ctg[k]], reg_se[2 * k], reg_se[2 * k + 1], -1, 0)) return;
}

// ---- P10: links --------------------------------------------------------------------------------------------------------
struct SpEntry { int32_t num, length, qual, off; };   // one KmerScore of snpslinkdata: site position, byte count, quality, first byte

struct SpLinks {
    const uint32_t* site_first;   // per contig: first site, [nc + 1]
    const int32_t* pos;           // per site: local position
    const int32_t* len;
    const uint32_t* roff;
    const uint32_t* rstride;
    const uint8_t* rpool;
    int32_t* num;                 // [site * 4 + (a - 1) * 2 + (b - 1)]
    int32_t* mapqual;
    int32_t* qual;
    unsigned long long* first;    // processing order of the first record that made the combination
    int32_t* total;               // per site
};

struct SpLinkSink {   // ts_snps_parse_read (snpphase.c:615-776) on the vote stream of the walk
    const KcCtx* c;
    uint32_t g0;
    const uint8_t* qual;
    uint32_t flagbrim;
    SpEntry* ents;
    uint32_t ecap, n;
    uint8_t* bytes;
    int32_t bcap;
    // the KmerScore being filled
    int32_t k_off, k_len, k_num, k_qual;
    int32_t q_off;            // where the next string starts
    int32_t del, curpos, sign, comfirm, budget;
    bool overflow;
    NP1_HD void put(uint32_t sym) {
        if (k_off + k_len < bcap) bytes[k_off + k_len] = (uint8_t)sym; else overflow = true;
        ++k_len;
    }
    NP1_HD void push() {
        if (n < ecap) ents[n] = SpEntry{k_num, k_len, k_qual, k_off}; else overflow = true;
        ++n;
    }
    NP1_HD bool stop_before(int32_t len) const { return budget - len < 0; }
    NP1_HD void vote(int32_t pos, uint32_t col, uint32_t sym, int32_t qpos, bool pad) {
        if (col != 0) {   // an insertion column: only while a string is open
            if (!curpos) return;
            put(sym);
            --budget;
            if (pad) ++del; else k_qual += qual[qpos];
            return;
        }
        const uint32_t bm = c->bmark[g0 + (uint32_t)pos], fl = bm & 0xffu;
        if (!(flagbrim == 0 || (fl & (F_LEFT | F_RIGHT)))) return;
        if (fl & F_SNP) {
            if (curpos == 0) {
                k_off = q_off; k_len = 0; k_num = pos; k_qual = 0; del = 0; curpos = 1;
                if (flagbrim == 0) sign = 1;
            } else {
                ++sign;
            }
        } else if (flagbrim) {
            if (c->sbase[c->soff[g0 + (uint32_t)pos]] == (cur_q < lq ? seq_nib(seq, cur_q) : 0xffu)) ++sign;   // (on a deletion: the next query base, as the reference reads it)
        } else {
            ++sign;
        }
        const bool has_cols = (bm & 0x100u) != 0;
        if (curpos) {
            put(sym);
            if (qpos >= 0) k_qual += qual[qpos];
            --budget;
            if (k_num != pos || !has_cols) {
                if (k_num != pos) {
                    if (k_len != del) k_qual = (int32_t)((double)k_qual / (double)(k_len - del)); else k_qual = 0;
                    --k_len;
                }
                push();
                q_off += k_len;
                curpos = 0;
            }
        }
        if (k_num != pos) {
            if (fl & F_SNP) {
                k_off = q_off; k_len = 1; k_num = pos; k_qual = cur_q < lq ? qual[cur_q] : 0; del = 0; curpos = 1;   // (quality on a deletion: the next query base's, as the reference reads it)
                if (flagbrim == 0) { ++comfirm; sign = 1; }
            } else if (fl & F_RIGHT) {
                if (sign == 2) comfirm = (int32_t)n;
                else if (comfirm >= 0) for (; comfirm < (int32_t)n; ++comfirm) if ((uint32_t)comfirm < ecap) ents[comfirm].length = 0;
                curpos = 0;
                sign = 0;
                if (fl & F_LEFT) ++sign;
            }
        }
    }
    int32_t cur_q, lq;    // query index the reference has at this step (on a deletion: of the next base), set by the walk; query length
    const uint8_t* seq;
};

// the walk of ts_snps_parse_read: kc_walk's order of votes, but (1) a leading insertion at position 0 does not move the query
// window while no string is open (snpphase.c:756-762), (2) the byte budget max_variant_count_lgs is tested before every
// operation (snpphase.c:634-637), (3) a deletion's "quality" is the next query base's
NP1_HD void sp_walk_links(const KcCtx& c, int64_t r, uint32_t g0, int32_t start, int32_t end, SpLinkSink& sink) {
    const uint32_t ncig = c.R.n_cigar[r];
    if (!ncig) return;
    const uint32_t* cg = c.R.cigar + c.R.cigar_off[r];
    const uint8_t* seq = c.R.seq + c.R.seq_off[r];
    int32_t qs, qe;
    kc_cut_read(c.R, r, c.trim, &qs, &qe);
    int32_t pos = c.R.pos[r], qpos = 0;
    uint32_t last = 1;
    for (uint32_t i = 0; i < ncig; ++i) {
        const uint32_t op = cig_op(cg[i]);
        const int32_t len = cig_len(cg[i]);
        if (sink.stop_before(len)) break;
        if (op == 0 || op == 2) {
            for (int32_t j = 0; j < len; ++j, ++pos) {
                if (!sink.curpos) {   // nothing happens on an unmarked base while no string is open: straight to the next marked one
                    const int32_t d = sp_next_marked(c.bbits, (uint64_t)g0 + (uint64_t)pos, len - j);
                    if (d > 0) {
                        pos += d; j += d;
                        if (op != 2) qpos += d;
                        last = op;
                        if (j >= len) break;
                    }
                }
                if (pos >= start && pos <= end && qpos >= qs && qpos <= qe) {
                    if (sink.curpos && last != 1 && pos > start && (qpos > qs || (qpos == qs && last == 2))) {   // (the columns matter only to an open string)
                        const uint32_t n = c.soff[g0 + (uint32_t)pos] - c.soff[g0 + (uint32_t)pos - 1] - 1;
                        for (uint32_t k = 0; k < n; ++k) sink.vote(pos - 1, k + 1, SYM_DEL, -1, true);
                    }
                    sink.cur_q = qpos;
                    if (op == 2) sink.vote(pos, 0, SYM_DEL, -1, false);
                    else sink.vote(pos, 0, seq_nib(seq, qpos), qpos, false);
                }
                if (op != 2) ++qpos;
                last = op;
            }
        } else if (op == 1) {
            if (sink.curpos) {
                if (pos != 0) {
                    const bool inr = pos > start && pos <= end;
                    for (int32_t j = 0; j < len; ++j, ++qpos)
                        if (inr && qpos >= qs && qpos <= qe) sink.vote(pos - 1, (uint32_t)j + 1, seq_nib(seq, qpos), qpos, false);
                    if (inr && qpos > qs && qpos <= qe + 1) {
                        const uint32_t n = c.soff[g0 + (uint32_t)pos] - c.soff[g0 + (uint32_t)pos - 1] - 1;
                        if (n == 0) np1_atomic_or(c.err, ERR_SP_UNDEFINED);   // snpphase.c:750 reads the length of a null list
                        for (uint32_t j = (uint32_t)len; j < n; ++j) sink.vote(pos - 1, j + 1, SYM_DEL, -1, true);
                    }
                } else {
                    qpos += len;
                    qs += len;
                }
            } else {
                qpos += len;
            }
            last = 1;
        } else if (op == 4 || op == 5) {
            qpos += len;
        }
    }
}

NP1_HD int32_t sp_site_find(const int32_t* pos, int32_t lo, int32_t hi, int32_t p) {   // snpslist_find (snpphase.c:68-85) on [lo, hi)
    int32_t i = lo, j = hi - 1;
    while (i <= j) {
        const int32_t mid = (i + j) / 2;
        if (pos[mid] == p) return mid;
        if (pos[mid] < p) i = mid + 1; else j = mid - 1;
    }
    return -1;
}
NP1_HD int32_t sp_allele_index(const SpLinks& L, int32_t site, const uint8_t* str) {   // snps_get_index (snpphase.c:40-48)
    const int32_t n = L.len[site];
    for (int32_t a = 0; a < 2; ++a) {
        const uint8_t* reg = L.rpool + L.roff[site] + (uint32_t)a * L.rstride[site];
        bool same = true;
        for (int32_t t = 0; t < n; ++t) if (reg[t] != str[t]) { same = false; break; }
        if (same) return a;
    }
    return -1;
}

// one record against one link region: parse, then ts_snps_deal_linkdata + ts_tranfer_link (snpphase.c:423-448,778-795)
NP1_HD void sp_link_record(const KcCtx& c, const SpParams& P, const SpLinks& L, int64_t r, uint32_t ct, int32_t start, int32_t end, uint32_t flagbrim,
                           unsigned long long order, SpEntry* ents, uint32_t ecap, uint8_t* bytes, int32_t bcap) {
    const uint32_t g0 = c.ctg_off[ct];
    SpLinkSink sink;
    sink.c = &c; sink.g0 = g0; sink.qual = c.qual + c.qual_off[r]; sink.flagbrim = flagbrim;
    sink.ents = ents; sink.ecap = ecap; sink.n = 0; sink.bytes = bytes; sink.bcap = bcap;
    sink.k_off = 0; sink.k_len = 0; sink.k_num = 0; sink.k_qual = 0; sink.q_off = 0;
    sink.del = 0; sink.curpos = 0; sink.sign = 0; sink.comfirm = 0; sink.budget = P.max_variant_count_lgs; sink.overflow = false;
    sink.cur_q = 0; sink.lq = c.R.l_qseq[r]; sink.seq = c.R.seq + c.R.seq_off[r];
    sp_walk_links(c, r, g0, start, end, sink);
    if (sink.overflow) { np1_atomic_or(c.err, ERR_SP_POOL); return; }
    const int32_t mq = c.mapq[r];
    const int32_t slo = (int32_t)L.site_first[ct], shi = (int32_t)L.site_first[ct + 1];
    for (uint32_t i = 1; i < sink.n; ++i) {
        SpEntry* p = &ents[i];
        SpEntry* q = &ents[i - 1];
        if (!p->length || !q->length) continue;
        if (flagbrim) {
            if (!(c.sflag[c.soff[g0 + (uint32_t)p->num]] & F_RIGHT) || !(c.sflag[c.soff[g0 + (uint32_t)q->num]] & F_LEFT)) continue;
        }
        const int32_t index = sp_site_find(L.pos, slo, shi, p->num);
        if (index < slo + 1) { np1_atomic_or(c.err, ERR_SP_UNDEFINED); return; }   // the reference indexes in front of its site array
        if (q->length != L.len[index - 1] || p->length != L.len[index]) continue;
        const int32_t a = sp_allele_index(L, index - 1, bytes + q->off);
        if (a < 0) continue;
        p->length = (a + 1) << 4;
        const int32_t b = sp_allele_index(L, index, bytes + p->off);
        if (b < 0) continue;
        p->length += b + 1;
        const uint32_t cell = (uint32_t)index * 4 + (uint32_t)a * 2 + (uint32_t)b;
        sp_atomic_add(&L.num[cell], 1);
        sp_atomic_add(&L.mapqual[cell], mq);
        sp_atomic_add(&L.qual[cell], p->qual);
        sp_atomic_min64(&L.first[cell], order);
        sp_atomic_add(&L.total[index], 1);
    }
}

}  // namespace np1p
