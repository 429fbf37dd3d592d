// C ABI for the host-side stream objects (include/nextpolish1.h, Part 2: np1_stream_*).
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np_stream.h"
#include "np1_priv.h"
#include "np_synth.h"
#include "np_inflate.h"
#include "np_inflate_lane.h"
#include "np_inflate_lds.h"
#include "np_crc32.h"


static thread_local std::string g_err;
extern "C" const char* np1_last_error(void) { return g_err.c_str(); }
void np1_set_error(const std::string& e) { g_err = e; }

// (np1_debug_inflate_lds* below)
template <int LB, int DB> static int debug_inflate_lds(const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_len) {
    std::vector<uint16_t> slots(nplds::Layout<LB, DB>::SLOTS);
    nplds::ArrayTab tab{slots.data()};
    nplds::Scratch sc;
    return nplds::inflate_block<LB, DB>(src, (uint32_t)src_len, dst, (uint32_t)dst_len, tab, &sc) == 0 ? 1 : 0;
}

extern "C" {

np1_stream* np1_stream_load(const char* fasta, const char* bam, const char* const* names, int n_names, int with_qual) {
    std::vector<std::string> nm;
    for (int i = 0; i < n_names; ++i) nm.push_back(names[i]);
    np1_stream* st = new np1_stream();
    std::string err;
    if (!np::load_stream(fasta, bam, nm, with_qual != 0, &st->s, &err)) {
        g_err = err;
        delete st;
        return nullptr;
    }
    return st;
}

void np1_stream_get_view(const np1_stream* st, np1_stream_view* v) {
    const np::ReadStream& s = st->s;
    memset(v, 0, sizeof(*v));
    v->n_contigs = (int64_t)s.n_contigs();
    v->n_reads = (int64_t)s.n_reads();
    v->ctg_len = s.ctg_len.data();
    v->ctg_off = s.ctg_off.data();
    v->read_begin = s.read_begin.data();
    v->draft = s.draft.data();
    v->draft_len = (int64_t)s.draft.size();
    v->pos = s.pos.data();
    v->ctg = s.ctg.data();
    v->flag = s.flag.data();
    v->n_cigar = s.n_cigar.data();
    v->l_qseq = s.l_qseq.data();
    v->cigar_off = s.cigar_off.data();
    v->seq_off = s.seq_off.data();
    v->mapq = s.mapq.data();
    v->isize = s.isize.data();
    v->qual_off = s.qual_off.data();
    v->cigar = s.cigar.data();
    v->cigar_len = (int64_t)s.cigar.size();
    v->seq = s.seq.data();
    v->seq_len = (int64_t)s.seq.size();
    v->qual = s.qual.data();
    v->qual_len = (int64_t)s.qual.size();
}

/* virtual offsets of the records of a stream loaded from a BAM file (start of each record, byte behind it); 0 entries for
 * streams that were built in memory */
int64_t np1_stream_voffs(const np1_stream* st, const uint64_t** beg, const uint64_t** end) {
    if (beg) *beg = st->s.voff.data();
    if (end) *end = st->s.voff_end.data();
    return st->s.voff.size() == st->s.n_reads() && st->s.voff_end.size() == st->s.n_reads() && st->s.voff.size() && st->s.voff.back() ? (int64_t)st->s.voff.size() : 0;
}

const char* np1_stream_contig_name(const np1_stream* st, int64_t i) {
    if (i < 0 || i >= (int64_t)st->s.n_contigs()) return nullptr;
    return st->s.names[(size_t)i].c_str();
}

uint64_t np1_stream_algorithmic_bytes(const np1_stream* st, int with_qual) {
    return st->s.algorithmic_input_bytes(with_qual != 0);
}

int np1_stream_write_files(const np1_stream* st, const char* fasta, const char* bam, int level) {
    std::string err;
    if (!np::write_stream_files(st->s, fasta, bam, level, &err)) { g_err = err; return -1; }
    return 0;
}

int np1_stream_write_files_aux(const np1_stream* st, const char* fasta, const char* bam, int level, const uint8_t* aux_pool,
                                const uint64_t* aux_off) {
    std::string err;
    if (!np::write_stream_files(st->s, fasta, bam, level, &err, aux_pool, aux_off)) { g_err = err; return -1; }
    return 0;
}

int np1_streams_write_files_q(np1_stream* const* sts, int n, const char* fasta, const char* bam, int level, int qual_model) {
    std::vector<const np::ReadStream*> ss;
    for (int i = 0; i < n; ++i) ss.push_back(&sts[i]->s);
    std::string err;
    if (!np::write_streams_files(ss, fasta, bam, level, &err, nullptr, nullptr, qual_model)) { g_err = err; return -1; }
    return 0;
}
int np1_streams_write_files(np1_stream* const* sts, int n, const char* fasta, const char* bam, int level) {
    return np1_streams_write_files_q(sts, n, fasta, bam, level, 0);
}

void np1_stream_free(np1_stream* st) {
    if (!st) return;
    np1_stream_unpin(st);
    delete st;
}

np1_stream* np1_stream_build(const np1_stream_view* v, const char* const* names) {
    np1_stream* st = new np1_stream();
    np::ReadStream& s = st->s;
    size_t nc = (size_t)v->n_contigs, nr = (size_t)v->n_reads;
    for (size_t c = 0; c < nc; ++c) s.names.push_back(names[c]);
    s.ctg_len.assign(v->ctg_len, v->ctg_len + nc);
    s.ctg_off.assign(v->ctg_off, v->ctg_off + nc + 1);
    s.read_begin.assign(v->read_begin, v->read_begin + nc + 1);
    s.draft.assign(v->draft, (size_t)v->draft_len);
    s.pos.assign(v->pos, v->pos + nr);
    s.ctg.assign(v->ctg, v->ctg + nr);
    s.flag.assign(v->flag, v->flag + nr);
    s.n_cigar.assign(v->n_cigar, v->n_cigar + nr);
    s.l_qseq.assign(v->l_qseq, v->l_qseq + nr);
    s.cigar_off.assign(v->cigar_off, v->cigar_off + nr);
    s.seq_off.assign(v->seq_off, v->seq_off + nr);
    s.mapq.assign(v->mapq, v->mapq + nr);
    s.isize.assign(v->isize, v->isize + nr);
    s.qual_off.assign(v->qual_off, v->qual_off + nr);
    s.cigar.assign(v->cigar, v->cigar + v->cigar_len);
    s.seq.assign(v->seq, v->seq + v->seq_len);
    s.qual.assign(v->qual, v->qual + v->qual_len);
    return st;
}

void np1_synth_defaults(np1_synth_params* p) { np::synth_default_params(p); }

np1_stream* np1_stream_synth(const np1_synth_params* p, const char* prefix) {
    np1_stream* st = new np1_stream();
    if (!np::synth_stream(*p, prefix ? prefix : "ctg", &st->s)) {
        g_err = "synthetic generation failed";
        delete st;
        return nullptr;
    }
    return st;
}

np1_stream* np1_stream_synth_long(const np1_synth_long_params* p, const char* prefix) {
    np1_stream* st = new np1_stream();
    if (!np::synth_long_stream(*p, prefix ? prefix : "ctg", &st->s)) {
        g_err = "synthetic generation failed";
        delete st;
        return nullptr;
    }
    return st;
}

/* test / bench helpers: the synthetic short-read (long_reads = 0: p = np1_synth_params) or long-read (1: np1_synth_long_params) workload over the
   contigs of a FASTA that is handed in -- the "re-mapped" reads of the next step of a multi-step run (np_synth.h) */
np1_stream* np1_stream_synth_on(const void* p, int long_reads, const char* const* names, const char* const* seqs, const int64_t* lens, int n) {
    std::vector<std::string> nm, sq;
    for (int i = 0; i < n; ++i) { nm.emplace_back(names[i]); sq.emplace_back(seqs[i], (size_t)lens[i]); }
    np1_stream* st = new np1_stream();
    const bool ok = long_reads ? np::synth_long_stream_on(*static_cast<const np1_synth_long_params*>(p), nm, sq, &st->s)
                               : np::synth_stream_on(*static_cast<const np1_synth_params*>(p), nm, sq, &st->s);
    if (!ok) { g_err = "synthetic generation failed"; delete st; return nullptr; }
    return st;
}

int np1_stream_synth_diploid(const np1_diploid_params* p, const char* prefix, np1_stream** sr, np1_stream** lr) {
    np1_stream *a = new np1_stream(), *b = new np1_stream();
    if (!np::synth_diploid_streams(*p, prefix ? prefix : "ctg", &a->s, &b->s)) {
        g_err = "synthetic generation failed (contigs shorter than 400 bases?)";
        delete a;
        delete b;
        return -1;
    }
    *sr = a;
    *lr = b;
    return 0;
}

/* test hook: the BGZF block decoder on a raw DEFLATE stream (1 = accepted and dst filled) */
int np1_debug_inflate(const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_len) {
    return np::inflate_raw(src, (size_t)src_len, dst, (size_t)dst_len) ? 1 : 0;
}
/* test hook: the lane-per-block decoder of the device-side ingest (np_inflate_lane.h), run on the host: 1 = accepted and dst filled */
int np1_debug_inflate_lane(const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_len) {
    std::vector<uint32_t> tab(nplane::LANE_TABLE_WORDS);
    return nplane::inflate_block(src, (uint32_t)src_len, dst, (uint32_t)dst_len, tab.data()) == 0 ? 1 : 0;
}
/* test hooks: the LDS-table decoder of the device-side ingest (np_inflate_lds.h), run on the host over a plain array, in the table sizes the
 * device kernels are built in and in a tiny one (6 / 4 bits) that sends most codes down the long-code path: 1 = accepted and dst filled */
int np1_debug_inflate_lds(const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_len) { return debug_inflate_lds<10, 8>(src, src_len, dst, dst_len); }
int np1_debug_inflate_lds85(const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_len) { return debug_inflate_lds<8, 5>(src, src_len, dst, dst_len); }
int np1_debug_inflate_lds96(const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_len) { return debug_inflate_lds<9, 6>(src, src_len, dst, dst_len); }
int np1_debug_inflate_lds75(const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_len) { return debug_inflate_lds<7, 5>(src, src_len, dst, dst_len); }      /* the kernel's default sizes */
int np1_debug_inflate_lds64(const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_len) { return debug_inflate_lds<6, 4>(src, src_len, dst, dst_len); }
/* test hook: the CRC-32 the BGZF reader / writer compute per block (np_crc32.h: carry-less-multiply folding) */
uint32_t np1_debug_crc32(const uint8_t* src, uint64_t len) { return np::crc32_block(src, (size_t)len); }

}  // extern "C"
