// Drop-in entry points of nextpolish1.so (include/nextpolish1.h, Part 1).
// One call = one contig, like the reference (reference: source/lib/scorechain.c:3-15,
// source/lib/nextpolish1.py:181-189): the contig and its BAM records are decoded on the host,
// staged into HBM, polished by the HIP pipeline and returned as a calloc'd PolishResult.
// Fatal conditions follow the reference's convention: message on stderr, exit(1).
#include <unistd.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nextpolish1.h"
#include "np_bam.h"
#include "np_stream.h"
#include "np1_priv.h"

int np1_batch_download_slots(np1_batch* b, int64_t c, std::vector<uint32_t>* soff, std::vector<uint16_t>* res);

namespace {

[[noreturn]] void die(const std::string& msg) {
    fprintf(stderr, "nextpolish1 (MI355X): %s\n", msg.c_str());
    exit(1);
}

// Lazily created per process, AFTER any fork of the caller's worker pool.
np1_ctx* g_ctx = nullptr;
pid_t g_ctx_pid = 0;

np1_ctx* process_ctx() {
    pid_t me = getpid();
    if (g_ctx && g_ctx_pid != me)
        die("HIP context was created before fork(); call the polish entry points only inside worker processes");
    if (!g_ctx) {
        int n = np1_device_count();
        if (n <= 0) die("no HIP device available; this library has no CPU fallback");
        int dev = (int)((unsigned long)me % (unsigned long)n);   // spread a worker pool over the node's GPUs
        if (const char* e = getenv("NP1_DEVICE")) dev = atoi(e);
        g_ctx = np1_ctx_create(dev);
        if (!g_ctx) die(std::string("cannot create device context: ") + np1_last_error());
        g_ctx_pid = me;
    }
    return g_ctx;
}

// -debug (trace_polish_open): the list of changed bases the reference builds inside contig_get_contig (source/lib/contig.c:743-797),
// which every task ends with (scorechain.c:12, kmercount.c:121, snpvalid.c:30, snpphase.c:129).  One walk over the contig's slots in
// their final state: a dropped base is ('.', draft), an insertion column that survived (base, '.'), a substituted base (base, draft).
// All four launch sequences leave that state in the batch's slot arrays (slot offsets + chosen base per slot).
void fill_points(np1_batch* b, const np::ReadStream& s, PolishResult* res) {
    std::vector<uint32_t> soff;
    std::vector<uint16_t> sres;
    if (np1_batch_download_slots(b, 0, &soff, &sres) != 0) die(np1_last_error());
    const std::string& draft = s.draft;
    std::vector<PolishPoint> pts;
    const int32_t L = s.ctg_len[0];
    static const char tbl[] = "=ACMGRSVTWYHKDBN";
    for (int32_t i = 0; i < L; ++i) {
        const uint32_t s0 = soff[i] - soff[0], s1 = soff[i + 1] - soff[0];
        const char was = (char)toupper((unsigned char)draft[i]);
        for (uint32_t t = s0; t < s1; ++t) {
            const int j = (int)(t - s0);
            const uint32_t base = sres[t] & 0xff;
            PolishPoint p;
            p.pos = i;
            p.index = (int16_t)j;
            if (base == 3) {
                if (j == 0) { p.curbase = '.'; p.base = was; pts.push_back(p); }
            } else {
                p.curbase = tbl[base & 0xf];
                if (j != 0) { p.base = '.'; pts.push_back(p); }
                else if (p.curbase != was) { p.base = was; pts.push_back(p); }
            }
        }
    }
    res->datalength = (int32_t)pts.size();
    res->data = (PolishPoint*)calloc(pts.size() ? pts.size() : 1, sizeof(PolishPoint));
    if (!pts.empty()) memcpy(res->data, pts.data(), pts.size() * sizeof(PolishPoint));
}

}  // namespace

// A C++ exception must not leave through the C boundary (the caller is ctypes: there is no frame that could catch it, the process would
// end in std::terminate without a word): it becomes the same kind of exit as every other failure of these entry points.
template <class F>
PolishResult* guarded(const char* what, F f) {
    try {
        return f();
    } catch (const std::exception& e) {
        die(std::string(what) + ": " + e.what());
    } catch (...) {
        die(std::string(what) + ": unknown exception");
    }
}

extern "C" {

Configure* config_init(const char* fastafn, const char* bamfn, const char* thirdbamfn) {
    Configure* r = (Configure*)calloc(sizeof(Configure), 1);
    r->trim_len_edge = 2;
    r->ext_len_edge = 2;
    r->min_map_quality = 0;
    r->indel_balance_factor_sgs = 0.5;
    r->min_count_ratio_skip = 0.8;
    r->min_len_ldr = 3;
    r->min_len_inter_kmer = 5;
    r->max_len_kmer = 50;
    r->max_count_kmer = 50;
    r->min_depth_snp = 3;
    r->min_count_snp = 5;
    r->min_count_snp_link = 5;
    r->ploidy = 2;
    r->indel_balance_factor_lgs = 0.33;
    r->max_indel_factor_lgs = 0.21;
    r->max_snp_factor_lgs = 0.53;
    r->min_snp_factor_sgs = 0.34;
    r->region_count = 10000;
    r->count_read_ins_sgs = 10000;
    r->max_ins_len_sgs = 10000;
    r->max_ins_fold_sgs = 5;
    r->max_variant_count_lgs = 150000;
    r->max_clip_ratio_sgs = 0.15;
    r->max_clip_ratio_lgs = 0.4;
    r->trace_polish_open = 0;
    r->fastafn = fastafn ? strdup(fastafn) : nullptr;
    r->bamfn = (bamfn && access(bamfn, 0) == 0) ? strdup(bamfn) : nullptr;
    if (r->bamfn) {
        uint32_t mean = 0;
        int32_t rl = 0;
        if (!np::bam_insert_probe(r->bamfn, r->count_read_ins_sgs, r->max_ins_len_sgs, &mean, &rl))
            die(std::string("cannot read BAM ") + r->bamfn);
        r->read_len = rl;
        r->read_tlen = (int32_t)(mean * (uint32_t)r->max_ins_fold_sgs);
    } else {
        r->read_tlen = 0;
    }
    r->thirdbamfn = (thirdbamfn && access(thirdbamfn, 0) == 0) ? strdup(thirdbamfn) : nullptr;
    return r;
}

void config_destory(Configure* c) {
    if (!c) return;
    free(c->fastafn);
    free(c->bamfn);
    free(c->thirdbamfn);
    free(c);
}

void polishresult_destory(PolishResult* p) {
    if (!p) return;
    free(p->contig);
    free(p->data);
    free(p);
}

static PolishResult* score_chain_task(const char* tigname, Configure* cfg) {
    if (!cfg || !cfg->fastafn) die("score_chain: configuration without a FASTA");
    if (!cfg->bamfn) die("score_chain: short-read BAM missing or unreadable");
    np1_stream st;
    std::string err;
    if (!np::load_stream(cfg->fastafn, cfg->bamfn, {std::string(tigname)}, false, &st.s, &err)) die(err);
    np1_ctx* ctx = process_ctx();
    np1_batch* b = np1_batch_upload(ctx, &st);
    if (!b) die(np1_last_error());
    if (np1_batch_score_chain(b, cfg, nullptr) != 0) die(np1_last_error());
    int64_t len = np1_batch_result_len(b, 0);
    PolishResult* res = (PolishResult*)calloc(sizeof(PolishResult), 1);
    res->contig = (char*)calloc(1, (size_t)len + 1);
    if (np1_batch_result_copy(b, 0, res->contig, len + 1) != 0) die(np1_last_error());
    res->length = (int32_t)len;
    if (cfg->trace_polish_open) fill_points(b, st.s, res);
    np1_batch_free(b);
    return res;
}

static PolishResult* kmer_task(const char* tigname, Configure* cfg, bool snp_valid_task) {
    const char* task = snp_valid_task ? "snp_valid" : "kmer_count";
    if (!cfg || !cfg->fastafn) die(std::string(task) + ": configuration without a FASTA");
    if (!cfg->bamfn) die(std::string(task) + ": short-read BAM missing or unreadable");
    np1_stream st;
    std::string err;
    if (!np::load_stream(cfg->fastafn, cfg->bamfn, {std::string(tigname)}, true, &st.s, &err)) die(err);
    np1_ctx* ctx = process_ctx();
    np1_batch* b = np1_batch_upload(ctx, &st);
    if (!b) die(np1_last_error());
    {   // kmer_count and snp_valid replay the reference's region iterator on the BAM index; NP1_ITER_REPLAY=0: records in file order
        const char* e = getenv("NP1_ITER_REPLAY");
        if (!(e && e[0] == '0') && np1_batch_enable_replay(b, &st, cfg->bamfn) != 0) die(np1_last_error());
    }
    if ((snp_valid_task ? np1_batch_snp_valid(b, cfg, nullptr) : np1_batch_kmer_count(b, cfg, nullptr)) != 0) die(np1_last_error());
    int64_t len = np1_batch_result_len(b, 0);
    PolishResult* res = (PolishResult*)calloc(sizeof(PolishResult), 1);
    res->contig = (char*)calloc(1, (size_t)len + 1);
    if (np1_batch_result_copy(b, 0, res->contig, len + 1) != 0) die(np1_last_error());
    res->length = (int32_t)len;
    if (cfg->trace_polish_open) fill_points(b, st.s, res);
    np1_batch_free(b);
    return res;
}
/* task 1 (reference: source/lib/scorechain.c:3-15) */
PolishResult* score_chain(const char* tigname, Configure* cfg) { return guarded("score_chain", [&] { return score_chain_task(tigname, cfg); }); }
PolishResult* kmer_count(const char* tigname, Configure* cfg) { return guarded("kmer_count", [&] { return kmer_task(tigname, cfg, false); }); }
/* task 4 (reference: source/lib/snpvalid.c:3-36) */
PolishResult* snp_valid(const char* tigname, Configure* cfg) { return guarded("snp_valid", [&] { return kmer_task(tigname, cfg, true); }); }
/* task 3 (reference: source/lib/snpphase.c:87-134): short reads from cfg->bamfn, long reads from cfg->thirdbamfn */
static PolishResult* snp_phase_task(const char* tigname, Configure* cfg) {
    if (!cfg || !cfg->fastafn) die("snp_phase: configuration without a FASTA");
    if (!cfg->bamfn) die("snp_phase: short-read BAM missing or unreadable");
    if (!cfg->thirdbamfn) die("snp_phase: long-read BAM missing or unreadable (the reference dereferences a null index here)");
    np1_stream ss, sl;
    std::string err;
    if (!np::load_stream(cfg->fastafn, cfg->bamfn, {std::string(tigname)}, true, &ss.s, &err)) die(err);
    if (!np::load_stream(cfg->fastafn, cfg->thirdbamfn, {std::string(tigname)}, true, &sl.s, &err)) die(err);
    np1_ctx* ctx = process_ctx();
    np1_batch* b = np1_batch_upload(ctx, &ss);
    if (!b) die(np1_last_error());
    np1_batch* l = np1_batch_upload(ctx, &sl);
    if (!l) die(np1_last_error());
    if (np1_batch_snp_phase(b, l, cfg) != 0) die(np1_last_error());
    int64_t len = np1_batch_result_len(b, 0);
    PolishResult* res = (PolishResult*)calloc(sizeof(PolishResult), 1);
    res->contig = (char*)calloc(1, (size_t)len + 1);
    if (np1_batch_result_copy(b, 0, res->contig, len + 1) != 0) die(np1_last_error());
    res->length = (int32_t)len;
    if (cfg->trace_polish_open) fill_points(b, ss.s, res);
    np1_batch_free(l);
    np1_batch_free(b);
    return res;
}
PolishResult* snp_phase(const char* tigname, Configure* cfg) { return guarded("snp_phase", [&] { return snp_phase_task(tigname, cfg); }); }
PolishResult* lgspolish(const char* tigname, Configure* cfg) {
    (void)tigname; (void)cfg;
    die("lgspolish (task 5) is disabled by the reference's own caller; use the long-read path");
}

}  // extern "C"
