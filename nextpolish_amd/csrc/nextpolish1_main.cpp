// CLI of the short-read core: same surface as the reference binary
// (reference: source/lib/main.c:12-75, contig_total source/lib/contig.c:1056-1110):
//   nextpolish1 <scorechain|kmercount|snpphase|snpvalid|lgspolish> fasta bam [bam3]
// prints ">name_<step>\nseq" per contig in FASTA-index order.  Unlike the reference, which loops
// score_chain contig by contig, the contigs travel to the GPU in batches of NP1_BATCH_BP draft bases (default 16 M) on
// NP1_LANES device lanes (default 3) while host threads inflate and split the records of the next batches (np1_pipe.cpp).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/nextpolish1.h"

static const char* kCmds[] = {"scorechain", "kmercount", "snpphase", "snpvalid", "lgspolish"};

static void stamp(FILE* f) {
    time_t t = time(nullptr);
    struct tm* lt = localtime(&t);
    fprintf(f, "[ %02d-%02d-%02d %02d:%02d:%02d ] ", lt->tm_year + 1900, lt->tm_mon + 1, lt->tm_mday, lt->tm_hour,
            lt->tm_min, lt->tm_sec);
}

int main(int argc, char* argv[]) {
    int step = 0;
    if (argc > 1)
        for (int i = 0; i < 5; ++i)
            if (strcmp(kCmds[i], argv[1]) == 0) step = i + 1;
    if (step == 0) {
        printf("Usage: %s <command> [options]\n\nCommands:\n"
               "\tscorechain\t\tscore chain run\n\t\t\t\teg. scorechain fastafn sgsbamf > output.fa\n"
               "\tkmercount\t\tkmer count run\n\t\t\t\teg. kmercount fastafn sgsbamf > output.fa\n"
               "\tsnpphase\t\tsnp phase run\n\t\t\t\teg. snpphase fastafn sgsbamf lgsbamf > output.fa\n"
               "\tsnpvalid\t\tsnp valid run\n\t\t\t\teg. snpvalid fastafn sgsbamf > output.fa\n"
               "\tlgspolish \t\tlgs polish run\n\t\t\t\teg. lgspolish fasta_file lgsbamf > output.fa\n\n",
               argv[0]);
        return 0;
    }
    if ((step == 3 && argc != 5) || (step != 3 && argc != 4)) {
        if (step == 3) printf("%s %s fastafn bamfn thirdbamfn\n", argv[0], argv[1]);
        else printf("%s %s fastafn lgsbam\n", argv[0], argv[1]);
        return 0;
    }
    time_t t_start = time(nullptr);
#if defined(F_SETPIPE_SZ)
    {   // stdout into a pipe (the usual `> output.fa` is a file): the largest pipe the system gives, so that a contig's text leaves in a few writes
        struct stat so;
        if (fstat(1, &so) == 0 && S_ISFIFO(so.st_mode)) (void)fcntl(1, F_SETPIPE_SZ, 1 << 20);
    }
#endif
    Configure* cfg = (step == 5) ? config_init(argv[2], nullptr, argv[3]) : config_init(argv[2], argv[3], argc > 4 ? argv[4] : nullptr);
    if (step == 1 || step == 2) {
        if (!cfg->bamfn) { fprintf(stderr, "cannot access BAM %s\n", argv[3]); return 1; }
        // contigs flow through the device in batches (FASTA-index order), loaders and lanes overlapped: np1_pipe.cpp
        int dev = 0, lanes = 3;      // (from files the device ingest of one batch runs under the kernels of the others: 3 lanes > 2 > 4, tests/tools/r4_e2e_lanes.py)
        long long batch_bp = 16000000;
        if (const char* e = getenv("NP1_DEVICE")) dev = atoi(e);
        if (const char* e = getenv("NP1_LANES")) lanes = atoi(e);
        if (const char* e = getenv("NP1_BATCH_BP")) batch_bp = atoll(e);
        timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
        np1_pipe* pipe = np1_pipe_open(dev, lanes);
        if (getenv("NP1_TIMING")) { timespec ts1; clock_gettime(CLOCK_MONOTONIC, &ts1); fprintf(stderr, "[np1 cli] device lanes open after %.1f ms\n", (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6); }
        if (!pipe) { fprintf(stderr, "%s\n", np1_last_error()); return 1; }
        struct Out { int step; } out{step};
        auto sink = [](void* user, const char* name, const char* seq, int64_t len) {
            printf(">%s_%d\n", name, static_cast<Out*>(user)->step);
            fwrite(seq, 1, (size_t)len, stdout);
            fputc('\n', stdout);
        };
        // NP1_TILE_BP (score_chain): contigs longer than this are polished as independent tiles with NP1_TILE_HALO bases of halo (default
        // 1000) and joined exactly -- a contig need not fit one HBM batch (np1_tile.cpp; the reference takes contigs up to 2^31 bases)
        const long long tile_bp = (step == 1 && getenv("NP1_TILE_BP")) ? atoll(getenv("NP1_TILE_BP")) : 0;
        const long long halo_bp = getenv("NP1_TILE_HALO") ? atoll(getenv("NP1_TILE_HALO")) : 1000;
        const int rc = tile_bp > 0 ? np1_run_files_tiled(pipe, dev, cfg->fastafn, cfg->bamfn, batch_bp, tile_bp, halo_bp, cfg, sink, &out)
                                   : np1_pipe_run_files(pipe, cfg->fastafn, cfg->bamfn, nullptr, 0, batch_bp, cfg, step, sink, &out);
        if (rc != 0) {
            fprintf(stderr, "%s\n", np1_last_error());
            return 1;
        }
        np1_pipe_close(pipe);
    } else if (step == 3) {
        // snp_phase: contigs in FASTA-index order, batches of NP1_BATCH_BP draft bases; per batch the short-read and the long-read
        // records become two resident batches and one np1_batch_snp_phase pass (np1_phase_device.hip)
        if (!cfg->bamfn) { fprintf(stderr, "cannot access BAM %s\n", argv[3]); return 1; }
        if (!cfg->thirdbamfn) { fprintf(stderr, "cannot access BAM %s\n", argv[4]); return 1; }
        int dev = 0, lanes = 3;
        long long batch_bp = 4000000;     // small batches, several lanes: the blocks of one batch are copied and inflated under the kernels of the others
        if (const char* e = getenv("NP1_DEVICE")) dev = atoi(e);
        if (const char* e = getenv("NP1_BATCH_BP")) batch_bp = atoll(e);
        if (const char* e = getenv("NP1_LANES")) lanes = atoi(e);
        if (lanes < 1) lanes = 1;
        np1_pipe* pipe = np1_pipe_open(dev, lanes);
        if (!pipe) { fprintf(stderr, "%s\n", np1_last_error()); return 1; }
        struct Out { int step; } out{step};
        auto sink = [](void* user, const char* name, const char* seq, int64_t len) {
            printf(">%s_%d\n", name, static_cast<Out*>(user)->step);
            fwrite(seq, 1, (size_t)len, stdout);
            fputc('\n', stdout);
        };
        if (np1_pipe_run_phase_files(pipe, cfg->fastafn, cfg->bamfn, cfg->thirdbamfn, nullptr, 0, batch_bp, cfg, sink, &out) != 0) {
            fprintf(stderr, "%s\n", np1_last_error());
            return 1;
        }
        np1_pipe_close(pipe);
    } else {
        PolishResult* (*fn)(const char*, Configure*) = step == 2 ? kmer_count : step == 3 ? snp_phase : step == 4 ? snp_valid : lgspolish;
        np1_stream* st = np1_stream_load(cfg->fastafn, cfg->bamfn ? cfg->bamfn : argv[3], nullptr, 0, 0);
        if (!st) { fprintf(stderr, "%s\n", np1_last_error()); return 1; }
        np1_stream_view v;
        np1_stream_get_view(st, &v);
        for (int64_t c = 0; c < v.n_contigs; ++c) {
            PolishResult* r = fn(np1_stream_contig_name(st, c), cfg);
            printf(">%s_%d\n%s\n", np1_stream_contig_name(st, c), step, r->contig);
            polishresult_destory(r);
        }
        np1_stream_free(st);
    }
    config_destory(cfg);
    stamp(stderr);
    fprintf(stderr, "total time:%ds\n", (int)(time(nullptr) - t_start));
    // Everything this process had to say is written; what is left is the HIP runtime taking itself apart and giving back device memory the
    // driver reclaims anyway -- 0.16 s of a 0.9 s run of 400 Mb (tests/tools/r6_cli_timeline.py).  Leave without it.
    fflush(stdout);
    fflush(stderr);
    _exit(0);
}
