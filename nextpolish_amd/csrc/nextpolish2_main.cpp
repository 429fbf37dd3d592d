// nextpolish2 CLI: `nextpolish2 ref.fa bam.fofn` -- ONT reads, 5 Mb windows, no splitting, every contig of the
// FASTA (reference: main of source/lib/ctg_cns.c:3625-3653; output format `>name_lgs len identity\nseq`).
#include <cassert>
#include <climits>
#include <cstdio>

#include <unistd.h>

#include "../../include/nextpolish2.h"

int main(int argc, char* argv[]) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s ref.fa bam.fofn\n", argv[0]);
        return 1;
    }
    refs_* refs = read_ref(argv[1], nullptr, 0);
    ctg_cns_cfg* cfg = ctg_cns_init(5000000, 1, 0, 0.8f, 0.8f, 0.8f);
    for (uint32_t i = 0; i < refs->i; ++i) {
        assert(refs->ref[i].length < INT_MAX);
        consensus_trimed_data* d = ctg_cns_core(cfg, &refs->ref[i], argv[2]);
        if (d->i_m > 1) {
            for (int j = 0; j < d->i_m; ++j)
                printf(">%s_%d_lgs %d %f\n%s\n", refs->ref[i].n, j, d->data[j].len, d->data[j].identity, d->data[j].seq);
        } else {
            printf(">%s_lgs %d %f\n%s\n", refs->ref[i].n, d->data[0].len, d->data[0].identity, d->data[0].seq);
        }
        free_consensus_trimed_data(d);
    }
    ctg_cns_destroy(cfg);
    refs_destroy(refs);
    fflush(stdout);      // (as nextpolish1: the output is written, what is left is the HIP runtime taking itself apart)
    fflush(stderr);
    _exit(0);
}
