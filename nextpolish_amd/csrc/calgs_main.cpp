// calgs CLI (reference: source/lib/calgs.c:26-31)
#include <cstdio>

#include "../../include/nextpolish1.h"

int main(int argc, char* argv[]) {
    if (argc < 2) { fprintf(stderr, "Usage: %s <fasta|fastq[.gz]>\n", argv[0]); return 1; }
    unsigned long gs = (unsigned long)calgs(argv[1]);
    printf("genome size: %lu bp\n", gs);
    return 0;
}
