// calgs: genome size = sum of sequence lengths of a (gzipped) FASTA/FASTQ
// (reference: source/lib/calgs.c:8-24, which drives the kseq reader; record grammar restated here:
// '>' or '@' header line, sequence lines up to the next '>', '@' or '+', FASTQ quality block skipped
// by length).  Host only.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/nextpolish1.h"

namespace {
struct GzIn {
    gzFile f;
    unsigned char buf[1 << 16];
    int n = 0, p = 0;
    bool eof = false;
    int getc() {
        if (p >= n) {
            if (eof) return -1;
            n = gzread(f, buf, sizeof(buf));
            p = 0;
            if (n <= 0) { eof = true; n = 0; return -1; }
        }
        return buf[p++];
    }
};
}  // namespace

extern "C" uint64_t calgs(const char* file) {
    gzFile fp = gzopen(file, "r");
    if (fp == NULL) {
        fprintf(stderr, "Error! %s does not exist!", file);
        exit(1);
    }
    GzIn in;
    in.f = fp;
    uint64_t gs = 0;
    int c;
    // skip to the first header
    while ((c = in.getc()) >= 0 && c != '>' && c != '@') {}
    while (c >= 0) {
        // header line
        while ((c = in.getc()) >= 0 && c != '\n') {}
        uint64_t seq_len = 0;
        // sequence lines: every character except '\n' counts; a '\r' that ends a line is dropped once the sequence is longer than
        // one character (kseq's line reader, source/util/kseq.h:141, so CRLF files give the same size as LF files)
        while ((c = in.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            ++seq_len;
            int last = c;
            while ((c = in.getc()) >= 0 && c != '\n') { ++seq_len; last = c; }
            if (seq_len > 1 && last == '\r') --seq_len;
            if (c < 0) break;
        }
        gs += seq_len;
        if (c == '+') {   // FASTQ: skip the '+' line, then quality characters until they match the sequence length
            while ((c = in.getc()) >= 0 && c != '\n') {}
            uint64_t q = 0;
            while (q < seq_len && (c = in.getc()) >= 0)
                if (c != '\n') ++q;
            while ((c = in.getc()) >= 0 && c != '>' && c != '@') {}
        }
    }
    gzclose(fp);
    return gs;
}
