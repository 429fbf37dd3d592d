// Contig loading for the long-read library: read_ref / refs_destroy / 2-bit codec / QV track.
// Host only (the caller runs this in the parent before it forks its workers: source/lib/nextpolish2.py:184-191).
//
// Reference behaviour restated here:
//   read_ref            source/lib/ctg_cns.c:2269-2295   (record grammar of the kseq reader: source/lib/mseq.h:193-233)
//   set_ref_qv          source/lib/ctg_cns.c:2233-2267
//   seq2bit1 / bit2seq1 source/lib/bseq.c:87-124, table :7-16
//   refs_destroy        source/lib/ctg_cns.c:2199-2208
#include <zlib.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nextpolish2.h"

namespace {

// FASTA/FASTQ record reader with the grammar of the reference's reader: header name = up to the first
// whitespace, comment = rest of the header line, sequence = every character of the following lines up to a
// line-leading-or-not '>', '@' or '+' (the reference tests single characters, not line starts), a trailing '\r'
// of a line is dropped when the accumulated string is longer than one character.
class FastxReader {
  public:
    explicit FastxReader(gzFile f) : f_(f) {}
    // comment_valid: whether this record had a comment (the reference keeps the previous buffer otherwise)
    bool next(std::string* name, std::string* comment, bool* comment_valid, std::string* seq) {
        int c;
        if (last_ == 0) {
            while ((c = getc()) != -1 && c != '>' && c != '@') {}
            if (c == -1) return false;
            last_ = c;
        }
        seq->clear();
        name->clear();
        *comment_valid = false;
        // name: up to whitespace
        if (at_eof()) return false;
        int d = -1;
        while ((c = getc()) != -1) {
            if (isspace(c)) { d = c; break; }
            name->push_back((char)c);
        }
        if (d != -1 && d != '\n') {
            comment->clear();
            *comment_valid = true;
            read_line(comment);
        }
        while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq->push_back((char)c);
            read_line(seq);
        }
        if (c == '>' || c == '@') last_ = c;
        if (c != '+') return true;
        // FASTQ: skip the '+' line and as many quality characters as there are bases
        while ((c = getc()) != -1 && c != '\n') {}
        std::string qual;
        if (c != -1) {
            while (!at_eof() && qual.size() < seq->size()) read_line(&qual);
        }
        last_ = 0;
        return true;
    }

  private:
    bool fill() {
        if (eof_) return false;
        n_ = gzread(f_, buf_, sizeof(buf_));
        p_ = 0;
        if (n_ < (int)sizeof(buf_)) eof_ = true;
        if (n_ <= 0) { n_ = 0; return false; }
        return true;
    }
    bool at_eof() { return p_ >= n_ && eof_; }
    int getc() {
        if (p_ >= n_ && !fill()) return -1;
        return buf_[p_++];
    }
    void read_line(std::string* s) {   // appends up to '\n' (consumed), drops one trailing '\r' if the string is longer than 1
        int c;
        while ((c = getc()) != -1 && c != '\n') s->push_back((char)c);
        if (s->size() > 1 && s->back() == '\r') s->pop_back();
    }
    gzFile f_;
    unsigned char buf_[16384];
    int n_ = 0, p_ = 0, last_ = 0;
    bool eof_ = false;
};

const uint8_t kNt[128] = {   // A/a 0, C/c 1, G/g 2, T/t/U/u 3, everything else 4 (bseq.c:7-16)
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};

// The per-base quality summary a previous long-read round left in the FASTA comment (read back by ctg_cns.c:2233-2267): among the
// blank-separated words of the comment, the last one beginning with "node" carries the number of entries from its 8th character on, the last
// one beginning with "qv" carries the entries from its 6th character on: ':'-separated hexadecimal words of 64 bits = position (32 bits),
// identity, and two rates (10 bits each).  Words are read with strtoull like the reference reads them (so "0x" prefixes and trailing junk
// behave the same); entries the comment does not supply stay zero, and entries beyond the announced count are ignored (the reference writes
// them past the end of its array).
// The comment is not modified.  The reference splits its reader's buffer in place, and a later record WITHOUT a comment is parsed from what
// is left visible of that buffer -- the text up to the end of its first word, or up to the end of the first hexadecimal entry when that first
// word is the "qv" one that was read.  The return value is that visible length; read_ref cuts its copy of the buffer there.
size_t parse_ref_qv(const char* comment, ref_* r) {
    r->qv_l = 0;
    r->qv = nullptr;
    if (!comment) return 0;
    const char *count_word = nullptr, *qv_word = nullptr, *qv_end = nullptr, *first_end = nullptr;
    for (const char* w = comment; *w;) {
        while (*w == ' ') ++w;
        const char* e = w;
        while (*e && *e != ' ') ++e;
        const size_t n = (size_t)(e - w);
        if (n && !first_end) first_end = e;
        if (n >= 4 && memcmp(w, "node", 4) == 0) count_word = n > 7 ? w + 7 : "";
        if (n >= 2 && memcmp(w, "qv", 2) == 0) { qv_word = n > 5 ? w + 5 : e; qv_end = e; }
        w = e;
    }
    size_t visible = first_end ? (size_t)(first_end - comment) : strlen(comment);
    const uint32_t announced = count_word ? (uint32_t)atoi(count_word) : 0;      // (atoi stops at the blank that ends the word)
    if (!announced || !qv_word) return visible;
    if (qv_end == first_end) {      // the entries are part of the first word: the reference's split ends the visible text behind the first entry
        const char* p = qv_word;
        while (p < qv_end && *p == ':') ++p;
        while (p < qv_end && *p != ':') ++p;
        if (p < qv_end && p > qv_word) visible = (size_t)(p - comment);
    }
    r->qv_l = announced;
    r->qv = (ref_qv*)calloc(announced, sizeof(ref_qv));
    uint32_t filled = 0;
    std::string word;
    for (const char* p = qv_word; p < qv_end && filled < announced;) {
        while (p < qv_end && *p == ':') ++p;
        const char* e = p;
        while (e < qv_end && *e != ':') ++e;
        if (e == p) break;
        word.assign(p, e);
        const uint64_t bits = strtoull(word.c_str(), nullptr, 16);
        ref_qv& q = r->qv[filled++];
        q.p = (uint32_t)(bits >> 32);
        q.ide = (bits >> 20) & 0x3ff;
        q.ort = (bits >> 10) & 0x3ff;
        q.irt = bits & 0x3ff;
        p = e;
    }
    return visible;
}

int str_cmp(const void* a, const void* b) { return strcmp(*(char* const*)a, *(char* const*)b); }

bool name_accepted(const char* target, char** array, int n) {   // binary search over the sorted names (ctg_cns.c:2210-2224)
    int low = 0, high = n;
    while (low < high) {
        const int middle = (low + high) / 2;
        const int c = strcmp(target, array[middle]);
        if (c == 0) return true;
        if (c < 0) high = middle;
        else low = middle + 1;
    }
    return false;
}

}  // namespace

extern "C" {

void seq2bit1(uint32_t* s, uint32_t len, char* seq) {
    uint32_t j = 0, base_cnt = 0, buffer = 0;
    for (uint32_t i = 0; i < len; ++i) {
        const uint8_t c = (uint8_t)seq[i];
        buffer = buffer << 2 | (c < 128 ? kNt[c] : 4u);   // 4 spills into the previous base's low bit, as in the reference
        if (++base_cnt == 16) {
            s[j++] = buffer;
            base_cnt = 0;
            buffer = 0;
        }
    }
    if (base_cnt) {
        buffer <<= (32 - (base_cnt << 1));
        s[j++] = buffer;
    }
}

void bit2seq1(uint32_t* s, uint32_t len, char* seq) {
    if (len == 0) { seq[0] = '\0'; return; }
    uint32_t j = len - 1;
    int32_t l = (int32_t)((len - 1) >> 4) + 1;
    int l_shift = (int)(((uint32_t)(l << 4) - len) << 1);
    while (l--) {
        uint32_t i = s[l];
        int i_m = 32;
        if (l_shift) {
            i_m -= l_shift;
            i >>= l_shift;
            l_shift = 0;
        }
        for (; i_m; i_m -= 2) {
            seq[j--] = "ACGT"[i & 3];
            i >>= 2;
        }
    }
    seq[len] = '\0';
}

refs_* read_ref(char* fasta, char** accept_names, int n) {
    gzFile fp = gzopen(fasta, "r");
    if (fp == nullptr) {
        fprintf(stderr, "Error! %s does not exist!", fasta);
        exit(1);
    }
    if (n) qsort(accept_names, (size_t)n, sizeof(char*), str_cmp);
    refs_* refs = (refs_*)malloc(sizeof(refs_));
    refs->i = 0;
    refs->i_m = 1000;
    refs->ref = (ref_*)calloc(refs->i_m, sizeof(ref_));
    FastxReader rd(fp);
    std::string name, comment, seq;
    std::vector<char> cbuf;      // the reader's comment buffer: persists (tokenised) across records without a comment
    bool have_cbuf = false;
    bool cv;
    while (rd.next(&name, &comment, &cv, &seq)) {
        if (cv) {
            cbuf.assign(comment.begin(), comment.end());
            cbuf.push_back('\0');
            have_cbuf = true;
        }
        if (n && !name_accepted(name.c_str(), accept_names, n)) continue;
        ref_* r = &refs->ref[refs->i];
        r->n = strdup(name.c_str());
        r->length = (uint32_t)seq.size();
        r->s = (uint32_t*)malloc(sizeof(uint32_t) * (r->length / 16 + 1));
        seq2bit1(r->s, r->length, seq.empty() ? (char*)"" : &seq[0]);
        const size_t visible = parse_ref_qv(have_cbuf ? cbuf.data() : nullptr, r);
        if (have_cbuf) { cbuf.resize(visible); cbuf.push_back('\0'); }
        if (++refs->i >= refs->i_m) {
            refs->i_m += 100;
            refs->ref = (ref_*)realloc(refs->ref, refs->i_m * sizeof(ref_));
            memset(refs->ref + refs->i_m - 100, 0, 100 * sizeof(ref_));
        }
    }
    gzclose(fp);
    return refs;
}

void refs_destroy(refs_* refs) {
    for (uint32_t i = 0; i < refs->i; ++i) {
        if (refs->ref[i].n) free(refs->ref[i].n);
        if (refs->ref[i].s) free(refs->ref[i].s);
        if (refs->ref[i].qv) free(refs->ref[i].qv);
    }
    free(refs->ref);
    free(refs);
}

}  // extern "C"
