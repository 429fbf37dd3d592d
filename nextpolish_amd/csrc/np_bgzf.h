// BGZF (blocked gzip) reader/writer, written from the SAM/BAM specification (SAMv1 §4.1).
// Replaces, for the polishing hot path only, the slice of the reference's vendored
// htslib that it uses for BAM access (reference: source/lib/contig.c:35-52,172-174,692-694
// call hts_open/bgzf_* / sam_itr_next; source/lib/config.c:80-101 calls bgzf_open/bam_read1).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <new>
#include <vector>

namespace np {

// Virtual file offset: (compressed block start << 16) | offset inside the inflated block.
typedef uint64_t voff_t;

class BgzfReader {
public:
    BgzfReader() = default;
    ~BgzfReader();
    BgzfReader(const BgzfReader&) = delete;
    BgzfReader& operator=(const BgzfReader&) = delete;

    bool open(const std::string& path);
    void close();
    // Reads exactly n bytes unless EOF; returns bytes read, or -1 on a corrupt block.
    int64_t read(void* dst, size_t n);
    // The next n inflated bytes as a pointer into the reader's window when they lie there contiguously with `slack`
    // readable bytes behind them (then they are consumed; valid until the next call on this reader), else nullptr and
    // nothing is consumed (the caller falls back to read()).
    const uint8_t* take_contiguous(size_t n, size_t slack);
    bool seek(voff_t v);
    voff_t tell() const;
    bool is_open() const { return fp_ != nullptr; }

private:
    // Blocks are independent deflate streams, so the reader inflates a window of consecutive blocks at once on a few
    // threads (NP_IO_THREADS, default min(8, cores); threads live only inside fill_window: fork-safe) and serves
    // read()/seek() from that window.  This is the "BGZF ingest" row of SURVEY.md section 8(f): inflate is ~60 % of
    // the reference's wall time.
    struct WinBlock { uint64_t coff; uint32_t total, isize; size_t cpos, upos; };
    bool load_block();                 // make the block at file offset next_coff_ current
    bool fill_window(uint64_t coff);   // read + inflate the blocks starting at coff
    FILE* fp_ = nullptr;
    // a window's bytes: storage that only grows and is NOT cleared (a std::vector zero-fills what resize() adds: with the batch inflater's
    // 48 MB windows that was ~40 ms per 5 Mb window of a long-read worker, more than the header walk it sat in)
    struct RawBuf {
        uint8_t* p = nullptr;
        size_t n = 0, cap = 0;
        RawBuf() = default;
        RawBuf(const RawBuf&) = delete;
        RawBuf& operator=(const RawBuf&) = delete;
        ~RawBuf() { free(p); }
        uint8_t* data() { return p; }
        const uint8_t* data() const { return p; }
        size_t size() const { return n; }
        void resize(size_t want) {
            if (want > cap) {
                free(p);
                cap = want + want / 8 + 4096;
                p = static_cast<uint8_t*>(malloc(cap));
                if (!p) { cap = 0; throw std::bad_alloc(); }
            }
            n = want;
        }
    };
    RawBuf cwin_;        // compressed window
    RawBuf uwin_;        // inflated window
    std::vector<WinBlock> win_;
    size_t win_i_ = 0;                 // current block inside win_
    const uint8_t* ubuf_ = nullptr;    // inflated current block (<= 64 KiB)
    uint64_t block_coff_ = 0;          // file offset of the current block
    uint64_t next_coff_ = 0;           // file offset of the next block
    uint32_t ulen_ = 0, upos_ = 0;
    bool eof_ = false;
    size_t batch_window_ = 0;          // batch inflater: current window size and where the last window ended
    uint64_t batch_next_coff_ = ~0ull;
};

class BgzfWriter {
public:
    BgzfWriter() = default;
    ~BgzfWriter();
    bool open(const std::string& path, int level = 1);
    // Memory sink: the compressed blocks are collected in memory() instead of a file, deflated on `threads` threads (several such
    // writers running side by side is how the big synthetic BAMs are written: one part per thread, np_synth.cpp); finish_memory()
    // flushes without the EOF marker.  Offsets (tell / resolve) are relative to the first block of this writer.
    bool open_memory(int level, unsigned threads);
    bool finish_memory();
    std::vector<uint8_t>& memory() { return mem_; }
    bool write(const void* src, size_t n);
    // Blocks are deflated a batch at a time on a few threads (NP_IO_THREADS), so the file offset of the current block is
    // not known while records are written: tell() returns a PROVISIONAL offset (block sequence number << 16 | offset in
    // the block), monotone in the real one; resolve() turns it into the real virtual offset once close() has run.
    voff_t tell() const { return ((uint64_t)(block_coff_.size() + pending_.size()) << 16) | (uint64_t)fill_; }
    voff_t resolve(voff_t provisional) const;
    bool flush_block();
    bool close();   // flushes and appends the 28-byte EOF marker block

private:
    bool drain();                      // deflate + write the pending blocks
    FILE* fp_ = nullptr;
    bool to_memory_ = false;
    unsigned threads_ = 0;                        // 0: NP_IO_THREADS
    std::vector<uint8_t> mem_;
    int level_ = 1;
    std::vector<uint8_t> ubuf_;
    std::vector<std::vector<uint8_t>> pending_;   // full blocks waiting for the deflate threads
    std::vector<uint64_t> block_coff_;            // file offset of every block written so far
    uint32_t fill_ = 0;
    uint64_t coff_ = 0;
};

// A batch inflater takes the blocks of one reader window at once (the long-read workers hand it to the device decoder of
// np_inflate_dev.h: np_bgzf_dev.hip).  comp = the window's file bytes, blocks[i] = payload offset / length inside comp and
// output offset / length inside out.  Returns false when it could not do the whole batch (the reader then inflates the
// window on its host threads).  `window_bytes` = how many compressed bytes the reader should gather per window while the
// batch inflater is installed (a device wants thousands of blocks per launch, the host threads are fine with 4 MiB).
struct BgzfBatchBlock { uint64_t in_off, out_off; uint32_t in_len, out_len; };
typedef bool (*bgzf_batch_inflate_fn)(const uint8_t* comp, size_t comp_len, const BgzfBatchBlock* blocks, size_t n, uint8_t* out, size_t out_len);
void set_bgzf_batch_inflater(bgzf_batch_inflate_fn fn, size_t window_bytes);

// Inflate one raw-deflate payload (used by the threaded whole-file loader).
bool bgzf_inflate_block(const uint8_t* cdata, size_t clen, uint8_t* out, size_t out_len);

// stage clocks of BgzfReader::fill_window summed since the last call (ms: read, block scan, batch inflate, host inflate; windows by inflater)
void bgzf_prof_take(double ms[4], uint64_t n[2]);

}  // namespace np
