// BGZF reader/writer (SAMv1 §4.1).  See np_bgzf.h for the reference call sites this replaces.
#include "np_bgzf.h"
#include "np_threads.h"
#include "np_crc32.h"
#include "np_inflate.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <time.h>
#include <thread>

namespace np {

static const size_t kMaxBlock = 65536;       // inflated size limit of one BGZF block
static const size_t kWriteFill = 0xff00;     // flush threshold used by common writers

bool bgzf_inflate_block(const uint8_t* cdata, size_t clen, uint8_t* out, size_t out_len) {
    static const bool use_zlib_only = getenv("NP_INFLATE_ZLIB") != nullptr;
    if (!use_zlib_only && inflate_raw(cdata, clen, out, out_len)) return true;   // own decoder first (np_inflate.cpp), zlib as the referee
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(cdata);
    zs.avail_in = (uInt)clen;
    zs.next_out = out;
    zs.avail_out = (uInt)out_len;
    int ret = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return ret == Z_STREAM_END && zs.avail_out == 0;
}

BgzfReader::~BgzfReader() { close(); }

bool BgzfReader::open(const std::string& path) {
    close();
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) return false;
    win_.clear();
    win_i_ = 0;
    ubuf_ = nullptr;
    block_coff_ = next_coff_ = 0;
    ulen_ = upos_ = 0;
    eof_ = false;
    return true;
}

void BgzfReader::close() {
    if (fp_) fclose(fp_);
    fp_ = nullptr;
}

static unsigned io_threads() {
    static unsigned n = 0;
    if (!n) {
        const char* e = getenv("NP_IO_THREADS");
        unsigned hw = std::thread::hardware_concurrency();
        n = e ? (unsigned)atoi(e) : (hw ? (hw < 8 ? hw : 8) : 4);
        if (n < 1) n = 1;
        if (n > 64) n = 64;
    }
    return n;
}

// stage clocks of fill_window, summed over the process (NP2_TIMING prints them per polished window: np2_pipeline.cpp)
static std::atomic<uint64_t> g_prof_ns[4];      // fread, block scan, batch (device) inflate, host inflate
static std::atomic<uint64_t> g_prof_n[2];       // windows inflated by the batch hook / on the host
static inline uint64_t prof_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }
void bgzf_prof_take(double ms[4], uint64_t n[2]) {
    for (int i = 0; i < 4; ++i) ms[i] = g_prof_ns[i].exchange(0) * 1e-6;
    for (int i = 0; i < 2; ++i) n[i] = g_prof_n[i].exchange(0);
}
static bgzf_batch_inflate_fn g_batch_fn = nullptr;
static size_t g_batch_window = 0;
void set_bgzf_batch_inflater(bgzf_batch_inflate_fn fn, size_t window_bytes) { g_batch_fn = fn; g_batch_window = window_bytes; }

// Reads kWindow compressed bytes from `coff`, splits them into BGZF blocks (gzip member header with the 'BC' extra
// subfield, SAMv1 4.1) and inflates the complete ones in parallel.
bool BgzfReader::fill_window(uint64_t coff) {
    // host threads: 4 MiB windows.  Batch inflater: windows double while the reader keeps reading on sequentially (a short
    // contig must not pay for tens of MB it will never look at) up to the batch size.
    size_t kWindow = (size_t)(4u << 20);
    if (g_batch_fn && g_batch_window) {
        batch_window_ = (coff == batch_next_coff_ && batch_window_) ? std::min(batch_window_ * 2, g_batch_window) : (size_t)(4u << 20);
        kWindow = batch_window_;
    }
    win_.clear();
    win_i_ = 0;
    if (fseeko(fp_, (off_t)coff, SEEK_SET) != 0) return false;
    cwin_.resize(kWindow + kMaxBlock + 64);
    const uint64_t pt0 = prof_now();
    const size_t got = fread(cwin_.data(), 1, cwin_.size(), fp_);
    const uint64_t pt1 = prof_now();
    g_prof_ns[0] += pt1 - pt0;
    size_t p = 0, utotal = 0;
    while (p + 18 <= got) {
        const uint8_t* h = cwin_.data() + p;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (p + 12 + xlen > got) break;
        int bsize = -1;
        for (uint32_t i = 0; i + 4 <= xlen;) {
            const uint8_t* x = h + 12 + i;
            const uint32_t slen = x[2] | (x[3] << 8);
            if (x[0] == 'B' && x[1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = x[4] | (x[5] << 8);
            i += 4 + slen;
        }
        if (bsize < 0) return false;
        const size_t total = (size_t)bsize + 1, hdr_len = 12 + xlen;
        if (total < hdr_len + 8) return false;
        if (p + total > got) break;   // incomplete block: the next window starts here
        uint32_t isize;
        memcpy(&isize, h + total - 4, 4);
        if (isize > kMaxBlock) return false;
        win_.push_back(WinBlock{coff + p, (uint32_t)total, isize, p + hdr_len, utotal});
        utotal += isize;
        p += total;
        if (p >= kWindow) break;
    }
    if (win_.empty()) return got == 0 || got < 18 ? (got == 0) : false;   // clean EOF only when nothing is left
    uwin_.resize(utotal + 8);
    memset(uwin_.data() + utotal, 0, 8);      // (the slack behind the last block reads as zero, as it did when the storage was a vector)
    const uint64_t pt2 = prof_now();
    g_prof_ns[1] += pt2 - pt1;
    struct Clock { uint64_t t0; int slot; ~Clock() { g_prof_ns[slot] += prof_now() - t0; ++g_prof_n[slot - 2]; } };
    if (g_batch_fn && win_.size() >= 64) {     // enough blocks to fill a device: one launch for the whole window (the hook checks the CRCs itself: np_bgzf_dev.hip)
        std::vector<BgzfBatchBlock> bb(win_.size());
        for (size_t i = 0; i < win_.size(); ++i) {
            const WinBlock& b = win_[i];
            bb[i] = BgzfBatchBlock{(uint64_t)b.cpos, (uint64_t)b.upos, (uint32_t)(b.total - (b.cpos - (size_t)(b.coff - coff)) - 8), b.isize};
        }
        batch_next_coff_ = win_.back().coff + win_.back().total;
        Clock c{prof_now(), 2};
        if (g_batch_fn(cwin_.data(), got, bb.data(), bb.size(), uwin_.data(), utotal)) return true;
    } else if (g_batch_fn && !win_.empty()) {
        batch_next_coff_ = win_.back().coff + win_.back().total;
    }
    const unsigned nt = io_threads();
    static const bool check_crc = getenv("NP_BGZF_NO_CRC") == nullptr;   // on unless switched off
    Clock c_host{prof_now(), 3};
    std::atomic<size_t> next(0);
    std::atomic<bool> ok(true);
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= win_.size()) break;
            const WinBlock& b = win_[i];
            const size_t clen = b.total - (b.cpos - (size_t)(b.coff - coff)) - 8;
            if (b.isize && !bgzf_inflate_block(cwin_.data() + b.cpos, clen, uwin_.data() + b.upos, b.isize)) ok = false;
            if (b.isize && check_crc) {   // gzip trailer: CRC32 of the inflated bytes (htslib rejects a block whose CRC does not match)
                uint32_t want;
                memcpy(&want, cwin_.data() + b.cpos + clen, 4);
                if (crc32_block(uwin_.data() + b.upos, b.isize) != want) ok = false;
            }
        }
    };
    if (nt <= 1 || win_.size() < 4) {
        work();
    } else {
        std::vector<std::thread> th;
        const unsigned n = (unsigned)std::min<size_t>(nt, win_.size());
        spawn_helpers(th, n - 1, work);
        work();
        for (std::thread& t : th) t.join();
    }
    return ok;
}

bool BgzfReader::load_block() {
    // the next block of the current window?
    size_t i = win_.size();
    if (win_i_ + 1 < win_.size() && win_[win_i_ + 1].coff == next_coff_) i = win_i_ + 1;
    else
        for (size_t k = 0; k < win_.size(); ++k)
            if (win_[k].coff == next_coff_) { i = k; break; }
    if (i == win_.size()) {
        if (!fill_window(next_coff_)) return false;
        if (win_.empty()) { eof_ = true; ulen_ = upos_ = 0; block_coff_ = next_coff_; return true; }
        i = 0;
    }
    win_i_ = i;
    const WinBlock& b = win_[i];
    ubuf_ = uwin_.data() + b.upos;
    block_coff_ = b.coff;
    next_coff_ = b.coff + b.total;
    ulen_ = b.isize;
    upos_ = 0;
    return true;
}

int64_t BgzfReader::read(void* dst, size_t n) {
    uint8_t* out = static_cast<uint8_t*>(dst);
    size_t done = 0;
    while (done < n) {
        if (upos_ == ulen_) {
            if (eof_) break;
            if (!load_block()) return -1;
            if (eof_) break;
            continue;
        }
        size_t take = ulen_ - upos_;
        if (take > n - done) take = n - done;
        memcpy(out + done, ubuf_ + upos_, take);
        upos_ += (uint32_t)take;
        done += take;
    }
    return (int64_t)done;
}

const uint8_t* BgzfReader::take_contiguous(size_t n, size_t slack) {
    if (upos_ == ulen_) {
        if (eof_ || !load_block() || eof_) return nullptr;
    }
    if (win_.empty() || win_i_ >= win_.size()) return nullptr;
    const uint8_t* p = ubuf_ + upos_;
    const uint8_t* win_end = uwin_.data() + win_.back().upos + win_.back().isize;
    if ((size_t)(win_end - p) < n + slack) return nullptr;
    // advance over the blocks the bytes span (the window holds them inflated back to back)
    size_t left = n;
    for (;;) {
        const size_t here = ulen_ - upos_;
        if (left <= here) { upos_ += (uint32_t)left; break; }
        left -= here;
        const WinBlock& b = win_[++win_i_];   // exists: n fits before the end of the window
        ubuf_ = uwin_.data() + b.upos;
        block_coff_ = b.coff;
        next_coff_ = b.coff + b.total;
        ulen_ = b.isize;
        upos_ = 0;
    }
    return p;
}

bool BgzfReader::seek(voff_t v) {
    uint64_t coff = v >> 16;
    uint32_t uoff = (uint32_t)(v & 0xffff);
    eof_ = false;
    if (coff != block_coff_ || ulen_ == 0) {
        next_coff_ = coff;
        if (!load_block()) return false;
    }
    if (uoff > ulen_) return false;
    upos_ = uoff;
    return true;
}

voff_t BgzfReader::tell() const {
    // htslib convention: a position at the end of a block is reported as the start of the next
    if (upos_ == ulen_ && ulen_ != 0) return next_coff_ << 16;
    return (block_coff_ << 16) | upos_;
}

BgzfWriter::~BgzfWriter() { if (fp_) close(); }

bool BgzfWriter::open(const std::string& path, int level) {
    to_memory_ = false;
    threads_ = 0;
    fp_ = fopen(path.c_str(), "wb");
    if (!fp_) return false;
    level_ = level;
    ubuf_.resize(kMaxBlock);
    pending_.clear();
    block_coff_.clear();
    fill_ = 0;
    coff_ = 0;
    return true;
}

bool BgzfWriter::open_memory(int level, unsigned threads) {
    fp_ = nullptr;
    to_memory_ = true;
    threads_ = threads ? threads : 1;
    mem_.clear();
    level_ = level;
    ubuf_.resize(kMaxBlock);
    pending_.clear();
    block_coff_.clear();
    fill_ = 0;
    coff_ = 0;
    return true;
}

bool BgzfWriter::finish_memory() { return to_memory_ && flush_block() && drain(); }

// One BGZF block (gzip member with the BC subfield) from `n` bytes; false if deflate fails.
static bool deflate_block(const uint8_t* in, uint32_t n, int level, std::vector<uint8_t>* out) {
    out->resize(kMaxBlock + 1024);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(in);
    zs.avail_in = n;
    zs.next_out = out->data() + 18;
    zs.avail_out = (uInt)(out->size() - 18 - 8);
    int ret = deflate(&zs, Z_FINISH);
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (ret != Z_STREAM_END) return false;
    size_t total = 18 + clen + 8;
    if (total > 65536) return false;  // cannot happen with fill <= 0xff00
    static const uint8_t magic[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(out->data(), magic, 16);
    (*out)[16] = (uint8_t)((total - 1) & 0xff);
    (*out)[17] = (uint8_t)((total - 1) >> 8);
    uint32_t crc = crc32_block(in, n);
    memcpy(out->data() + 18 + clen, &crc, 4);
    memcpy(out->data() + 18 + clen + 4, &n, 4);
    out->resize(total);
    return true;
}

bool BgzfWriter::drain() {
    if (pending_.empty()) return true;
    std::vector<std::vector<uint8_t>> comp(pending_.size());
    std::atomic<size_t> next(0);
    std::atomic<bool> ok(true);
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= pending_.size()) break;
            if (!deflate_block(pending_[i].data(), (uint32_t)pending_[i].size(), level_, &comp[i])) ok = false;
        }
    };
    const unsigned nt = (unsigned)std::min<size_t>(threads_ ? threads_ : io_threads(), pending_.size());
    std::vector<std::thread> th;
    if (nt > 1) spawn_helpers(th, nt - 1, work);
    work();
    for (std::thread& t : th) t.join();
    if (!ok) return false;
    for (size_t i = 0; i < comp.size(); ++i) {
        if (to_memory_) mem_.insert(mem_.end(), comp[i].begin(), comp[i].end());
        else if (fwrite(comp[i].data(), 1, comp[i].size(), fp_) != comp[i].size()) return false;
        block_coff_.push_back(coff_);
        coff_ += comp[i].size();
    }
    pending_.clear();
    return true;
}

bool BgzfWriter::flush_block() {
    if (fill_ == 0) return true;
    pending_.emplace_back(ubuf_.begin(), ubuf_.begin() + fill_);
    fill_ = 0;
    if (pending_.size() >= (to_memory_ && threads_ <= 1 ? 16u : 512u)) return drain();
    return true;
}

voff_t BgzfWriter::resolve(voff_t v) const {
    const uint64_t seq = v >> 16;
    const uint64_t c = seq < block_coff_.size() ? block_coff_[seq] : coff_;   // seq == #blocks: the position after the last one
    return (c << 16) | (v & 0xffff);
}

bool BgzfWriter::write(const void* src, size_t n) {
    const uint8_t* in = static_cast<const uint8_t*>(src);
    while (n) {
        size_t take = kWriteFill - fill_;
        if (take > n) take = n;
        memcpy(ubuf_.data() + fill_, in, take);
        fill_ += (uint32_t)take;
        in += take;
        n -= take;
        if (fill_ == kWriteFill && !flush_block()) return false;
    }
    return true;
}

bool BgzfWriter::close() {
    if (!fp_) return true;
    bool ok = flush_block() && drain();
    static const uint8_t eof_marker[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43,
                                           0x02, 0x00, 0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0};
    ok = ok && fwrite(eof_marker, 1, 28, fp_) == 28;
    ok = (fclose(fp_) == 0) && ok;
    fp_ = nullptr;
    return ok;
}

}  // namespace np
