// BGZF reader/writer (SAMv1 §4.1).  See np_bgzf.h for the reference call sites this replaces.
#include "np_bgzf.h"

#include <zlib.h>

#include <cstring>

namespace np {

static const size_t kMaxBlock = 65536;       // inflated size limit of one BGZF block
static const size_t kWriteFill = 0xff00;     // flush threshold used by common writers

bool bgzf_inflate_block(const uint8_t* cdata, size_t clen, uint8_t* out, size_t out_len) {
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
    cbuf_.resize(kMaxBlock + 64);
    ubuf_.resize(kMaxBlock);
    block_coff_ = next_coff_ = 0;
    ulen_ = upos_ = 0;
    eof_ = false;
    return true;
}

void BgzfReader::close() {
    if (fp_) fclose(fp_);
    fp_ = nullptr;
}

bool BgzfReader::load_block() {
    // gzip member header with the BGZF 'BC' extra subfield
    uint8_t hdr[18];
    if (fseeko(fp_, (off_t)next_coff_, SEEK_SET) != 0) return false;
    size_t got = fread(hdr, 1, 18, fp_);
    if (got == 0) { eof_ = true; ulen_ = upos_ = 0; block_coff_ = next_coff_; return true; }
    if (got != 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) return false;
    uint32_t xlen = hdr[10] | (hdr[11] << 8);
    // find BC subfield (almost always the first one)
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), hdr + 12, xlen < 6 ? xlen : 6);
    if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp_) != xlen - 6) return false;
    int bsize = -1;
    for (uint32_t i = 0; i + 4 <= xlen;) {
        uint32_t slen = extra[i + 2] | (extra[i + 3] << 8);
        if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= xlen)
            bsize = extra[i + 4] | (extra[i + 5] << 8);
        i += 4 + slen;
    }
    if (bsize < 0) return false;
    size_t total = (size_t)bsize + 1;
    size_t hdr_len = 12 + xlen;
    if (total < hdr_len + 8) return false;
    size_t clen = total - hdr_len - 8;
    if (clen + 8 > cbuf_.size()) cbuf_.resize(clen + 8);
    if (fread(cbuf_.data(), 1, clen + 8, fp_) != clen + 8) return false;
    uint32_t isize;
    memcpy(&isize, cbuf_.data() + clen + 4, 4);
    if (isize > kMaxBlock) return false;
    if (isize && !bgzf_inflate_block(cbuf_.data(), clen, ubuf_.data(), isize)) return false;
    block_coff_ = next_coff_;
    next_coff_ += total;
    ulen_ = isize;
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
        memcpy(out + done, ubuf_.data() + upos_, take);
        upos_ += (uint32_t)take;
        done += take;
    }
    return (int64_t)done;
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
    fp_ = fopen(path.c_str(), "wb");
    if (!fp_) return false;
    level_ = level;
    ubuf_.resize(kMaxBlock);
    cbuf_.resize(kMaxBlock + 1024);
    fill_ = 0;
    coff_ = 0;
    return true;
}

bool BgzfWriter::flush_block() {
    if (fill_ == 0) return true;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = ubuf_.data();
    zs.avail_in = fill_;
    zs.next_out = cbuf_.data() + 18;
    zs.avail_out = (uInt)(cbuf_.size() - 18 - 8);
    int ret = deflate(&zs, Z_FINISH);
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (ret != Z_STREAM_END) return false;
    size_t total = 18 + clen + 8;
    if (total > 65536) return false;  // cannot happen with fill <= 0xff00
    static const uint8_t magic[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(cbuf_.data(), magic, 16);
    cbuf_[16] = (uint8_t)((total - 1) & 0xff);
    cbuf_[17] = (uint8_t)((total - 1) >> 8);
    uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), ubuf_.data(), fill_);
    memcpy(cbuf_.data() + 18 + clen, &crc, 4);
    uint32_t isize = fill_;
    memcpy(cbuf_.data() + 18 + clen + 4, &isize, 4);
    if (fwrite(cbuf_.data(), 1, total, fp_) != total) return false;
    coff_ += total;
    fill_ = 0;
    return true;
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
    bool ok = flush_block();
    static const uint8_t eof_marker[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43,
                                           0x02, 0x00, 0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0};
    ok = ok && fwrite(eof_marker, 1, 28, fp_) == 28;
    ok = (fclose(fp_) == 0) && ok;
    fp_ = nullptr;
    return ok;
}

}  // namespace np
