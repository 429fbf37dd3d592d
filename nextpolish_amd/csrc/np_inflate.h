// Raw DEFLATE decoder for whole BGZF blocks (np_inflate.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

namespace np {
// Inflates exactly dst_len bytes from the raw DEFLATE stream src[0..src_len).  false = not accepted (the caller falls
// back to zlib): never a wrong answer, at worst a slower one.
bool inflate_raw(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len);
}  // namespace np
