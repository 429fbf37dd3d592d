// Batch inflater of the BGZF reader on the device: the long-read workers (nextpolish2.so) hand whole reader windows --
// thousands of independent BGZF blocks -- to the wave-per-block DEFLATE decoder of np_inflate_dev.h instead of inflating them on
// their few host threads (BGZF inflate was 0.35 of the 0.9 host CPU-seconds of a 5 Mb window).  Blocks the decoder does not
// accept are inflated by the host decoder afterwards, so a refusal costs time, never correctness.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "np_bgzf.h"
#include "np_inflate_dev.h"
#include "np_crc_dev.h"
#include "np_crc32.h"
#include "np_devalloc.h"
#include "np_hostcopy.h"

namespace {

__global__ __launch_bounds__(256, 4) void k_bgzf_inflate(const uint8_t* __restrict__ comp, const npdev::BlockDesc* __restrict__ blocks, uint32_t n_blocks,
                                                         uint8_t* out, uint32_t* __restrict__ status) {
    __shared__ npdev::InflateLds lds[4];
    const uint32_t wave = npdev::uni(threadIdx.x >> 6);
    const uint32_t b = blockIdx.x * 4 + wave;
    if (b >= n_blocks) return;
    const npdev::BlockDesc d = blocks[b];
    int rc = 0;
    if (d.out_len) rc = npdev::inflate_block_wave(comp + d.in_off, d.in_len, out + d.out_off, d.out_len, lds[wave]);
    if ((threadIdx.x & 63u) == 0) status[b] = (uint32_t)rc;
}

// gzip trailer CRC of the blocks the decoder accepted (np_crc_dev.h), like the host reader and the reference's htslib check it
__global__ __launch_bounds__(256) void k_bgzf_crc(const uint8_t* __restrict__ comp, const npdev::BlockDesc* __restrict__ blocks, uint32_t n_blocks,
                                                  const uint8_t* __restrict__ out, uint32_t* __restrict__ status, const uint32_t* __restrict__ shift) {
    npdev::crc_check_body(comp, blocks, n_blocks, out, status, shift);
}

struct DevInflater {
    int device = -1;
    hipStream_t q = nullptr;
    void *d_comp = nullptr, *d_out = nullptr, *d_blocks = nullptr, *d_status = nullptr, *d_shift = nullptr;
    size_t c_comp = 0, c_out = 0, c_blocks = 0;
    std::vector<uint32_t> status;
    std::mutex mu;
    uint64_t n_batches = 0, n_blocks = 0, n_host_blocks = 0;

    bool grow(void** p, size_t* cap, size_t want, bool host) {
        if (want <= *cap) return true;
        if (*p) { if (host) (void)npalloc::host_free(*p); else (void)npalloc::dev_free(*p); }
        *p = nullptr;
        *cap = 0;
        const size_t n = (!host && npalloc::efence()) ? want : want + want / 4 + (1u << 20);
        const hipError_t e = host ? npalloc::host_malloc(p, n, hipHostMallocPortable) : npalloc::dev_malloc(p, n);
        if (e != hipSuccess) { *p = nullptr; return false; }
        *cap = n;
        return true;
    }
    bool run(const uint8_t* comp, size_t comp_len, const np::BgzfBatchBlock* bl, size_t n, uint8_t* out, size_t out_len) {
        std::lock_guard<std::mutex> g(mu);
        if (hipSetDevice(device) != hipSuccess) return false;
        if (!q && npalloc::stream_create(&q) != hipSuccess) return false;
        size_t cb = c_blocks;
        if (!grow(&d_comp, &c_comp, comp_len + 4096, false) || !grow(&d_out, &c_out, out_len + 4096, false) ||
            !grow(&d_blocks, &cb, sizeof(npdev::BlockDesc) * n + 64, false))
            return false;
        if (cb != c_blocks || !d_status) {       // the status words follow the block table's size
            if (d_status) (void)npalloc::dev_free(d_status);
            d_status = nullptr;
            c_blocks = 0;                        // (stays 0 if the allocation fails: the next batch allocates again instead of launching on a null pointer)
            if (npalloc::dev_malloc(&d_status, cb / sizeof(npdev::BlockDesc) * 4 + 64) != hipSuccess) { d_status = nullptr; return false; }
            c_blocks = cb;
        }
        static const bool check_crc = getenv("NP_BGZF_NO_CRC") == nullptr;
        if (check_crc && !d_shift) {
            uint32_t t[64];
            npdev::crc_shift_table(t);
            if (npalloc::dev_malloc(&d_shift, sizeof(t)) != hipSuccess) { d_shift = nullptr; return false; }
            if (npcopy::h2d_sync(d_shift, t, sizeof(t)) != hipSuccess) return false;
        }
        static thread_local std::vector<npdev::BlockDesc> desc;
        desc.resize(n);
        for (size_t i = 0; i < n; ++i) desc[i] = npdev::BlockDesc{bl[i].in_off, bl[i].out_off, bl[i].in_len, bl[i].out_len};
        status.resize(n);
        // straight from / into the reader's own (pageable) windows: the runtime stages them, no copy of ours in between.  Once anything
        // is enqueued every way out waits for the stream first: the caller inflates into the same `out` window on its host threads
        // when this returns false
        bool ok = npcopy::h2d(d_comp, comp, comp_len, q) == hipSuccess &&
                  hipMemsetAsync((char*)d_comp + comp_len, 0, 4096, q) == hipSuccess &&
                  npcopy::h2d(d_blocks, desc.data(), sizeof(npdev::BlockDesc) * n, q) == hipSuccess;
        if (ok) {
            k_bgzf_inflate<<<(unsigned)((n + 3) / 4), 256, 0, q>>>((const uint8_t*)d_comp, (const npdev::BlockDesc*)d_blocks, (uint32_t)n, (uint8_t*)d_out, (uint32_t*)d_status);
            ok = hipGetLastError() == hipSuccess;
        }
        if (ok && check_crc) {
            k_bgzf_crc<<<(unsigned)((n + 3) / 4), 256, 0, q>>>((const uint8_t*)d_comp, (const npdev::BlockDesc*)d_blocks, (uint32_t)n, (const uint8_t*)d_out, (uint32_t*)d_status,
                                                               (const uint32_t*)d_shift);
            ok = hipGetLastError() == hipSuccess;
        }
        ok = ok && npcopy::d2h(out, d_out, out_len, q) == hipSuccess;
        ok = ok && npcopy::d2h(status.data(), d_status, 4 * n, q) == hipSuccess;
        const bool synced = hipStreamSynchronize(q) == hipSuccess;
        if (!ok || !synced) return false;
        ++n_batches;
        n_blocks += n;
        for (size_t i = 0; i < n; ++i) {
            if (!status[i]) continue;
            ++n_host_blocks;          // refused by the device decoder, or its CRC differs: the host decoder has the last word
            if (!np::bgzf_inflate_block(comp + bl[i].in_off, bl[i].in_len, out + bl[i].out_off, bl[i].out_len)) return false;
            if (check_crc && bl[i].out_len) {
                uint32_t want;
                memcpy(&want, comp + bl[i].in_off + bl[i].in_len, 4);
                if (np::crc32_block(out + bl[i].out_off, bl[i].out_len) != want) return false;   // the reader's own pass then fails on the same block
            }
        }
        return true;
    }
};

int g_device = -1;
std::mutex g_pool_mu;
std::vector<DevInflater*> g_free;      // inflaters not in use (stream + device buffers each); as many get built as batches overlap

// The decode threads of a worker read different parts of the file and live only for one call: each batch checks an inflater
// out of the pool, so the batches overlap on the device and the buffers outlive the threads.
bool batch_hook(const uint8_t* comp, size_t comp_len, const np::BgzfBatchBlock* blocks, size_t n, uint8_t* out, size_t out_len) {
    DevInflater* inf = nullptr;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        if (!g_free.empty()) { inf = g_free.back(); g_free.pop_back(); }
    }
    if (!inf) { inf = new DevInflater; inf->device = g_device; }
    const bool ok = inf->run(comp, comp_len, blocks, n, out, out_len);
    std::lock_guard<std::mutex> g(g_pool_mu);
    g_free.push_back(inf);
    return ok;
}

}  // namespace

namespace np {

// Installs the device inflater for every BgzfReader of this process (call after the HIP device of the worker is chosen).
// Opt-in (NP2_INFLATE=device): it takes 43 % of a worker's host CPU time away (0.160 -> 0.091 CPU-s per Mbp) but adds ~15 ms of
// GPU time per 5 Mb window, and eight workers sharing one GPU are GPU-bound, so the default keeps the host
// threads.  NP2_INFLATE_WINDOW_MB = compressed bytes gathered per launch at most (default 48; sequential reads grow towards it).
void bgzf_device_inflate_enable(int device) {
    const char* e = getenv("NP2_INFLATE");
    if (!e || strcmp(e, "device") != 0) return;
    g_device = device;
    size_t mb = 48;
    if (const char* w = getenv("NP2_INFLATE_WINDOW_MB")) mb = (size_t)atoi(w);
    if (mb < 4) mb = 4;
    set_bgzf_batch_inflater(batch_hook, mb << 20);
}

}  // namespace np
