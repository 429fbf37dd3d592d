// Device allocations of both libraries go through here.
//
// Default: hipMalloc / hipFree.
//
// NP_EFENCE=1 (debugging): every buffer is placed through the HIP virtual-memory API so that its LAST byte (rounded up to 16, the
// widest vector access the kernels use) is the last mapped byte of its own address reservation, with an unmapped granule behind it.
// A kernel that reads or writes past the end of a buffer then faults at the first such access, deterministically, instead of
// touching whatever the allocator happened to place there (how the over-reads behind the round-3 abort were
// found).  The mode costs one granule (2 MiB) of physical memory per buffer and a few hundred microseconds per allocation.
//
// NP_DEVPOISON=1 (debugging; or a hex byte, e.g. NP_DEVPOISON=ff): every new device buffer is filled with that byte (default a5) before it
// is handed out.  hipMalloc hands a long-lived process the bytes of whatever it freed before -- a fresh process mostly sees zeros -- so a
// kernel that reads memory nobody wrote behaves differently in the one-process test suite than in any single test; with the poison it
// misbehaves the same way everywhere.  NP_DEVPOISON=2 also fills the work buffers of a short-read batch before
// every run (what an earlier run left in a reused buffer is the other thing only a long-lived process has).  Works with and without NP_EFENCE.
#pragma once
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>

namespace npalloc {

// NP_ALLOCLOG=<path> (debugging): every device allocation, pinned host allocation and host registration of both libraries, and every
// release, is appended to <path>.<pid> as one line "<ms> <tid> <op> <address> <bytes>" (ops D+ D- device, H+ H- hipHostMalloc; round 5's hunt also logged R+ R- hipHostRegister,
// which the libraries no longer use), written with one write(2) each so that the file is complete when the runtime ends the process on a GPU memory fault:
// the address in the runtime's "Memory access fault by GPU ... on address" line is then looked up among the ranges that were live, or had
// just been released, at that moment (tests/tools/r5_fault_lookup.py).
inline int alloc_log_fd() {
    static const int fd = [] {
        const char* e = getenv("NP_ALLOCLOG");
        if (!e || !e[0]) return -1;
        char path[600];
        snprintf(path, sizeof path, "%.500s.%d", e, (int)getpid());
        return open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    }();
    return fd;
}
inline void alloc_log(const char* op, const void* p, size_t bytes) {
    const int fd = alloc_log_fd();
    if (fd < 0) return;
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    char line[160];
    const int n = snprintf(line, sizeof line, "%lld.%03ld %ld %s %p %zu\n", (long long)ts.tv_sec, ts.tv_nsec / 1000000, (long)syscall(SYS_gettid), op, p, bytes);
    if (n > 0) (void)!write(fd, line, (size_t)n);
}

// Page-locked host memory of both libraries comes from here: hipHostMalloc, and a list of the ranges handed out -- np_hostcopy.h lets
// the DMA engines touch host memory only when it lies in one of them (never the user's pageable pages, never a hipHostRegister of heap
// memory: DESIGN.md section 12).
struct PinnedRanges {
    std::mutex mu;
    std::map<uintptr_t, size_t> live;      // start -> bytes
};
inline PinnedRanges& pinned_ranges() { static PinnedRanges* r = new PinnedRanges(); return *r; }
inline bool is_pinned(const void* p, size_t bytes) {
    PinnedRanges& R = pinned_ranges();
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> g(R.mu);
    auto it = R.live.upper_bound(a);
    if (it == R.live.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
}
inline hipError_t host_malloc(void** p, size_t bytes, unsigned flags) {
    const hipError_t e = hipHostMalloc(p, bytes, flags);
    if (e == hipSuccess) {
        alloc_log("H+", *p, bytes);
        PinnedRanges& R = pinned_ranges();
        std::lock_guard<std::mutex> g(R.mu);
        R.live[(uintptr_t)*p] = bytes;
    }
    return e;
}
inline hipError_t host_free(void* p) {
    if (!p) return hipSuccess;
    alloc_log("H-", p, 0);
    {
        PinnedRanges& R = pinned_ranges();
        std::lock_guard<std::mutex> g(R.mu);
        R.live.erase((uintptr_t)p);
    }
    return hipHostFree(p);
}

inline bool efence() {
    static const bool on = [] { const char* e = getenv("NP_EFENCE"); return e && e[0] && e[0] != '0'; }();
    return on;
}

struct FenceRec { void* base; size_t reserved, mapped; hipMemGenericAllocationHandle_t handle; };
struct FenceTable {
    std::mutex mu;
    std::unordered_map<void*, FenceRec> live;
};
inline FenceTable& fence_table() { static FenceTable t; return t; }

inline int poison() {      // -1: off
    static const int v = [] {
        const char* e = getenv("NP_DEVPOISON");
        if (!e || !e[0] || !strcmp(e, "0")) return -1;
        if (!strcmp(e, "1") || !strcmp(e, "2")) return 0xa5;
        return (int)(strtol(e, nullptr, 16) & 0xff);
    }();
    return v;
}

inline bool poison_runs() {      // NP_DEVPOISON=2: also the work buffers of a batch before every run (np1_batch::poison_work)
    static const bool on = [] { const char* e = getenv("NP_DEVPOISON"); return poison() >= 0 && e && !strcmp(e, "2"); }();
    return on;
}

inline hipError_t dev_malloc_raw(void** p, size_t bytes);
inline hipError_t dev_malloc(void** p, size_t bytes) {
    const hipError_t e = dev_malloc_raw(p, bytes);
    if (e == hipSuccess) alloc_log("D+", *p, bytes);
    if (e == hipSuccess && poison() >= 0 && bytes) {
        (void)hipMemset(*p, poison(), bytes);
        (void)hipDeviceSynchronize();
    }
    return e;
}

inline hipError_t dev_malloc_raw(void** p, size_t bytes) {
    if (!efence()) return hipMalloc(p, bytes);
    *p = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran == 0) gran = (size_t)2 << 20;
    const size_t need = (bytes + 15) & ~(size_t)15;
    const size_t mapped = ((need ? need : 16) + gran - 1) / gran * gran;
    FenceRec r{nullptr, mapped + gran, mapped, {}};
    e = hipMemAddressReserve(&r.base, r.reserved, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&r.handle, mapped, &prop, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(r.base, r.reserved); return e; }
    e = hipMemMap(r.base, mapped, 0, r.handle, 0);
    if (e != hipSuccess) { (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.base, r.reserved); return e; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(r.base, mapped, &acc, 1);
    if (e != hipSuccess) { (void)hipMemUnmap(r.base, mapped); (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.base, r.reserved); return e; }
    void* user = static_cast<char*>(r.base) + (mapped - need);
    {
        FenceTable& t = fence_table();
        std::lock_guard<std::mutex> g(t.mu);
        t.live[user] = r;
    }
    *p = user;
    return hipSuccess;
}

inline hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    alloc_log("D-", p, 0);
    if (!efence()) return hipFree(p);
    FenceRec r;
    {
        FenceTable& t = fence_table();
        std::lock_guard<std::mutex> g(t.mu);
        auto it = t.live.find(p);
        if (it == t.live.end()) return hipFree(p);      // (allocated before the mode was read: cannot happen, the flag is read once)
        r = it->second;
        t.live.erase(it);
    }
    (void)hipDeviceSynchronize();                        // hipFree waits for work in flight; so does this
    (void)hipMemUnmap(r.base, r.mapped);
    (void)hipMemRelease(r.handle);
    // The address range is NOT handed back: a range that is reserved and mapped again showed stale contents on this runtime (ROCm 7.2:
    // tests/tools/efence_probe.hip, third reuse of one range: 43 % of a buffer wrong after a fill kernel + hipMemsetAsync), and a debugging
    // mode can afford to leak address space (47 bits of it) -- a freed buffer's range also stays unmapped, so a use after free faults too.
    return hipSuccess;
}

}  // namespace npalloc
