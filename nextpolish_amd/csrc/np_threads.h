// Host-side parallel loop for independent items (low-quality regions, records).  The threads live only inside the
// call (created and joined here), so a process that forks afterwards (the reference's worker model) stays safe.
// NP_HOST_THREADS overrides the default of min(8, cores); 1 runs inline.
#pragma once
#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

namespace np {

inline unsigned host_threads() {
    static const unsigned n = [] {
        const char* e = getenv("NP_HOST_THREADS");
        if (e && atoi(e) > 0) return (unsigned)atoi(e);
        const unsigned hw = std::thread::hardware_concurrency();
        return hw == 0 ? 1u : (hw < 8u ? hw : 8u);
    }();
    return n;
}

// f(begin, end) over [0, n) in blocks of `grain`, handed out dynamically
template <class F>
void parallel_for(size_t n, size_t grain, F f) {
    if (grain == 0) grain = 1;
    const size_t blocks = (n + grain - 1) / grain;
    unsigned nt = host_threads();
    if (blocks < nt) nt = (unsigned)blocks;
    if (nt <= 1) { if (n) f((size_t)0, n); return; }
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= blocks) return;
            const size_t lo = b * grain, hi = lo + grain < n ? lo + grain : n;
            f(lo, hi);
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
}

}  // namespace np
