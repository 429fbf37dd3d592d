// Host-side parallel loop for independent items (low-quality regions, records).  The threads live only inside the
// call (created and joined here), so a process that forks afterwards (the reference's worker model) stays safe.
// NP_HOST_THREADS overrides the default of min(8, cores); 1 runs inline.
#pragma once
#include <atomic>
#include <cstdlib>
#include <system_error>
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

// Starts up to `extra` helper threads on f.  A thread the system refuses (pthread_create: EAGAIN -- a long-lived process out of
// mappings or pids) is simply not there: every caller hands its work out dynamically and works itself, so fewer helpers mean less
// overlap, never less work; an exception escaping here with started threads still joinable would end the process instead.
template <class F>
unsigned spawn_helpers(std::vector<std::thread>& th, unsigned extra, F f) {
    unsigned started = 0;
    th.reserve(th.size() + extra);
    for (unsigned t = 0; t < extra; ++t) {
        try { th.emplace_back(f); } catch (const std::system_error&) { break; }
        ++started;
    }
    return started;
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
    spawn_helpers(th, nt - 1, work);
    work();
    for (std::thread& t : th) t.join();
}

}  // namespace np
