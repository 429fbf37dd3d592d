// The device code of the pseudo-seed kernel (nextpolish_amd/csrc/np2_poa_dev.h, both size classes) run on the HOST: one OS thread per lane,
// 64 of them in lockstep, the wave intrinsics the header uses supplied here (shuffles, ballot, the LDS / global fences and
// readfirstlane -- which the header applies to every read of wave-uniform LDS state, always from uniform control flow -- are collectives over a barrier).  TEST INFRASTRUCTURE ONLY:
// it lets the CPU suite compare the device procedure with the host version (np2_poa.cpp, itself pinned to the reference's poa_to_consensus)
// on thousands of random regions without a GPU, and it is how round 5 debugged the two-class kernel.
#include <climits>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <pthread.h>
#include <string>
#include <thread>
#include <vector>

#define NP2_POA_HOST_EMU 1
#define __device__
#define __forceinline__ inline
#define __restrict__
#ifndef INT32_MIN
#define INT32_MIN (-2147483647 - 1)
#endif

namespace emu {
struct Barrier {
    pthread_barrier_t b;
    Barrier() { pthread_barrier_init(&b, nullptr, 64); }
    ~Barrier() { pthread_barrier_destroy(&b); }
    void wait() { pthread_barrier_wait(&b); }
};
Barrier* g_bar = nullptr;
long long g_slot[64];
thread_local unsigned t_lane = 0;
inline long long exchange(long long v, unsigned src_lane) {      // every lane deposits, every lane reads the lane it asks for
    g_slot[t_lane] = v;
    g_bar->wait();
    const long long r = g_slot[src_lane & 63u];
    g_bar->wait();
    return r;
}
}  // namespace emu

static inline unsigned __lane_id() { return emu::t_lane; }
// the value was read (by every lane, from wave-uniform state) before this call: nobody goes on -- and overwrites that state -- before everybody has read it
static inline int __builtin_amdgcn_readfirstlane(int v) { emu::g_bar->wait(); return v; }
#define __ATOMIC_ACQ_REL_EMU 0
static inline void emu_fence() { emu::g_bar->wait(); }
#define __builtin_amdgcn_fence(...) emu_fence()
static inline void __builtin_amdgcn_wave_barrier() {}
static inline int __shfl_up(int v, int d, int) { const unsigned l = emu::t_lane; const long long r = emu::exchange(v, l >= (unsigned)d ? l - (unsigned)d : l); return (int)r; }
static inline int __shfl(int v, int src, int) { return (int)emu::exchange(v, (unsigned)src); }
static inline int __shfl_xor(int v, int m, int) { return (int)emu::exchange(v, emu::t_lane ^ (unsigned)m); }
static inline unsigned long long __ballot(bool p) {
    emu::g_slot[emu::t_lane] = p ? 1 : 0;
    emu::g_bar->wait();
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) if (emu::g_slot[i]) m |= 1ull << i;
    emu::g_bar->wait();
    return m;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }

#include "../../nextpolish_amd/csrc/np2_poa_dev.h"

template <class C> static int run_class(const char* pool, const uint32_t* str_off, const uint32_t* str_len, uint32_t first, uint32_t n, char* out, uint32_t out_cap, uint32_t* out_len) {
    static np2poa::PoaLdsT<C> L;      // "LDS"
    std::vector<int32_t> TS(1u << 16);
    std::vector<uint32_t> TF(1u << 16);
    np2poa::Job J{first, n, 0ull, out_cap, 1u};
    emu::Barrier bar;
    emu::g_bar = &bar;
    bool ok[64];
    std::vector<std::thread> th;
    for (unsigned l = 0; l < 64; ++l)
        th.emplace_back([&, l] {
            emu::t_lane = l;
            ok[l] = np2poa::poa_region<C>(pool, str_off, str_len, J, TS.data(), TF.data(), 1u << 16, out, out_len, &L);
        });
    for (auto& t : th) t.join();
    for (unsigned l = 1; l < 64; ++l) if (ok[l] != ok[0]) return -2;      // a verdict that is not wave-uniform is a bug of the header
    return ok[0] ? 0 : 1;
}

extern "C" {
// seqs: n strings; cls 0 = Small, 1 = Big.  Returns 0 (out / *out_len filled), 1 (the class gives the region back) or -2.
int np2poa_emu(const char** seqs, int n, int cls, char* out, int cap) {
    std::string pool;
    std::vector<uint32_t> off, len;
    for (int i = 0; i < n; ++i) { off.push_back((uint32_t)pool.size()); len.push_back((uint32_t)strlen(seqs[i])); pool.append(seqs[i]); pool.push_back('\0'); }
    uint32_t out_len = 0;
    std::vector<char> buf((size_t)cap + 64, 0);
    const int rc = cls == 0 ? run_class<np2poa::Small>(pool.data(), off.data(), len.data(), 0, (uint32_t)n, buf.data(), (uint32_t)cap, &out_len)
                            : run_class<np2poa::Big>(pool.data(), off.data(), len.data(), 0, (uint32_t)n, buf.data(), (uint32_t)cap, &out_len);
    if (rc == 0) { memcpy(out, buf.data(), out_len); out[out_len] = '\0'; }
    return rc;
}
}
