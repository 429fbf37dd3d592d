// Host <-> device copies of both libraries go through here.
//
// Why (round 5): on this runtime (ROCm 7.2) hipMemcpy / hipMemcpyAsync with PAGEABLE host memory is served three ways,
// by size (tests/tools/r5_stale_pin_probe.hip `thresh` under AMD_LOG_LEVEL=4, profiles/r5_fault_hunt.txt):
//   <= 1 MiB          staged through the runtime's own pinned buffer: the bytes are taken (H2D) while the call runs;
//   1 MiB .. 32 MiB   the USER's pages are locked in place ("Locking to pool", a userptr registration with the kernel driver), the DMA engine
//                     reads / writes them directly and the call RETURNS WITHOUT WAITING -- the copy is really asynchronous;
//   > 32 MiB          the same in chunks of 32 MiB, and the call waits for the last chunk.
// The middle class is what ended the one-process GPU suite: "Memory access fault by GPU ... on address 0x58a0ab503000" is an address of the
// brk heap, hit during the H2D copy of a 3.25 MB std::vector (np1_batch_upload).  A vector freed before its stream is synchronised -- legal
// under the "pageable copies are synchronous" assumption this code was written on -- is read by the DMA engine after the heap has given the
// pages back; and a userptr registration of heap pages is only as good as the kernel driver's tracking of every trim, growth and migration
// of those pages underneath it.
//
// So: the GPU never touches memory it did not get from hipHostMalloc.  Copies whose host side is page-locked memory of ours (allocated through
// npalloc::host_malloc, which keeps the list) go straight to hipMemcpyAsync and are asynchronous.  Everything else:
//   H2D  the bytes are taken at the time of the call through rings of our own pinned slots (up to 256 KiB: one small slot; above: 8 MiB
//        slots, the host memcpy of chunk k + 1 overlapping the DMA of chunk k) -- the source may be freed as soon as the call returns;
//   D2H  small: the runtime's staging (the bytes arrive with the next synchronisation of the stream, as before); above 256 KiB: through the
//        ring, complete when the call returns.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#include "np_devalloc.h"

namespace npcopy {

constexpr size_t kDirectMax = 256 << 10;      // pageable copies up to this size are left to the runtime's own staging
constexpr size_t kSlotBytes = 8 << 20;
constexpr int kSlots = 16;      // (round 6: 6 -> 16: the from-files ingest reads half of them full in parallel while the other half is on its way to the device;
                                // 32 x 4 MiB was measured too: no faster)

struct Slot { void* p = nullptr; hipEvent_t ev = nullptr; bool pending = false; };

class Ring {      // one per device, library and slot size, made at the first pageable copy that needs it
public:
    Ring(size_t slot_bytes, int max_slots) : slot_bytes_(slot_bytes), max_slots_(max_slots) {}
    size_t slot_bytes() const { return slot_bytes_; }
    // A slot to fill.  Blocks only while all kSlots exist and other threads hold them.  (Round 6: a new slot is allocated OUTSIDE the lock --
    // hipHostMalloc can take milliseconds and used to stall every release() behind it; and a failed allocation only fails the call when
    // there is no slot at all to wait for.)
    bool acquire(Slot* out) { return take(out, true); }
    // a slot only if one can be had without waiting for another thread to give one back (a thread that already holds a slot must not
    // block here: with as many such threads as slots nobody could ever release)
    bool try_acquire(Slot* out) { return take(out, false); }
    void release(const Slot& s) {
        { std::lock_guard<std::mutex> g(mu_); free_.push_back(s); }
        cv_.notify_one();
    }
    // Before a stream is destroyed: every event this ring recorded on ANY stream is waited for now, so that no slot is left with an event
    // whose stream no longer exists (slots in other threads' hands are theirs to settle: they synchronise their own stream).
    void settle() {
        std::deque<Slot> mine;
        { std::lock_guard<std::mutex> g(mu_); mine.swap(free_); }
        for (Slot& s : mine) if (s.pending) { (void)hipEventSynchronize(s.ev); s.pending = false; }
        { std::lock_guard<std::mutex> g(mu_); for (Slot& s : mine) free_.push_back(s); }
        cv_.notify_all();
    }
private:
    bool take(Slot* out, bool may_wait) {
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            if (!free_.empty() && !free_.front().pending) break;                 // an idle slot: take it
            if (made_ + making_ < max_slots_ && !alloc_failed_) {                     // room for one more: make it, unlocked
                ++making_;
                g.unlock();
                Slot s;
                bool ok = npalloc::host_malloc(&s.p, slot_bytes_, hipHostMallocPortable) == hipSuccess;
                if (ok && hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) { (void)npalloc::host_free(s.p); ok = false; }
                g.lock();
                --making_;
                if (ok) { ++made_; *out = s; return true; }
                alloc_failed_ = true;                                             // (no second attempt while slots exist: wait for one instead)
                cv_.notify_all();
                continue;
            }
            if (!free_.empty()) break;                                            // a slot whose last copy is still in flight: its event is waited for below
            if (made_ + making_ == 0) { alloc_failed_ = false; return false; }    // no pinned memory to be had at all: the caller reports it
            if (!may_wait) return false;
            cv_.wait(g, [&] { return !free_.empty() || made_ + making_ == 0; });
        }
        *out = free_.front();
        free_.pop_front();
        g.unlock();
        if (out->pending) { (void)hipEventSynchronize(out->ev); out->pending = false; }
        return true;
    }
    const size_t slot_bytes_;
    const int max_slots_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Slot> free_;
    int made_ = 0, making_ = 0;
    bool alloc_failed_ = false;
};

constexpr size_t kSmallSlotBytes = kDirectMax;      // small pageable H2D copies (round 6): their own ring of small slots
constexpr int kSmallSlots = 32;
struct Rings { std::mutex mu; std::map<int, Ring*> of; };      // (never destroyed: the runtime may already be gone when static destructors run)
inline Rings& rings() { static Rings* r = new Rings(); return *r; }
inline void settle_all(hipStream_t) {
    std::vector<Ring*> all;
    { Rings& R = rings(); std::lock_guard<std::mutex> g(R.mu); for (auto& kv : R.of) all.push_back(kv.second); }
    for (Ring* r : all) r->settle();
}
inline Ring& ring(bool small = false) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    Rings& R = rings();
    std::lock_guard<std::mutex> g(R.mu);
    Ring*& r = R.of[2 * dev + (small ? 1 : 0)];
    if (!r) { r = small ? new Ring(kSmallSlotBytes, kSmallSlots) : new Ring(kSlotBytes, kSlots); npalloc::stream_destroy_hook() = settle_all; }
    return *r;
}

inline hipError_t h2d(void* dst, const void* src, size_t bytes, hipStream_t q) {
    if (!bytes) return hipSuccess;
    if (npalloc::is_pinned(src, bytes)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, q);
    if (bytes <= kDirectMax) {
        // Small and pageable.  ROCm 7.2 stages such a copy through a pinned buffer of its own while the call runs -- but where "small" ends is
        // the runtime's business (and an environment variable's), and callers free the source as soon as this returns: the bytes go through a
        // slot of ours, taken now.  Only without any pinned memory to be had is the copy left to the runtime, and waited for.
        Ring& S = ring(true);
        Slot s;
        if (!S.acquire(&s)) {
            const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, q);
            return e != hipSuccess ? e : hipStreamSynchronize(q);
        }
        memcpy(s.p, src, bytes);
        hipError_t e = hipMemcpyAsync(dst, s.p, bytes, hipMemcpyHostToDevice, q);
        if (e == hipSuccess) e = hipEventRecord(s.ev, q);
        s.pending = e == hipSuccess;
        if (e != hipSuccess) (void)hipStreamSynchronize(q);
        S.release(s);
        return e;
    }
    Ring& R = ring();
    for (size_t off = 0; off < bytes;) {
        const size_t n = bytes - off < kSlotBytes ? bytes - off : kSlotBytes;
        Slot s;
        if (!R.acquire(&s)) return hipErrorOutOfMemory;
        memcpy(s.p, static_cast<const char*>(src) + off, n);
        hipError_t e = hipMemcpyAsync(static_cast<char*>(dst) + off, s.p, n, hipMemcpyHostToDevice, q);
        if (e == hipSuccess) e = hipEventRecord(s.ev, q);
        s.pending = e == hipSuccess;
        if (e != hipSuccess) (void)hipStreamSynchronize(q);
        R.release(s);
        if (e != hipSuccess) return e;
        off += n;
    }
    return hipSuccess;
}

// (large pageable destination: complete on return; two slots in flight so that the memcpy out of one overlaps the DMA into the other)
inline hipError_t d2h(void* dst, const void* src, size_t bytes, hipStream_t q) {
    if (!bytes) return hipSuccess;
    if (bytes <= kDirectMax || npalloc::is_pinned(dst, bytes)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, q);
    Ring& R = ring();
    Slot cur, nxt;
    size_t cur_off = 0, cur_n = 0;
    bool have_cur = false;
    hipError_t err = hipSuccess;
    for (size_t off = 0; off < bytes || have_cur;) {
        bool have_nxt = false;
        size_t nxt_off = 0, nxt_n = 0;
        if (off < bytes && err == hipSuccess) {
            nxt_n = bytes - off < kSlotBytes ? bytes - off : kSlotBytes;
            nxt_off = off;
            // the first slot may be waited for; a second one (to overlap the memcpy out of the first with the next DMA) only if it is free
            const bool got = have_cur ? R.try_acquire(&nxt) : R.acquire(&nxt);
            if (!got) {
                if (!have_cur) err = hipErrorOutOfMemory;
            } else {
                hipError_t e = hipMemcpyAsync(nxt.p, static_cast<const char*>(src) + off, nxt_n, hipMemcpyDeviceToHost, q);
                if (e == hipSuccess) e = hipEventRecord(nxt.ev, q);
                if (e != hipSuccess) { (void)hipStreamSynchronize(q); R.release(nxt); err = e; }
                else { have_nxt = true; off += nxt_n; }
            }
        }
        if (have_cur) {
            const hipError_t e = hipEventSynchronize(cur.ev);
            if (e == hipSuccess) memcpy(static_cast<char*>(dst) + cur_off, cur.p, cur_n);
            else if (err == hipSuccess) err = e;
            cur.pending = false;
            R.release(cur);
            have_cur = false;
        }
        if (have_nxt) { cur = nxt; cur_off = nxt_off; cur_n = nxt_n; have_cur = true; }
        else if (err != hipSuccess) break;
    }
    return err;
}

// the synchronous forms: on the null stream, complete on return.  (Not hipMemcpy: the synchronous API waits through
// Command::awaitCompletion, the wake-up by the runtime's handler thread that hipFree was seen to lose -- np_devalloc.h, quiesce;
// hipMemcpyAsync + hipStreamSynchronize waits on the hardware signal.)
inline hipError_t h2d_sync(void* dst, const void* src, size_t bytes) {
    if (!bytes) return hipSuccess;
    const hipError_t e = h2d(dst, src, bytes, nullptr);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}
inline hipError_t d2h_sync(void* dst, const void* src, size_t bytes) {
    if (!bytes) return hipSuccess;
    const hipError_t e = d2h(dst, src, bytes, nullptr);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

}  // namespace npcopy
