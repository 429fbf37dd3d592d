// Definitions shared by the translation units behind include/nextpolish1.h, Part 2 (not part of the ABI).
#pragma once
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <atomic>
#include <mutex>

#include "np_stream.h"

struct np1_stream {
    np::ReadStream s;
    // np1_stream_pin: every array an upload moves (above the size the runtime stages itself) has a copy in ONE page-locked arena from
    // hipHostMalloc, and uploads read from there -- asynchronous at full PCIe rate.  (Rounds 2-4 registered the std::vector storage
    // itself with hipHostRegister; round 5 found the GPU faulting on heap addresses and took the GPU off the heap: DESIGN.md section 12.)
    std::atomic<bool> pinned{false};      // set (release) once the arena and its map are complete, read (acquire) by every upload: ADVICE r5
    std::mutex pin_mu;                    // one np1_stream_pin / np1_stream_unpin at a time
    void* arena = nullptr;
    size_t arena_bytes = 0;
    std::vector<std::pair<const void*, const void*>> arena_map;   // (array, its copy in the arena), sorted by array address
    const void* up(const void* p) const {      // where an upload of array p reads from
        if (!pinned.load(std::memory_order_acquire) || arena_map.empty()) return p;
        size_t lo = 0, hi = arena_map.size();
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (arena_map[mid].first < p) lo = mid + 1; else hi = mid; }
        return (lo < arena_map.size() && arena_map[lo].first == p) ? arena_map[lo].second : p;
    }
    // facts about the record arrays an upload needs, found once (np1_device.hip:stream_facts): the longest record, and whether the
    // per-record arrays a device can rebuild itself really are what it would rebuild (pool offsets = running sums, contig = the
    // record's place in read_begin) -- then 20 of the 32 fixed bytes per record need not cross PCIe
    int facts = 0;         // 0 not looked at yet, 1 dense, 2 not dense
    std::mutex facts_mu;   // the facts and the upload forms are built lazily, by whichever lane or thread asks first (np1_device.hip:stream_facts)
    uint32_t max_lq = 0;
    // CIGAR operation counts in the 16 bits the BAM record gives them, for the upload (the device widens them): only a CG-tag CIGAR
    // needs more, and a stream that holds one uploads its 32-bit counts as they are.  Empty: not made / does not apply.
    std::vector<uint16_t> ncig16;
    // The packed bases in the form they cross PCIe in: 2 bits per base (A C G T; the 4-bit code is 1 << that), i.e. half the bytes of
    // the largest array of a short-read batch -- the streamed pass is bound by the H2D copies (bench.py: h2d_gb_per_s).  Bytes of
    // `seq` holding anything else (N, ambiguity codes, the pad nibble of an odd-length record) travel as (byte index, byte) pairs and
    // are patched in after the device has expanded the rest.  seq2[j] = codes of seq[2j] (high 4 bits) and seq[2j + 1] (low 4 bits).
    // Empty: not made, or more than 1 byte in 64 would be an exception (then `seq` is uploaded as it is).
    std::vector<uint8_t> seq2;
    std::vector<uint64_t> esc_at;
    std::vector<uint8_t> esc_val;
    // The per-record fields in the form they cross PCIe in (dense streams of at least a few records; np1_device.hip:stream_facts
    // builds it, np1_kernels.hip:launch_expand_records undoes it).  Short-read records are alike: one match operation as long as the
    // read, the read as long as all the others, a position a few bases behind the previous record's.  A record of that kind ("plain")
    // sends one bit instead of n_cigar + l_qseq + its operation; every record sends its position as a one-byte step from the record
    // before it.  What does not fit -- other CIGARs and lengths, steps of 255 or more, the first record of a contig -- travels in
    // full, in record order, in the x_ arrays.
    struct Compact {
        bool on = false;
        uint32_t common_lq = 0;
        std::vector<uint32_t> plain;      // bit r: record r is plain
        std::vector<int32_t> x_lq;        // the records that are not, in record order: l_qseq, n_cigar, operations
        std::vector<uint32_t> x_ncig;
        std::vector<uint32_t> x_cigar;
        std::vector<uint8_t> dpos;        // pos[r] - pos[r - 1]; 255: the next entry of x_pos
        std::vector<int32_t> x_pos;
        uint64_t n_ops = 0;               // operations of all records (size of the rebuilt pool)
    } compact;
    // the draft as it crosses PCIe: 4 bits per character (2-bit base, lower-case flag); every other character (N, ambiguity letters,
    // anything else) is an exception (index, character) patched in on the device.  Empty: not made / too many exceptions.
    std::vector<uint8_t> draft4;
    std::vector<uint64_t> desc_at;
    std::vector<uint8_t> desc_val;
    uint64_t upload_bytes = 0;     // what a reload of this stream moves over PCIe (np1_stream_upload_bytes)
};

void np1_set_error(const std::string& e);   // np_host_abi.cpp
void np1_stream_unpin(np1_stream* st);      // np1_device.hip (no-op when the stream was never pinned)

// np1_device.hip: everything a pass allocates besides the uploaded inputs changes places between the two batches
struct np1_batch;
void np1_batch_swap_work(np1_batch* a, np1_batch* b);
// results of the last run: total bytes, and the D2H copy straight into a caller's (page-locked) buffer
size_t np1_batch_results_total(np1_batch* b);
int np1_batch_results_fetch_to(np1_batch* b, char* dst, size_t cap);
void* np1_host_alloc_pinned(size_t bytes);
void np1_host_free_pinned(void* p);
