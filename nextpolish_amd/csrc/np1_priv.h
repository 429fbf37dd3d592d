// Definitions shared by the translation units behind include/nextpolish1.h, Part 2 (not part of the ABI).
#pragma once
#include <string>

#include "np_stream.h"

struct np1_stream {
    np::ReadStream s;
    bool pinned = false;   // the arrays are registered with the HIP runtime (np1_stream_pin): H2D copies from them are asynchronous
    // facts about the record arrays an upload needs, found once (np1_device.hip:stream_facts): the longest record, and whether the
    // per-record arrays a device can rebuild itself really are what it would rebuild (pool offsets = running sums, contig = the
    // record's place in read_begin) -- then 20 of the 32 fixed bytes per record need not cross PCIe
    int facts = 0;         // 0 not looked at yet, 1 dense, 2 not dense
    uint32_t max_lq = 0;
    // CIGAR operation counts in the 16 bits the BAM record gives them, for the upload (the device widens them): only a CG-tag CIGAR
    // needs more, and a stream that holds one uploads its 32-bit counts as they are.  Empty: not made / does not apply.
    std::vector<uint16_t> ncig16;
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
