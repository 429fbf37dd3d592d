// Definitions shared by the translation units behind include/nextpolish1.h, Part 2 (not part of the ABI).
#pragma once
#include <string>

#include "np_stream.h"

struct np1_stream {
    np::ReadStream s;
    bool pinned = false;   // the arrays are registered with the HIP runtime (np1_stream_pin): H2D copies from them are asynchronous
};

void np1_set_error(const std::string& e);   // np_host_abi.cpp
void np1_stream_unpin(np1_stream* st);      // np1_device.hip (no-op when the stream was never pinned)

// np1_device.hip: everything a pass allocates besides the uploaded inputs changes places between the two batches
struct np1_batch;
void np1_batch_swap_work(np1_batch* a, np1_batch* b);
