// kmx_internal.h -- host-side state shared by the translation units behind include/kmx.h
// (kmx_api.hip: entry points; kmx_comm.hip: the RCCL communicator).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/kmx.h"

struct kmx_ctx {
    int device;
    hipStream_t stream;
    bool owns_stream;
    int n_cu;
    unsigned long long* h_pinned;   // one pinned host word: the sticky flag read back by kmx_ctx_synchronize on the context's own stream
    unsigned long long* d_scratch;  // 8 KiB: [0] first_bad, [2..3] fastx totals, [4] length range, [16..] the queue block (kmx_device.h: ticket heads, marks, partial summaries)
    unsigned long long* h_pub;      // eight pinned host words the bit-sliced scan's last block can leave {token, marked reads, summary} in (kmx_canonical_reduce_host)
    uint32_t pub_token;             // the token of the last such launch (24 bits)
    bool queue_clean;               // the heads and [512] of the queue block are zero: only self-closing launches (kmx_device.h) ran since the last clear
    void* d_big;                    // grow-only work buffer of the partitioned histogram (bucket-id streams)
    size_t big_bytes;
    size_t big_limit;               // kmx_ctx_set_work_buffer_limit: 0 = automatic (an eighth of the device memory, at most half of what is free)
    unsigned long long big_allocs;  // how often the work buffer was (re)allocated (kmx_ctx_work_buffer_info)
    unsigned long long dirty_desc;  // address of the dirty-tile flags as last written behind the queue heads
    uint8_t* d_flags;               // one byte per tile, all zero between calls
    size_t flags_bytes;
    // kmx_fastx_parse: what the last counting pass was run on -- a second call on the same image with KMX_FASTX_SAME_TEXT
    // reuses its chunk prefixes (they live in d_big) instead of summarising the text again
    const uint8_t* fx_text;
    uint64_t fx_bytes;
    uint32_t fx_fasta;
    bool fx_valid;
    unsigned long long fx_totals[2];
    char last_error[256];
};

namespace kmx {
int fail_hip(kmx_ctx* ctx, hipError_t e, const char* where);

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
}  // namespace kmx

#define KMX_HIP(ctx, expr)                                            \
    do {                                                              \
        hipError_t e__ = (expr);                                      \
        if (e__ != hipSuccess) return kmx::fail_hip(ctx, e__, #expr); \
    } while (0)
