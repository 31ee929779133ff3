// kmx_comm.hip -- the path's only exchange steps, on RCCL (xGMI inside a node), behind the C ABI (include/kmx.h).
//
// The reference has no distributed code.  Reads shard embarrassingly -- k-mers never span reads
// (src/naive_impl/canonical_kmer_iterator.rs:72-83) -- so the scans run without any collective.  What crosses GPUs:
//   * the optional bucket histogram: ONE ncclAllReduce(ncclUint64, ncclSum) of 2^b counters (8 MiB at b = 20:
//     latency-bound; 512 MiB at b = 26: bound by the 7 x ~153 GB/s xGMI links of each GPU);
//   * the 32-byte kmx_summary of every rank: an ncclAllGather of 4 words and a fold by one wave (RCCL has no xor
//     reduction, and wrapping u64 sums are what the fold does anyway).
// One communicator per context (one process or thread per GPU); everything is enqueued on the context's stream.
//
// RCCL is resolved at the first kmx_comm_* call (dlopen + dlsym), not at link time: a single-GPU C, C++ or Rust host that
// never exchanges anything loads libkmx.so without librccl on its path, and inside a torch process the soname resolves to
// the librccl torch has already mapped -- one RCCL per process, not two.  <rccl/rccl.h> is used for its types only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "kmx_internal.h"

static_assert(KMX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "kmx_comm id = ncclUniqueId");

struct kmx_comm {
    kmx_ctx* ctx;
    ncclComm_t comm;
    int n_ranks, rank;
    unsigned long long* d_gather;   // n_ranks x 4 words: the summaries of all ranks
};

namespace kmx {

// summaries of all ranks -> combined summary (one wave; rank r's summary at all[4r .. 4r+3])
__global__ void __launch_bounds__(64) fold_summaries_kernel(const unsigned long long* __restrict__ all, int n_ranks,
                                                            unsigned long long* __restrict__ out) {
    const unsigned f = threadIdx.x;   // field: 0 n_valid (+), 1 sum_canon (+), 2 xor_hash (^), 3 sum_fw (+)
    if (f >= 4u) return;
    unsigned long long acc = 0;
    for (int r = 0; r < n_ranks; ++r) {
        const unsigned long long v = all[4 * r + f];
        acc = f == 2u ? (acc ^ v) : (acc + v);   // wrapping, like the per-shard sums
    }
    out[f] = acc;
}

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char err[200] = {0};
};

static RcclApi load_rccl() {
    RcclApi a;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        std::snprintf(a.err, sizeof a.err, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
        return a;
    }
    bool all = true;
    auto sym = [&](const char* n) -> void* {
        void* p = dlsym(h, n);
        if (!p) {
            all = false;
            std::snprintf(a.err, sizeof a.err, "RCCL symbol %s missing", n);
        }
        return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.ok = all;
    return a;
}

static const RcclApi& rccl() {
    static const RcclApi api = load_rccl();   // (thread-safe: one load per process)
    return api;
}

static int fail_nccl(kmx_ctx* ctx, ncclResult_t r, const char* where) {
    if (ctx) std::snprintf(ctx->last_error, sizeof ctx->last_error, "%s: RCCL: %s", where, rccl().ok ? rccl().GetErrorString(r) : rccl().err);
    return KMX_E_HIP;
}

static int fail_no_rccl(kmx_ctx* ctx, const char* where) {
    if (ctx) std::snprintf(ctx->last_error, sizeof ctx->last_error, "%s: %s", where, rccl().err);
    return KMX_E_HIP;
}

}  // namespace kmx

#define KMX_NCCL(ctx, expr)                                         \
    do {                                                            \
        ncclResult_t r__ = (expr);                                  \
        if (r__ != ncclSuccess) return kmx::fail_nccl(ctx, r__, #expr); \
    } while (0)

extern "C" {

int kmx_comm_get_unique_id(uint8_t* h_id) {
    if (!h_id) return KMX_E_ARG;
    if (!kmx::rccl().ok) return KMX_E_HIP;   // (no context to carry the text: kmx_comm_create on the same host reports it)
    ncclUniqueId id;
    if (kmx::rccl().GetUniqueId(&id) != ncclSuccess) return KMX_E_HIP;
    std::memcpy(h_id, id.internal, KMX_COMM_ID_BYTES);
    return KMX_OK;
}

int kmx_comm_create(kmx_ctx* ctx, const uint8_t* h_id, int n_ranks, int rank, kmx_comm** out) {
    if (!ctx || !h_id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return KMX_E_ARG;
    *out = nullptr;
    if (!kmx::rccl().ok) return kmx::fail_no_rccl(ctx, "kmx_comm_create");
    const kmx::RcclApi& R = kmx::rccl();
    kmx::DeviceGuard g(ctx->device);
    kmx_comm* c = new (std::nothrow) kmx_comm();
    if (!c) return KMX_E_NOMEM;
    c->ctx = ctx;
    c->comm = nullptr;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->d_gather = nullptr;
    ncclUniqueId id;
    std::memcpy(id.internal, h_id, KMX_COMM_ID_BYTES);
    ncclResult_t r = R.CommInitRank(&c->comm, n_ranks, id, rank);   // collective: returns when every rank has joined
    if (r != ncclSuccess) {
        delete c;
        return kmx::fail_nccl(ctx, r, "ncclCommInitRank");
    }
    int count = 0;
    if (R.CommCount(c->comm, &count) != ncclSuccess || count != n_ranks) {
        std::snprintf(ctx->last_error, sizeof ctx->last_error, "kmx_comm_create: communicator has %d ranks, expected %d", count, n_ranks);
        (void)R.CommDestroy(c->comm);
        delete c;
        return KMX_E_HIP;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->d_gather), (size_t)n_ranks * 32u);
    if (e != hipSuccess) {
        (void)R.CommDestroy(c->comm);
        delete c;
        return kmx::fail_hip(ctx, e, "kmx_comm_create");
    }
    *out = c;
    return KMX_OK;
}

void kmx_comm_destroy(kmx_comm* comm) {
    if (!comm) return;
    kmx::DeviceGuard g(comm->ctx->device);
    (void)hipStreamSynchronize(comm->ctx->stream);
    if (comm->comm) (void)kmx::rccl().CommDestroy(comm->comm);
    if (comm->d_gather) (void)hipFree(comm->d_gather);
    delete comm;
}

int kmx_comm_size(const kmx_comm* comm) { return comm ? comm->n_ranks : -1; }
int kmx_comm_rank(const kmx_comm* comm) { return comm ? comm->rank : -1; }

int kmx_histogram_allreduce(kmx_comm* comm, uint64_t* d_counts, uint64_t n_counts) {
    if (!comm || (n_counts && !d_counts)) return KMX_E_ARG;
    if (n_counts == 0) return KMX_OK;
    kmx_ctx* ctx = comm->ctx;
    kmx::DeviceGuard g(ctx->device);
    KMX_NCCL(ctx, kmx::rccl().AllReduce(d_counts, d_counts, (size_t)n_counts, ncclUint64, ncclSum, comm->comm, ctx->stream));
    return KMX_OK;
}

int kmx_summary_allreduce(kmx_comm* comm, kmx_summary* d_summary) {
    if (!comm || !d_summary) return KMX_E_ARG;
    kmx_ctx* ctx = comm->ctx;
    kmx::DeviceGuard g(ctx->device);
    static_assert(sizeof(kmx_summary) == 32, "4 words");
    KMX_NCCL(ctx, kmx::rccl().AllGather(d_summary, comm->d_gather, 4, ncclUint64, comm->comm, ctx->stream));
    hipLaunchKernelGGL(kmx::fold_summaries_kernel, dim3(1), dim3(64), 0, ctx->stream, comm->d_gather, comm->n_ranks,
                       reinterpret_cast<unsigned long long*>(d_summary));
    KMX_HIP(ctx, hipGetLastError());
    return KMX_OK;
}

}  // extern "C"
