// kmx_api.hip -- the extern "C" boundary declared in include/kmx.h: argument checking,
// kernel selection and HIP error mapping.  No CPU compute path exists here: every entry
// point launches a gfx950 kernel or returns an error.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "kmx_internal.h"

namespace kmx {
typedef uint32_t u32;
typedef uint64_t u64;
// kmx_scan.hip
hipError_t launch_scan_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw,
                               kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                               const u64* offsets);
hipError_t launch_hist_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 hasher, u32 hk, u32 log2_buckets,
                               u64* counts, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                               void* (*get_scratch)(void*, size_t), void* user, size_t scratch_budget, const u64* offsets);
hipError_t launch_windows_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u64* fw, u64* rc, u64* canon,
                                  uint8_t* flags, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled);
hipError_t launch_windows_ragged(const uint8_t* bases, const u64* offsets, const u64* win_offsets, u64 n_reads, u32 L, u32 k,
                                 u64* fw, u64* rc, u64* canon, uint8_t* flags, unsigned long long* queue, int n_cu,
                                 hipStream_t stream, bool* handled, const u64* ends = nullptr /* the reads' ends: nullptr = offsets + 1 */);
size_t uniform_segments_scratch_bytes(u64 n_seg);
hipError_t launch_uniform_segments_plan(u64 n_reads, u32 L, u32 k, u32 T, void* scratch, const u64** starts, const u64** ends, const u64** wins, u64* n_seg_out,
                                        hipStream_t stream);
// kmx_bitslice.hip
hipError_t launch_scan_bitsliced(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, u32 mode /* KMX_BS_* (kmx_device.h): bit 0 = sum_fw */,
                                 kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled);
hipError_t launch_sweep_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw, kmx_summary* out,
                                unsigned long long* queue, int n_cu, hipStream_t stream);
hipError_t launch_scan_bitsliced2(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, kmx_summary2* out,
                                  unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled);
u64 bitsliced_segments_per_read(u32 L, u32 k);
hipError_t launch_scan_bitsliced2_ragged(const uint8_t* bases, const u64* offsets, u64 n_reads, u32 L_hint, u32 k, bool want_hash,
                                         kmx_summary2* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                                         const u64* ends = nullptr);
hipError_t launch_scan_bitsliced_ragged(const uint8_t* bases, const u64* offsets, u64 n_reads, u32 L_hint, u32 k, bool want_hash,
                                        kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                                        bool want_sumfw = false, const u64* ends = nullptr /* the reads' ends: nullptr = offsets + 1 */);
// kmx_segments.hip
size_t segments_scratch_bytes(u64 n_reads, u64 seg_capacity, bool with_wins = false);
u64 segments_capacity(u64 n_reads, u64 total_bases, u32 t_max);
hipError_t launch_segments_build(const u64* offsets, u64 n_reads, u32 k, u32 t_max, u64 seg_capacity, void* scratch,
                                 const u64** starts_out, const u64** ends_out, const u64** total_out, unsigned long long* too_long,
                                 hipStream_t stream, const u64* win_offsets = nullptr, const u64** wins_out = nullptr);
hipError_t launch_scan_bitsliced_packed(const uint64_t* words, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw,
                                        kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled);
// kmx_fastx.hip
size_t fastx_scratch_bytes(u64 n_bytes);
hipError_t launch_fastx_count(const uint8_t* text, u64 n, bool fasta, void* scratch, unsigned long long* totals, hipStream_t st);
hipError_t launch_fastx_emit(const uint8_t* text, u64 n, bool fasta, const void* scratch, const unsigned long long* totals,
                             uint8_t* bases, u64* offsets, hipStream_t st);
// kmx_seqvec.hip
hipError_t launch_seqvec_push(u64* words, u64 first, const uint8_t* bytes, u64 n, unsigned long long* first_bad, int n_cu,
                              hipStream_t st);
hipError_t launch_seqvec_to_bytes(const u64* words, u64 n, uint8_t* out, int n_cu, hipStream_t st);
hipError_t launch_seqvec_get_kmers(const u64* words, u64 n_bases, const u64* pos, u64 n, u32 k, u64* out,
                                   unsigned long long* first_bad, int n_cu, hipStream_t st);
hipError_t launch_seqvec_iter_kmers(const u64* words, u64 n_bases, u64 start, u64 count, u32 k, u64* out, int n_cu,
                                    hipStream_t st);
hipError_t launch_minimizer_words(const u64* in, u64 n, u32 k, u32 w, u32 hasher, u32 hk, u64* out_mm, u32* out_off, int n_cu,
                                  hipStream_t st);
hipError_t launch_seqvec_minimizers(const u64* words, u64 n_reads, u32 L, u32 k, u32 w, u32 hasher, u32 hk, u64* out_word,
                                    u32* out_pos, int n_cu, hipStream_t st);
hipError_t launch_reduce_packed_generic(const u64* words, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw,
                                        kmx_summary* out, int n_cu, hipStream_t st);
// kmx_generic.hip
hipError_t launch_reduce_generic(const kmx_reads* r, u32 k, u32 hasher, u32 hk, u32 want_sumfw, kmx_summary* out,
                                 int n_cu, hipStream_t st, unsigned long long* too_long);
hipError_t launch_windows_generic(const kmx_reads* r, const u64* win_off, u32 k, u64* fw, u64* rc, u64* canon,
                                  uint8_t* flags, int n_cu, hipStream_t st, unsigned long long* too_long);
hipError_t launch_histogram_generic(const kmx_reads* r, u32 k, u32 hasher, u32 hk, u32 log2_buckets, u64* counts,
                                    int n_cu, hipStream_t st, unsigned long long* too_long);
hipError_t launch_reduce2_generic(const kmx_reads* r, u32 k, u32 with_hash, kmx_summary2* out, int n_cu, hipStream_t st, unsigned long long* too_long,
                                  const u32* gate);
hipError_t launch_windows2_tiled(const kmx_reads* r, u32 k, u64* fw, u64* rc, u64* canon, uint8_t* flags, int n_cu, hipStream_t st,
                                 bool* handled, unsigned long long* queue);
hipError_t launch_windows2_tiled_ragged(const kmx_reads* r, const u64* win_offsets, u32 k, u64* fw, u64* rc, u64* canon, uint8_t* flags, int n_cu,
                                        hipStream_t st, bool* handled, unsigned long long* too_long, const u64* ends, unsigned long long* queue);
hipError_t launch_windows2_generic(const kmx_reads* r, const u64* win_off, u32 k, u64* fw, u64* rc, u64* canon,
                                   uint8_t* flags, int n_cu, hipStream_t st, unsigned long long* too_long);
// kmx_elem.hip
hipError_t launch_gen_reads(u64 seed, u64 first_byte, uint8_t* out, u64 nbytes, int n_cu, hipStream_t st);
hipError_t launch_kmers_from_bytes(const uint8_t* seqs, u64 n, u32 k, u64* words, unsigned long long* first_bad, int n_cu,
                                   hipStream_t st);
hipError_t launch_revcomp_words(const u64* in, u64 n, u32 k, u64* out, int n_cu, hipStream_t st);
hipError_t launch_canonical_words(const u64* in, u64 n, u32 k, u64* canon, uint8_t* is_canon, int n_cu, hipStream_t st);
hipError_t launch_hash_words(const u64* in, u64 n, u32 hasher, u32 hk, u64* out, int n_cu, hipStream_t st);
hipError_t launch_hash_words_sip13(const u64* in, u64 n, u64 k0, u64 k1, u64* out, int n_cu, hipStream_t st);
hipError_t launch_match_words(const u64* fw, const u64* rc, const u64* other, u64 n, uint8_t* out, int n_cu, hipStream_t st);
hipError_t launch_ck_shift(bool append, u64* fw, u64* rc, const uint8_t* bases, u64 n, u32 k, uint8_t* dropped, int n_cu,
                           hipStream_t st);
hipError_t launch_encode_kmers(const uint8_t* seqs, u64 n, u32 seq_len, u32 enc, u32 B, u64* words, int n_cu, hipStream_t st);
hipError_t launch_encode_windows(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 enc, u32 B, u64* words, int n_cu,
                                 hipStream_t st);
hipError_t launch_encoding_rev_comp(const u64* in, u64 n, u32 K, u32 comp_lut, u32 B, u64* out, int n_cu, hipStream_t st);
hipError_t launch_encoding_decode(const u64* in, u64 n, u32 nuc_lut, u32 B, uint8_t* seqs, int n_cu, hipStream_t st);
hipError_t launch_sub_kmer_words(const u64* in, u64 n, u32 pos, u32 width, u64* out, int n_cu, hipStream_t st);
hipError_t launch_kmers_to_bytes(const u64* in, u64 n, u32 k, bool upper, uint8_t* out, int n_cu, hipStream_t st);
hipError_t launch_encode_kmers_bytes(const uint8_t* seqs, u64 n, u32 seq_len, u32 enc, u32 nb, uint8_t* arrays, int n_cu, hipStream_t st);
hipError_t launch_encoding_rev_comp_bytes(const uint8_t* in, u64 n, u32 K, u32 comp_lut, u32 nb, uint8_t* out, int n_cu, hipStream_t st);
hipError_t launch_encoding_decode_bytes(const uint8_t* in, u64 total_bytes, u32 nuc_lut, uint8_t* seqs, int n_cu, hipStream_t st);
hipError_t launch_minimizers_reads(const uint8_t* bases, u64 total_bytes, const u64* offsets, const u64* win_offsets, u64 n_reads, u32 L,
                                   u32 bound, u32 k, u32 w, u32 hasher, u32 hk, u64* out_word, u32* out_pos,
                                   unsigned long long* first_bad, int n_cu, hipStream_t st, bool* tiled);
hipError_t launch_calib_stream_read(const uint8_t* buf, u64 nbytes, unsigned long long* out, int n_cu, hipStream_t st);
hipError_t launch_length_range(const u64* offsets, u64 n_reads, u32* out, int n_cu, hipStream_t st);
hipError_t launch_offsets_uniform_gate(const u64* offsets, u64 n_reads, u32 bound, u32 k, u32* gate, int n_cu, hipStream_t st);
hipError_t launch_fix_hash_fold(kmx_summary* out, u32 k, u32 hasher, u32 hk, hipStream_t st);
}  // namespace kmx

using kmx::u32;
using kmx::u64;

namespace kmx {
int fail_hip(kmx_ctx* ctx, hipError_t e, const char* where) {
    if (ctx) std::snprintf(ctx->last_error, sizeof ctx->last_error, "%s: %s", where, hipGetErrorString(e));
    return KMX_E_HIP;
}
}  // namespace kmx
using kmx::DeviceGuard;
using kmx::fail_hip;

namespace {

void* big_scratch(void* user, size_t bytes);

// Zero the first `bytes` of the context's queue block (ticket heads, then the marked-reads count and the gate) ahead of a launch
// that takes tickets.  The bit-sliced scans put back what they use (kmx_device.h); every other user leaves its tickets behind.
hipError_t queue_clear(kmx_ctx* ctx, size_t bytes) {
    ctx->queue_clean = false;
    return hipMemsetAsync(ctx->d_scratch + 16, 0, bytes, ctx->stream);
}

// The bit-sliced scan blanks the reads that hold an invalid byte out of their tile and leaves their 64-bit mask (8 bytes
// per tile) for sweep_flagged_kernel (kmx_sweep.hip; kmx_bitslice_kernel.h, "reads with an invalid byte").  The masks are the context's
// own grow-only array, zeroed when it is allocated; the rolling kernel clears every mask it consumes, so it is all-zero
// again when a call ends and nothing has to be cleared per call.  Its address sits behind the 32 tile-queue heads (d_scratch[16 + 515]), rewritten only
// when it changes.  No array: 0, and such tiles take the per-lane path as a whole, as they do for k without a bit-sliced kernel.
int prepare_dirty_flags(kmx_ctx* ctx, uint64_t n_reads, uint32_t k, bool any_k = false) {
    const uint64_t n_tiles = n_reads >> 6;
    uint8_t* buf = nullptr;
    if (n_tiles && (any_k || (k >= 9 && k <= 31) || (k >= 33 && k <= 64))) {   // the k with a bit-sliced kernel (any_k: the word-domain scan's histogram sinks mark too)
        if (8u * n_tiles > ctx->flags_bytes) {   // one 64-bit read mask per tile
            if (ctx->d_flags) {
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipFree(ctx->d_flags);
                ctx->d_flags = nullptr;
                ctx->flags_bytes = 0;
            }
            const size_t want_bytes = 8u * (size_t)(n_tiles + n_tiles / 4u + 4096u);
            void* q = nullptr;
            // (cleared on the context's stream: a non-blocking stream is not ordered against the null stream hipMemset runs on)
            if (hipMalloc(&q, want_bytes) == hipSuccess && hipMemsetAsync(q, 0, want_bytes, ctx->stream) == hipSuccess) {
                ctx->d_flags = static_cast<uint8_t*>(q);
                ctx->flags_bytes = want_bytes;
            } else {
                (void)hipGetLastError();
                if (q) (void)hipFree(q);
            }
        }
        buf = 8u * n_tiles <= ctx->flags_bytes ? ctx->d_flags : nullptr;
        if (!buf) {   // one byte per 64 reads: if even that cannot be had, nothing else will work either
            std::snprintf(ctx->last_error, sizeof ctx->last_error, "kmx: no memory for %llu tile flags", (unsigned long long)n_tiles);
            return KMX_E_NOMEM;
        }
    }
    const unsigned long long want = (unsigned long long)reinterpret_cast<uintptr_t>(buf);
    if (want != ctx->dirty_desc) {
        ctx->dirty_desc = want;
        hipError_t e = hipMemcpyAsync(ctx->d_scratch + 16 + 515, &ctx->dirty_desc, 8, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail_hip(ctx, e, "dirty-tile flags");
    }
    return KMX_OK;
}

// work buffer of the partitioned histogram: grown on demand (hipFree/hipMalloc synchronise, so only when it must grow),
// kept until the context is destroyed; nullptr => the caller falls back to the global-atomic kernel
void* big_scratch(void* user, size_t bytes) {
    kmx_ctx* ctx = static_cast<kmx_ctx*>(user);
    if (bytes <= ctx->big_bytes) return ctx->d_big;
    ctx->fx_valid = false;   // (the buffer moves: the fastx chunk prefixes in it are gone)
    if (ctx->d_big) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->d_big);
        ctx->d_big = nullptr;
        ctx->big_bytes = 0;
    }
    void* q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    ctx->big_allocs += 1;
    ctx->d_big = q;
    ctx->big_bytes = bytes;
    return q;
}

// The segment arrays of the long-read paths (16-24 bytes per segment): inside the cap of kmx_ctx_set_work_buffer_limit, or not at
// all -- the call then takes the per-read kernels, as it does when the allocation fails.
static void* capped_scratch(kmx_ctx* ctx, size_t bytes) {
    if (ctx->big_limit != 0 && bytes > ctx->big_limit) return nullptr;
    return big_scratch(ctx, bytes);
}

// an eighth of the device memory, at least 8 GiB (kmx_ctx_set_work_buffer_limit overrides), and at most half of what is free: fewer,
// larger chunks of reads per call (configs[4], 1.25e8 reads: 6 chunks at 8 GiB 19.3 ms, 2 at 36 GiB 18.6 ms)
// (`held`: the work buffer the context already owns -- it is not part of "free" any more, but it IS available: without adding
// it back a second call under memory pressure got a smaller budget than the buffer it holds, cut the reads into more chunks
// than the first call had, and timing depended on call order)
size_t hist_scratch_budget(size_t held, size_t limit) {
    if (limit) return limit;
    size_t free_b = 0, total_b = 0;
    const bool have = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    size_t budget = (size_t)8 << 30;
    if (have && total_b / 8u > budget) budget = total_b / 8u;
    if (have && budget > (free_b + held) / 2) budget = (free_b + held) / 2;
    if (budget < held) budget = held;
    return budget;
}

bool reads_ok(const kmx_reads* r) {
    if (!r) return false;
    if (r->n_reads && !r->d_bases && (r->d_offsets || r->read_len)) return false;
    return true;
}

// ---- host-side restatement of the per-encoding tables of src/encoding/naive.rs ----
// rev_encoding (naive.rs:28-39)
u32 rev_encoding(u32 enc) {
    u32 rev = 0;
    rev ^= 0u << (6 - ((enc >> 6) * 2));
    rev ^= 1u << (6 - (((enc >> 4) & 3) * 2));
    rev ^= 2u << (6 - (((enc >> 2) & 3) * 2));
    rev ^= 3u << (6 - ((enc & 3) * 2));
    return rev & 0xFFu;
}
// complement (naive.rs:98-109) for the four codes, packed 2 bits each
u32 comp_lut_for(u32 enc) {
    const u32 rev = rev_encoding(enc);
    u32 lut = 0;
    for (u32 bits = 0; bits < 4; ++bits) {
        const u32 internal = (rev >> (6 - bits * 2)) & 3;
        const u32 comp_internal = (internal ^ 2u) & 3;
        lut |= ((enc >> (6 - comp_internal * 2)) & 3) << (2 * bits);
    }
    return lut;
}
// bits2nuc (naive.rs:88-95, INTERNAL2NUC :19) for the four codes, one letter per byte
u32 nuc_lut_for(u32 enc) {
    static const unsigned char internal2nuc[4] = {'A', 'C', 'T', 'G'};
    const u32 rev = rev_encoding(enc);
    u32 lut = 0;
    for (u32 bits = 0; bits < 4; ++bits) lut |= (u32)internal2nuc[(rev >> (6 - bits * 2)) & 3] << (8 * bits);
    return lut;
}
// the 24 discriminants of `enum Naive` (naive.rs:48-74) are exactly the bytes whose four 2-bit
// fields are a permutation of {0,1,2,3}
bool enc_ok(u32 enc) {
    if (enc > 0xFFu) return false;
    u32 seen = 0;
    for (int i = 0; i < 4; ++i) seen |= 1u << ((enc >> (2 * i)) & 3);
    return seen == 0xFu;
}

}  // namespace

extern "C" {

int kmx_version(void) { return KMX_VERSION; }

const char* kmx_strerror(int status) {
    switch (status) {
    case KMX_OK: return "ok";
    case KMX_E_ARG: return "invalid argument";
    case KMX_E_K_RANGE: return "k outside the supported range for this call";
    case KMX_E_HIP: return "HIP runtime error (see kmx_last_error)";
    case KMX_E_INVALID_BASE: return "byte is not one of ACGTacgt (reference: encode_binary panics)";
    case KMX_E_TOO_LONG: return "sequence longer than the k-mer storage (reference panics)";
    case KMX_E_NOMEM: return "out of memory";
    default: return "unknown kmx status";
    }
}

static int ctx_create_common(int device, hipStream_t stream, bool owns, kmx_ctx** out) {
    if (!out) return KMX_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return KMX_E_HIP;  // fail loudly: no GPU, no kmx
    if (device < 0 || device >= count) return KMX_E_ARG;
    kmx_ctx* c = new (std::nothrow) kmx_ctx();
    if (!c) return KMX_E_NOMEM;
    c->device = device;
    c->stream = stream;
    c->owns_stream = owns;
    c->d_scratch = nullptr;
    c->h_pinned = nullptr;
    c->h_pub = nullptr;
    c->pub_token = 0;
    c->queue_clean = false;
    c->d_big = nullptr;
    c->big_bytes = 0;
    c->big_limit = 0;
    c->big_allocs = 0;
    c->dirty_desc = 0;
    c->d_flags = nullptr;
    c->flags_bytes = 0;
    c->last_error[0] = 0;
    DeviceGuard g(device);
    hipDeviceProp_t prop;
    hipError_t e = g.ok ? hipGetDeviceProperties(&prop, device) : hipErrorInvalidDevice;
    if (e == hipSuccess && owns) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_scratch), 8192);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), 64, hipHostMallocDefault);
    // (the dirty-list descriptor behind the queue heads starts as "no list"; cleared on the context's own stream so that
    // the clear is ordered before every kernel the context launches)
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->h_pub), 64, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_scratch, 0, 8192, c->stream);
    if (e == hipSuccess) {
        // the pinned words as the device addresses them, left in the queue block once (kmx_device.h, KMX_Q_HOST)
        void* dev_view = nullptr;
        e = hipHostGetDevicePointer(&dev_view, c->h_pub, 0);
        if (e == hipSuccess) {
            for (int i = 0; i < 8; ++i) c->h_pub[i] = 0;
            c->h_pinned[7] = (unsigned long long)reinterpret_cast<uintptr_t>(dev_view);
            e = hipMemcpyAsync(c->d_scratch + 16 + 517, c->h_pinned + 7, 8, hipMemcpyHostToDevice, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        if (c->h_pinned) (void)hipHostFree(c->h_pinned);
        if (c->h_pub) (void)hipHostFree(c->h_pub);
        if (owns && c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return KMX_E_HIP;
    }
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = c;
    return KMX_OK;
}

int kmx_ctx_create(int device, kmx_ctx** out) { return ctx_create_common(device, nullptr, true, out); }

int kmx_ctx_create_on_stream(int device, void* hip_stream, kmx_ctx** out) {
    return ctx_create_common(device, static_cast<hipStream_t>(hip_stream), false, out);
}

void kmx_ctx_destroy(kmx_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_pub) (void)hipHostFree(ctx->h_pub);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->d_big) (void)hipFree(ctx->d_big);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int kmx_ctx_synchronize(kmx_ctx* ctx) {
    if (!ctx) return KMX_E_ARG;
    DeviceGuard g(ctx->device);
    // the sticky flag of the scans: a ragged read of 2^31 bases or more was skipped (kmx.h "Limits").  Read back on the context's
    // own stream into pinned memory, ahead of the one wait: no blocking copy on the null stream (which would also synchronise
    // with every other blocking stream of the process)
    KMX_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long too_long = *ctx->h_pinned;
    if (too_long) {
        KMX_HIP(ctx, hipMemsetAsync(ctx->d_scratch + 8, 0, 8, ctx->stream));
        KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::snprintf(ctx->last_error, sizeof ctx->last_error,
                      "a read of 2^31 bases or more was skipped by a scan since the last kmx_ctx_synchronize (cut such records into overlapping pieces)");
        return KMX_E_ARG;
    }
    return KMX_OK;
}

int kmx_ctx_device(const kmx_ctx* ctx) { return ctx ? ctx->device : -1; }

int kmx_ctx_set_work_buffer_limit(kmx_ctx* ctx, size_t bytes) {
    if (!ctx) return KMX_E_ARG;
    ctx->big_limit = bytes;
    return KMX_OK;
}

int kmx_ctx_work_buffer_info(const kmx_ctx* ctx, size_t* bytes_held, uint64_t* n_allocations) {
    if (!ctx) return KMX_E_ARG;
    if (bytes_held) *bytes_held = ctx->big_bytes;
    if (n_allocations) *n_allocations = ctx->big_allocs;
    return KMX_OK;
}

const char* kmx_last_error(const kmx_ctx* ctx) { return ctx ? ctx->last_error : "null ctx"; }

int kmx_malloc(kmx_ctx* ctx, size_t nbytes, void** d_out) {
    if (!ctx || !d_out) return KMX_E_ARG;
    DeviceGuard g(ctx->device);
    *d_out = nullptr;
    if (nbytes == 0) return KMX_OK;
    hipError_t e = hipMalloc(d_out, nbytes);
    if (e == hipErrorOutOfMemory) return KMX_E_NOMEM;
    KMX_HIP(ctx, e);
    return KMX_OK;
}

int kmx_free(kmx_ctx* ctx, void* d_ptr) {
    if (!ctx) return KMX_E_ARG;
    if (!d_ptr) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipFree(d_ptr));
    return KMX_OK;
}

int kmx_memcpy_h2d(kmx_ctx* ctx, void* d_dst, const void* h_src, size_t nbytes) {
    if (!ctx || (nbytes && (!d_dst || !h_src))) return KMX_E_ARG;
    if (!nbytes) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMX_OK;
}

int kmx_memcpy_d2h(kmx_ctx* ctx, void* h_dst, const void* d_src, size_t nbytes) {
    if (!ctx || (nbytes && (!h_dst || !d_src))) return KMX_E_ARG;
    if (!nbytes) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMX_OK;
}

int kmx_memset(kmx_ctx* ctx, void* d_dst, int value, size_t nbytes) {
    if (!ctx || (nbytes && !d_dst)) return KMX_E_ARG;
    if (!nbytes) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipMemsetAsync(d_dst, value, nbytes, ctx->stream));
    return KMX_OK;
}

/* ------------------------------------------------------------ hot path ---- */

// Long ragged reads (round 4): the reads of a batch behind an offsets array cut into overlapping segments of at most t_max windows
// on the device (kmx_segments.hip), in the context's work buffer.  Two host round trips: the first and the last offset (the
// arrays are sized from the number of bases), then the number of segments.  0: *starts / *ends / *n_seg are set (n_seg may be 0);
// -1: no scratch (the caller falls back); > 0: a status to return.
static int long_ragged_segments(kmx_ctx* ctx, const kmx_reads* reads, uint32_t k, uint32_t t_max, const uint64_t** starts, const uint64_t** ends,
                                uint64_t* n_seg, const uint64_t* win_offsets = nullptr, const uint64_t** wins = nullptr) {
    KMX_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, reads->d_offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipMemcpyAsync(ctx->h_pinned + 1, reads->d_offsets + reads->n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t o_first = ctx->h_pinned[0], o_last = ctx->h_pinned[1];
    if (!(o_last >= o_first && o_last - o_first < (1ull << 62))) return -1;
    const uint64_t cap = kmx::segments_capacity(reads->n_reads, o_last - o_first, t_max);
    void* scratch = capped_scratch(ctx, kmx::segments_scratch_bytes(reads->n_reads, cap, win_offsets != nullptr));
    if (!scratch) return -1;
    ctx->fx_valid = false;   // (the work buffer is overwritten: the fastx chunk prefixes in it are gone)
    const uint64_t* d_total = nullptr;
    KMX_HIP(ctx, kmx::launch_segments_build(reads->d_offsets, reads->n_reads, k, t_max, cap, scratch, starts, ends, &d_total, ctx->d_scratch + 8, ctx->stream,
                                            win_offsets, wins));
    // (the second and last round trip: how many segments there are -- the bound above is up to one per read too high)
    KMX_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_seg = ctx->h_pinned[0];
    if (*n_seg > cap) return fail_hip(ctx, hipErrorUnknown, "segment count above its bound");
    if (wins != nullptr && *wins != nullptr && *n_seg != 0)     // the slot behind the last segment's windows: the batch's total
        KMX_HIP(ctx, hipMemcpyAsync(const_cast<uint64_t*>(*wins) + *n_seg, win_offsets + reads->n_reads, 8, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

// The body of kmx_canonical_reduce.  `host_mode` (kmx_canonical_reduce_host): KMX_BS_PUBLISH | KMX_BS_NO_SWEEP | token << 8 for
// the launch that can end the call by itself -- uniform reads on the bit-sliced scan, no second kernel behind it; `*published`
// says whether that launch was made.
static int reduce_impl(kmx_ctx* ctx, const kmx_reads* reads, uint32_t k, uint32_t hasher, uint32_t hasher_k, uint32_t flags,
                       kmx_summary* d_out, uint32_t host_mode, bool* published) {
    if (published) *published = false;
    if (!ctx || !reads_ok(reads) || !d_out) return KMX_E_ARG;
    if (k < 1 || k > 31) return KMX_E_K_RANGE;  // MASK_TABLE[32]==0 (kmer.rs:617) breaks the reference's own rolling at 32
    if (hasher > KMX_HASH_IDENTITY) return KMX_E_ARG;
    if (hasher == KMX_HASH_LEX && (hasher_k < 1 || hasher_k > 32)) return KMX_E_K_RANGE;
    DeviceGuard g(ctx->device);
    if (reads->n_reads == 0) {
        KMX_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(kmx_summary), ctx->stream));
        return KMX_OK;
    }
    const bool want_sumfw = (flags & KMX_REDUCE_SUM_FW) != 0;
    // The tiled kernels fold LexHasher(k); the fold under LexHasher(hasher_k != k) or the identity hasher follows from it
    // (fix_hash_fold_kernel: every hasher offered is linear over GF(2)), so no hasher sends a call to the per-lane kernel.
    const bool want_fold = hasher != KMX_HASH_NONE;
    const bool fix_fold = want_fold && !(hasher == KMX_HASH_LEX && hasher_k == k);
    // Uniform reads on the bit-sliced scan (round 6): its last block STORES the summary and puts the queue block back as it found
    // it (kmx_device.h), so neither `d_out` nor -- after a launch of that kind -- the queue block is cleared here: two fill kernels
    // and the gaps around them were 11 us of the 70 a batch of 1e5 reads took (profiles/r06_small_batches.txt).
    if (!reads->d_offsets) {
        bool handled = false;
        if (!ctx->queue_clean) KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
        // (reads longer than a frame are scanned as segments: a mask word per 64 of THOSE)
        if (int st = prepare_dirty_flags(ctx, reads->n_reads * kmx::bitsliced_segments_per_read(reads->read_len, k), k)) return st;
        const bool alone = host_mode != 0u && !fix_fold && reads->read_len <= 256u;   // (segments: the sweep's geometry is the launcher's)
        const uint32_t mode = (want_sumfw ? 1u : 0u) | 2u /* KMX_BS_STORE */ | (alone ? host_mode : 0u);
        KMX_HIP(ctx, kmx::launch_scan_bitsliced(reads->d_bases, reads->n_reads, reads->read_len, k, want_fold, mode, d_out,
                                                ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &handled));
        if (handled) {
            ctx->queue_clean = true;
            if (published) *published = alone;
            if (fix_fold) KMX_HIP(ctx, kmx::launch_fix_hash_fold(d_out, k, hasher, hasher_k, ctx->stream));
            return KMX_OK;
        }
    }
    KMX_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(kmx_summary), ctx->stream));
    {
        bool handled = false;
        KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));  // 32 tile-queue heads, 128 B apart, + the "a tile was flagged" word + the uniform/ragged gate
        // Reads behind an offsets array with a length bound L that the uniform bit-sliced kernels take: most FASTQ is
        // untrimmed -- every read exactly L bases -- and the uniform kernel is ~1.4x the ragged one.  Decided on the device:
        // a small kernel checks offsets[i] == i*L, both scans are launched behind its verdict, the one it names runs.
        // (round 5: uniform at ANY length up to the bound -- the gate leaves the length it found for the uniform scan, which is laid out
        // for the bound; no bound = the 160-base frame the ragged launcher assumes as well)
        const uint32_t Lh = reads->read_len ? reads->read_len : 160u;
        // (only where BOTH bit-sliced launchers take the call: the ragged one needs a 16-byte aligned base -- with a misaligned
        // base the uniform scan used to be enqueued behind the gate and the generic kernel then counted the batch a second time)
        if (reads->d_offsets && !want_sumfw && k >= 9 && k <= 31 && Lh >= k && Lh <= 256 && reads->read_len <= 256 &&
            (reinterpret_cast<uintptr_t>(reads->d_bases) & 15u) == 0u) {
            uint32_t* gate = reinterpret_cast<uint32_t*>(ctx->d_scratch + 16 + 513);
            KMX_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(gate), 1, 1, ctx->stream));
            KMX_HIP(ctx, kmx::launch_offsets_uniform_gate(reads->d_offsets, reads->n_reads, Lh, k, gate, ctx->n_cu, ctx->stream));
            if (int st = prepare_dirty_flags(ctx, reads->n_reads, k)) return st;
            bool h_u = false, h_r = false;
            KMX_HIP(ctx, kmx::launch_scan_bitsliced(reads->d_bases, reads->n_reads, Lh, k, want_fold, 0u, d_out, ctx->d_scratch + 16,
                                                    ctx->n_cu, ctx->stream, &h_u));
            if (h_u) {
                // (the uniform scan took tickets from the queue heads only if it ran; if it did not they are still zero)
                KMX_HIP(ctx, kmx::launch_scan_bitsliced_ragged(reads->d_bases, reads->d_offsets, reads->n_reads, Lh, k, want_fold, d_out,
                                                               ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &h_r));
                if (h_r) {
                    KMX_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(gate), 0, 2, ctx->stream));   // never left armed
                    if (fix_fold) KMX_HIP(ctx, kmx::launch_fix_hash_fold(d_out, k, hasher, hasher_k, ctx->stream));
                    return KMX_OK;
                }
                // The ragged launcher takes every aligned (k, L) the uniform one takes.  Should that ever stop being true, the uniform
                // scan is already enqueued behind the gate and nothing may run after it: fail loudly, never count twice.
                KMX_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(gate), 0, 2, ctx->stream));
                KMX_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(kmx_summary), ctx->stream));
                std::snprintf(ctx->last_error, sizeof ctx->last_error, "kmx: internal -- the ragged scan refused k=%u, L<=%u that the uniform scan accepted", k, Lh);
                return KMX_E_HIP;
            } else {
                KMX_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(gate), 0, 2, ctx->stream));
            }
        }
        if (!handled && reads->d_offsets && reads->read_len > 256 && k >= 9 && k <= 31 &&
            (reinterpret_cast<uintptr_t>(reads->d_bases) & 15u) == 0u) {
            // Long ragged reads (a length bound above the frames: PacBio / ONT reads, contigs), round 4: cut into overlapping
            // segments on the device (kmx_segments.hip) and scanned by the ragged bit-sliced kernel as reads of their own.  Two
            // host round trips: the first and the last offset (the segment arrays are sized from the number of bases), then the
            // number of segments.  No scratch -> the lane-per-read kernel below.
            const uint32_t t_max = 161u - k < 128u ? 161u - k : 128u;      // windows per segment: at most 4 per lane (the three-wave variant)
            const uint64_t *starts = nullptr, *ends = nullptr;
            uint64_t n_seg = 0;
            const int st = long_ragged_segments(ctx, reads, k, t_max, &starts, &ends, &n_seg);
            if (st > 0) return st;
            if (st == 0) {
                if (n_seg == 0) return KMX_OK;   // no read holds a window
                if (int st2 = prepare_dirty_flags(ctx, n_seg, k)) return st2;
                KMX_HIP(ctx, kmx::launch_scan_bitsliced_ragged(reads->d_bases, starts, n_seg, t_max + k - 1u, k, want_fold, d_out,
                                                               ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &handled, want_sumfw, ends));
            }
        }
        if (!handled && reads->d_offsets) {   // ragged reads on the bit-sliced kernel (read_len = optional length bound)
            if (int st = prepare_dirty_flags(ctx, reads->n_reads, k)) return st;
            KMX_HIP(ctx, kmx::launch_scan_bitsliced_ragged(reads->d_bases, reads->d_offsets, reads->n_reads, reads->read_len, k,
                                                           want_fold, d_out, ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &handled, want_sumfw));
        }
        // word-domain kernel: uniform reads of any (k, L) in its domain, and ragged reads (read_len = optional length bound)
        if (!handled)
            KMX_HIP(ctx, kmx::launch_scan_uniform(reads->d_bases, reads->n_reads, reads->read_len, k, want_fold, want_sumfw,
                                                  d_out, ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &handled, reads->d_offsets));
        if (handled) {
            if (fix_fold) KMX_HIP(ctx, kmx::launch_fix_hash_fold(d_out, k, hasher, hasher_k, ctx->stream));
            return KMX_OK;
        }
    }
    KMX_HIP(ctx, kmx::launch_reduce_generic(reads, k, hasher, hasher_k, want_sumfw ? 1u : 0u, d_out, ctx->n_cu, ctx->stream, ctx->d_scratch + 8));
    return KMX_OK;
}

int kmx_canonical_reduce(kmx_ctx* ctx, const kmx_reads* reads, uint32_t k, uint32_t hasher, uint32_t hasher_k,
                         uint32_t flags, kmx_summary* d_out) {
    return reduce_impl(ctx, reads, k, hasher, hasher_k, flags, d_out, 0u, nullptr);
}

// The summary straight into host memory, the call returning when it is there (round 6; kmx.h).  Uniform reads of up to 256
// bases on the bit-sliced scan: ONE kernel launch -- its last block leaves {token, marked reads, summary} in the context's
// pinned words, which this thread watches -- and, only if the scan marked reads with an invalid byte, the sweep and a copy
// back.  Everything else: kmx_canonical_reduce into a summary of the context's own, a copy, a wait.
int kmx_canonical_reduce_host(kmx_ctx* ctx, const kmx_reads* reads, uint32_t k, uint32_t hasher, uint32_t hasher_k,
                              uint32_t flags, kmx_summary* h_out) {
    if (!ctx || !h_out) return KMX_E_ARG;
    kmx_summary* const d_res = reinterpret_cast<kmx_summary*>(ctx->d_scratch + 16 + 528);   // (eight free words of the queue block)
    uint32_t token = (ctx->pub_token + 1u) & 0xFFFFFFu;
    if (token == 0u) token = 1u;                       // (the pinned word starts at 0)
    bool published = false;
    const int st = reduce_impl(ctx, reads, k, hasher, hasher_k, flags, d_res, 4u /* KMX_BS_PUBLISH */ | 8u /* KMX_BS_NO_SWEEP */ | (token << 8), &published);
    if (st != KMX_OK) return st;
    DeviceGuard g(ctx->device);
    if (published) {
        ctx->pub_token = token;
        volatile unsigned long long* const hp = ctx->h_pub;
        unsigned spins = 0, done_seen = 0;
        while (hp[0] != (unsigned long long)token) {
            if ((++spins & 0x3FFFu) == 0u) {           // (a launch that died never writes the token)
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q != hipErrorNotReady) {
                    if (q != hipSuccess) return fail_hip(ctx, q, "kmx_canonical_reduce_host");
                    if (++done_seen > 64u) {
                        std::snprintf(ctx->last_error, sizeof ctx->last_error, "kmx: internal -- the scan ended without publishing its summary");
                        return KMX_E_HIP;
                    }
                }
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (hp[1] == 0ull) {                           // no read was marked: what the scan left is the answer
            h_out->n_valid = hp[2]; h_out->sum_canon = hp[3]; h_out->xor_hash = hp[4]; h_out->sum_fw = hp[5];
            return KMX_OK;
        }
        KMX_HIP(ctx, kmx::launch_sweep_uniform(reads->d_bases, reads->n_reads, reads->read_len, k, hasher != KMX_HASH_NONE,
                                               (flags & KMX_REDUCE_SUM_FW) != 0, d_res, ctx->d_scratch + 16, ctx->n_cu, ctx->stream));
    }
    KMX_HIP(ctx, hipMemcpyAsync(ctx->h_pinned + 2, d_res, sizeof(kmx_summary), hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h_out->n_valid = ctx->h_pinned[2]; h_out->sum_canon = ctx->h_pinned[3]; h_out->xor_hash = ctx->h_pinned[4]; h_out->sum_fw = ctx->h_pinned[5];
    return KMX_OK;
}

int kmx_canonical_windows(kmx_ctx* ctx, const kmx_reads* reads, const uint64_t* d_win_offsets, uint32_t k,
                          uint64_t* d_fw, uint64_t* d_rc, uint64_t* d_canon, uint8_t* d_flags) {
    if (!ctx || !reads_ok(reads)) return KMX_E_ARG;
    if (k < 1 || k > 31) return KMX_E_K_RANGE;
    if (reads->d_offsets && !d_win_offsets) return KMX_E_ARG;
    if (reads->n_reads == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    if (!reads->d_offsets && !d_win_offsets) {   // uniform layout: fast word-domain kernel
        bool handled = false;
        // (the window sinks mark the reads of a tile with an invalid byte and the sweep behind the passes zeroes their spoiled slots: round 6)
        KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
        if (int st = prepare_dirty_flags(ctx, reads->n_reads, k, k >= 2u /* (k = 1: no sweep behind the passes, so no marks) */)) return st;
        KMX_HIP(ctx, kmx::launch_windows_uniform(reads->d_bases, reads->n_reads, reads->read_len, k, d_fw, d_rc, d_canon,
                                                 d_flags, ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &handled));
        if (handled) return KMX_OK;
        // Uniform reads longer than a frame (round 4): every read as segments of at most 257 - k windows, each a read of its own for the ragged
        // materialise kernels (its start, its end, its first output slot: three arrays in the work buffer, 24 bytes per segment against
        // the ~1.8 KB a segment writes).  16-byte aligned base; no scratch -> the lane-per-read kernel below.
        const uint32_t L = reads->read_len;
        if (L > 256 && k >= 2 && k <= 31 /* the tiled materialise kernel's domain: nothing is planned for a k it refuses */ && (reinterpret_cast<uintptr_t>(reads->d_bases) & 15u) == 0u && reads->n_reads < (1ull << 40) && (uint64_t)L * reads->n_reads < (1ull << 62)) {
            // (as few segments as the 16-word frame allows, all of the same size but the last)
            const uint32_t W = L - k + 1u, J = (W + (257u - k) - 1u) / (257u - k), T = (W + J - 1u) / J;
            const uint64_t n_seg_host = reads->n_reads * J;
            void* scratch = capped_scratch(ctx, kmx::uniform_segments_scratch_bytes(n_seg_host));
            if (scratch) {
                ctx->fx_valid = false;   // (the work buffer is overwritten: the fastx chunk prefixes in it are gone)
                const uint64_t *starts = nullptr, *ends = nullptr, *wins = nullptr;
                uint64_t n_seg = 0;
                KMX_HIP(ctx, kmx::launch_uniform_segments_plan(reads->n_reads, L, k, T, scratch, &starts, &ends, &wins, &n_seg, ctx->stream));
                KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
                if (int st = prepare_dirty_flags(ctx, n_seg, k, k >= 2u /* (k = 1: no sweep behind the passes, so no marks) */)) return st;
                KMX_HIP(ctx, kmx::launch_windows_ragged(reads->d_bases, starts, wins, n_seg, 256u, k, d_fw, d_rc, d_canon, d_flags, ctx->d_scratch + 16, ctx->n_cu,
                                                        ctx->stream, &handled, ends));
                if (handled) return KMX_OK;
            }
        }
    }
    if (reads->d_offsets && d_win_offsets && reads->read_len > 256 && k >= 2 && (reinterpret_cast<uintptr_t>(reads->d_bases) & 15u) == 0u) {
        // long ragged reads (a length bound above the frames), round 4: cut into segments of at most 257 - k windows on the device
        // (kmx_segments.hip: a start, an end and a first output slot each; two host round trips), materialised as reads of their own
        const uint64_t *starts = nullptr, *ends = nullptr, *wins = nullptr;
        uint64_t n_seg = 0;
        const int st = long_ragged_segments(ctx, reads, k, 257u - k, &starts, &ends, &n_seg, d_win_offsets, &wins);
        if (st > 0) return st;
        if (st == 0) {
            if (n_seg == 0) return KMX_OK;      // no read holds a window
            bool handled = false;
            KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
            if (int st2 = prepare_dirty_flags(ctx, n_seg, k, k >= 2u /* (k = 1: no sweep behind the passes, so no marks) */)) return st2;
            KMX_HIP(ctx, kmx::launch_windows_ragged(reads->d_bases, starts, wins, n_seg, 256u, k, d_fw, d_rc, d_canon, d_flags, ctx->d_scratch + 16, ctx->n_cu,
                                                    ctx->stream, &handled, ends));
            if (handled) return KMX_OK;
        }
    }
    if (reads->d_offsets && d_win_offsets) {     // ragged reads: the tiled word-domain kernel (read_len = optional length bound)
        bool handled = false;
        KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
        if (int st = prepare_dirty_flags(ctx, reads->n_reads, k, k >= 2u /* (k = 1: no sweep behind the passes, so no marks) */)) return st;
        KMX_HIP(ctx, kmx::launch_windows_ragged(reads->d_bases, reads->d_offsets, d_win_offsets, reads->n_reads, reads->read_len, k,
                                                d_fw, d_rc, d_canon, d_flags, ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &handled));
        if (handled) return KMX_OK;
    }
    KMX_HIP(ctx, kmx::launch_windows_generic(reads, d_win_offsets, k, d_fw, d_rc, d_canon, d_flags, ctx->n_cu, ctx->stream, ctx->d_scratch + 8));
    return KMX_OK;
}

int kmx_canonical_reduce2(kmx_ctx* ctx, const kmx_reads* reads, uint32_t k, uint32_t with_hash, kmx_summary2* d_out) {
    if (!ctx || !reads_ok(reads) || !d_out) return KMX_E_ARG;
    if (k < 33 || k > 64) return KMX_E_K_RANGE;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(kmx_summary2), ctx->stream));
    if (reads->n_reads == 0) return KMX_OK;
    if (!reads->d_offsets) {
        bool handled = false;
        KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));   // queue heads + the "a tile was flagged" word + the uniform / ragged gate
        if (int st = prepare_dirty_flags(ctx, reads->n_reads * kmx::bitsliced_segments_per_read(reads->read_len, k), k)) return st;
        KMX_HIP(ctx, kmx::launch_scan_bitsliced2(reads->d_bases, reads->n_reads, reads->read_len, k, with_hash != 0, d_out,
                                                 ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &handled));
        if (handled) return KMX_OK;
    }
    // Reads behind an offsets array (what kmx_fastx_parse hands over), 16-byte aligned base: the tiled kernels.  With a length bound
    // the uniform two-word kernel takes, "is every read exactly that long" (untrimmed FASTQ) is decided on the device as
    // kmx_canonical_reduce does it: a small kernel checks offsets[i] == i*L, the uniform scan and the ragged one (the 10-word frame:
    // the reads themselves with a bound of at most 160 or none, segments of them above) are both launched behind its verdict, exactly one counts.
    const uint32_t Lh = reads->read_len;
    if (reads->d_offsets && (reinterpret_cast<uintptr_t>(reads->d_bases) & 15u) == 0u) {
        uint32_t* gate = reinterpret_cast<uint32_t*>(ctx->d_scratch + 16 + 513);
        KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
        // A bound above the 10-word frame (161...: 250-base reads, long reads): the ragged kernel scans SEGMENTS of at most 161 - k
        // windows, cut on the device as kmx_canonical_reduce does for long reads (two host round trips) -- first of all, so that the
        // masks of the dirty reads are sized once for whichever kernel will run.
        const uint64_t *starts = nullptr, *ends = nullptr;
        uint64_t n_seg = 0;
        bool segmented = false;
        bool cut = Lh > 160;
        if (cut && Lh <= 256) {
            // (a bound the uniform kernel takes: if the batch holds exactly n_reads * bound bases it is, almost certainly, untrimmed --
            // the gate below will say so for sure -- and the segments would be built for nothing: 16 bytes of offsets tell.  Should
            // the gate disagree, the lane-per-read kernel counts.)
            KMX_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, reads->d_offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
            KMX_HIP(ctx, hipMemcpyAsync(ctx->h_pinned + 1, reads->d_offsets + reads->n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
            KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            // (round 6: at ANY one length up to the bound -- 150-base reads handed over with a bound of 250 paid the segment cut, three host
            // round trips and a second set of marks, for the uniform kernel to run in the end: the gate passes every uniform length)
            const uint64_t o_first = ctx->h_pinned[0], o_last = ctx->h_pinned[1];
            if (o_first == 0 && o_last % reads->n_reads == 0 && o_last / reads->n_reads >= k && o_last / reads->n_reads <= Lh) cut = false;
        }
        if (cut) {
            const int st = long_ragged_segments(ctx, reads, k, 161u - k, &starts, &ends, &n_seg);
            if (st > 0) return st;
            segmented = st == 0;
            if (segmented && n_seg == 0) return KMX_OK;      // no read holds a window
        }
        if (int st = prepare_dirty_flags(ctx, segmented && n_seg > reads->n_reads ? n_seg : reads->n_reads, k)) return st;
        bool h_u = false, h_r = false;
        const uint32_t Lg = Lh ? Lh : 160u;   // (round 5: the gate passes reads that are uniform at ANY length up to the bound; no bound = the 160-base frame)
        if (Lg >= k && Lg <= 256) {
            KMX_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(gate), 1, 1, ctx->stream));
            KMX_HIP(ctx, kmx::launch_offsets_uniform_gate(reads->d_offsets, reads->n_reads, Lg, k, gate, ctx->n_cu, ctx->stream));
            KMX_HIP(ctx, kmx::launch_scan_bitsliced2(reads->d_bases, reads->n_reads, Lg, k, with_hash != 0, d_out, ctx->d_scratch + 16,
                                                     ctx->n_cu, ctx->stream, &h_u));
            // (not launched: the verdict must not keep the other kernel from running)
            if (!h_u) KMX_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(gate), 0, 2, ctx->stream));
        }
        if (segmented)
            KMX_HIP(ctx, kmx::launch_scan_bitsliced2_ragged(reads->d_bases, starts, n_seg, 160u, k, with_hash != 0, d_out, ctx->d_scratch + 16, ctx->n_cu,
                                                            ctx->stream, &h_r, ends));
        else if (Lh <= 160)
            KMX_HIP(ctx, kmx::launch_scan_bitsliced2_ragged(reads->d_bases, reads->d_offsets, reads->n_reads, Lh, k, with_hash != 0, d_out,
                                                            ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &h_r));
        if (!h_r) KMX_HIP(ctx, kmx::launch_reduce2_generic(reads, k, with_hash, d_out, ctx->n_cu, ctx->stream, ctx->d_scratch + 8, h_u ? gate : nullptr));
        if (h_u) KMX_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(gate), 0, 2, ctx->stream));   // never left armed
        return KMX_OK;
    }
    KMX_HIP(ctx, kmx::launch_reduce2_generic(reads, k, with_hash, d_out, ctx->n_cu, ctx->stream, ctx->d_scratch + 8, nullptr));
    return KMX_OK;
}

int kmx_canonical_windows2(kmx_ctx* ctx, const kmx_reads* reads, const uint64_t* d_win_offsets, uint32_t k,
                           uint64_t* d_fw2, uint64_t* d_rc2, uint64_t* d_canon2, uint8_t* d_flags) {
    if (!ctx || !reads_ok(reads)) return KMX_E_ARG;
    if (k < 33 || k > 64) return KMX_E_K_RANGE;
    if (reads->d_offsets && !d_win_offsets) return KMX_E_ARG;
    if (reads->n_reads == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    bool handled = false;   // uniform reads of up to 256 bases in the dense layout (slot r*W + p): the tiled kernel (kmx_generic.hip)
    if (!reads->d_offsets && !d_win_offsets) {   // (a caller's win_offsets for uniform reads are honoured by the lane-per-read kernel, as kmx_canonical_windows does)
        // (a tile with an invalid byte stays on the tiled path: its reads are marked, the sweep behind the passes zeroes the spoiled slots -- round 6)
        KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
        if (int st = prepare_dirty_flags(ctx, reads->n_reads, k, true)) return st;
        KMX_HIP(ctx, kmx::launch_windows2_tiled(reads, k, d_fw2, d_rc2, d_canon2, d_flags, ctx->n_cu, ctx->stream, &handled, ctx->d_scratch + 16));
    }
    if (!handled && !reads->d_offsets && !d_win_offsets && reads->read_len > 256 && (reinterpret_cast<uintptr_t>(reads->d_bases) & 15u) == 0u &&
        reads->n_reads < (1ull << 40) && (uint64_t)reads->read_len * reads->n_reads < (1ull << 62)) {
        // uniform reads longer than a frame (round 4): planned as segments on the device, as kmx_canonical_windows does
        const uint32_t L = reads->read_len, W = L - k + 1u, J = (W + (257u - k) - 1u) / (257u - k), T = (W + J - 1u) / J;
        void* scratch = capped_scratch(ctx, kmx::uniform_segments_scratch_bytes(reads->n_reads * J));
        if (scratch) {
            ctx->fx_valid = false;
            const uint64_t *starts = nullptr, *ends = nullptr, *wins = nullptr;
            uint64_t n_seg = 0;
            KMX_HIP(ctx, kmx::launch_uniform_segments_plan(reads->n_reads, L, k, T, scratch, &starts, &ends, &wins, &n_seg, ctx->stream));
            kmx_reads segs = *reads;
            segs.d_offsets = starts;
            segs.n_reads = n_seg;
            segs.read_len = 256u;
            KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
            if (int stf = prepare_dirty_flags(ctx, segs.n_reads, k, true)) return stf;
            KMX_HIP(ctx, kmx::launch_windows2_tiled_ragged(&segs, wins, k, d_fw2, d_rc2, d_canon2, d_flags, ctx->n_cu, ctx->stream, &handled, ctx->d_scratch + 8, ends, ctx->d_scratch + 16));
        }
    }
    if (!handled && reads->d_offsets && d_win_offsets && reads->read_len > 256 && (reinterpret_cast<uintptr_t>(reads->d_bases) & 15u) == 0u) {
        // long ragged reads (a length bound above the frames): segments cut on the device, as kmx_canonical_windows does
        const uint64_t *starts = nullptr, *ends = nullptr, *wins = nullptr;
        uint64_t n_seg = 0;
        const int st = long_ragged_segments(ctx, reads, k, 257u - k, &starts, &ends, &n_seg, d_win_offsets, &wins);
        if (st > 0) return st;
        if (st == 0) {
            if (n_seg == 0) return KMX_OK;
            kmx_reads segs = *reads;
            segs.d_offsets = starts;
            segs.n_reads = n_seg;
            segs.read_len = 256u;
            KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
            if (int stf = prepare_dirty_flags(ctx, segs.n_reads, k, true)) return stf;
            KMX_HIP(ctx, kmx::launch_windows2_tiled_ragged(&segs, wins, k, d_fw2, d_rc2, d_canon2, d_flags, ctx->n_cu, ctx->stream, &handled, ctx->d_scratch + 8, ends, ctx->d_scratch + 16));
        }
    }
    if (!handled && reads->d_offsets && d_win_offsets) {   // ragged reads (round 4): tiled too; read_len = optional length bound
        KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));
        if (int stf = prepare_dirty_flags(ctx, reads->n_reads, k, true)) return stf;
        KMX_HIP(ctx, kmx::launch_windows2_tiled_ragged(reads, d_win_offsets, k, d_fw2, d_rc2, d_canon2, d_flags, ctx->n_cu, ctx->stream, &handled,
                                                       ctx->d_scratch + 8, nullptr, ctx->d_scratch + 16));
    }
    if (handled) return KMX_OK;
    KMX_HIP(ctx, kmx::launch_windows2_generic(reads, d_win_offsets, k, d_fw2, d_rc2, d_canon2, d_flags, ctx->n_cu, ctx->stream, ctx->d_scratch + 8));
    return KMX_OK;
}

int kmx_histogram(kmx_ctx* ctx, const kmx_reads* reads, uint32_t k, uint32_t hasher, uint32_t hasher_k,
                  uint32_t log2_buckets, uint64_t* d_counts) {
    if (!ctx || !reads_ok(reads) || !d_counts) return KMX_E_ARG;
    if (k < 1 || k > 31) return KMX_E_K_RANGE;
    if (hasher > KMX_HASH_IDENTITY || log2_buckets > 30) return KMX_E_ARG;
    if (hasher == KMX_HASH_LEX && (hasher_k < 1 || hasher_k > 32)) return KMX_E_K_RANGE;
    if (reads->n_reads == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    ctx->fx_valid = false;   // (the partitioned histogram writes its id streams over the work buffer)
    {
        bool handled = false;
        KMX_HIP(ctx, queue_clear(ctx, 32 * 128 + 16));   // queue heads + the count of marked reads + the gate
        // (the histogram sinks mark the reads of a dirty tile like the bit-sliced scan does; without the array they roll such tiles, exactly)
        if (k >= 2 && k <= 31) {
            if (int st = prepare_dirty_flags(ctx, reads->n_reads, k, true)) return st;
        }
        KMX_HIP(ctx, kmx::launch_hist_uniform(reads->d_bases, reads->n_reads, reads->read_len, k, hasher, hasher_k,
                                              log2_buckets, d_counts, ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &handled,
                                              &big_scratch, ctx, hist_scratch_budget(ctx->big_bytes, ctx->big_limit), reads->d_offsets));
        if (handled) return KMX_OK;
    }
    KMX_HIP(ctx, kmx::launch_histogram_generic(reads, k, hasher, hasher_k, log2_buckets, d_counts, ctx->n_cu, ctx->stream, ctx->d_scratch + 8));
    return KMX_OK;
}

int kmx_gen_reads(kmx_ctx* ctx, uint64_t seed, uint64_t first_byte, uint8_t* d_out, uint64_t nbytes) {
    if (!ctx || (nbytes && !d_out)) return KMX_E_ARG;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_gen_reads(seed, first_byte, d_out, nbytes, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

/* -------------------------------------------------------- element-wise ---- */

int kmx_kmers_from_bytes(kmx_ctx* ctx, const uint8_t* d_seqs, uint64_t n, uint32_t k, uint64_t* d_words,
                         uint64_t* h_first_bad) {
    if (!ctx || (n && (!d_seqs || !d_words))) return KMX_E_ARG;
    if (k > 32) return KMX_E_TOO_LONG;  // kmer.rs:236-238
    if (k < 1) return KMX_E_K_RANGE;
    if (h_first_bad) *h_first_bad = ~0ull;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0xFF, 8, ctx->stream));
    KMX_HIP(ctx, kmx::launch_kmers_from_bytes(d_seqs, n, k, d_words, ctx->d_scratch, ctx->n_cu, ctx->stream));
    unsigned long long bad = 0;
    KMX_HIP(ctx, hipMemcpyAsync(&bad, ctx->d_scratch, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bad != ~0ull) {
        if (h_first_bad) *h_first_bad = bad;
        return KMX_E_INVALID_BASE;
    }
    return KMX_OK;
}

int kmx_revcomp_words(kmx_ctx* ctx, const uint64_t* d_in, uint64_t n, uint32_t k, uint64_t* d_out) {
    if (!ctx || (n && (!d_in || !d_out))) return KMX_E_ARG;
    if (k < 1 || k > 32) return KMX_E_K_RANGE;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_revcomp_words(d_in, n, k, d_out, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_canonical_words(kmx_ctx* ctx, const uint64_t* d_in, uint64_t n, uint32_t k, uint64_t* d_canon,
                        uint8_t* d_is_canonical) {
    if (!ctx || (n && !d_in)) return KMX_E_ARG;
    if (k < 1 || k > 32) return KMX_E_K_RANGE;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_canonical_words(d_in, n, k, d_canon, d_is_canonical, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_hash_words(kmx_ctx* ctx, const uint64_t* d_in, uint64_t n, uint32_t hasher, uint32_t hasher_k, uint64_t* d_out) {
    if (!ctx || (n && (!d_in || !d_out))) return KMX_E_ARG;
    if (hasher != KMX_HASH_LEX && hasher != KMX_HASH_IDENTITY) return KMX_E_ARG;
    if (hasher == KMX_HASH_LEX && (hasher_k < 1 || hasher_k > 32)) return KMX_E_K_RANGE;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_hash_words(d_in, n, hasher, hasher_k, d_out, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_hash_words_sip13(kmx_ctx* ctx, const uint64_t* d_in, uint64_t n, uint64_t key0, uint64_t key1, uint64_t* d_out) {
    if (!ctx || (n && (!d_in || !d_out))) return KMX_E_ARG;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_hash_words_sip13(d_in, n, key0, key1, d_out, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_match_words(kmx_ctx* ctx, const uint64_t* d_fw, const uint64_t* d_rc, const uint64_t* d_other, uint64_t n,
                    uint8_t* d_out) {
    if (!ctx || (n && (!d_fw || !d_rc || !d_other || !d_out))) return KMX_E_ARG;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_match_words(d_fw, d_rc, d_other, n, d_out, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

static int ck_shift(kmx_ctx* ctx, bool append, uint64_t* d_fw, uint64_t* d_rc, const uint8_t* d_bases, uint64_t n,
                    uint32_t k, uint8_t* d_dropped) {
    if (!ctx || (n && (!d_fw || !d_rc || !d_bases))) return KMX_E_ARG;
    if (k < 1 || k > 31) return KMX_E_K_RANGE;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_ck_shift(append, d_fw, d_rc, d_bases, n, k, d_dropped, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_ck_append_bases(kmx_ctx* ctx, uint64_t* d_fw, uint64_t* d_rc, const uint8_t* d_bases, uint64_t n, uint32_t k,
                        uint8_t* d_dropped) {
    return ck_shift(ctx, true, d_fw, d_rc, d_bases, n, k, d_dropped);
}

int kmx_ck_prepend_bases(kmx_ctx* ctx, uint64_t* d_fw, uint64_t* d_rc, const uint8_t* d_bases, uint64_t n, uint32_t k,
                         uint8_t* d_dropped) {
    return ck_shift(ctx, false, d_fw, d_rc, d_bases, n, k, d_dropped);
}

int kmx_encode_kmers(kmx_ctx* ctx, const uint8_t* d_seqs, uint64_t n, uint32_t seq_len, uint8_t enc_byte,
                     uint32_t words_per_kmer, uint64_t* d_words) {
    if (!ctx || (n && ((seq_len && !d_seqs) || !d_words))) return KMX_E_ARG;
    if (!enc_ok(enc_byte) || words_per_kmer < 1 || words_per_kmer > 4) return KMX_E_ARG;
    if (seq_len > 32u * words_per_kmer) return KMX_E_TOO_LONG;  // bit_field set_bits would panic
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_encode_kmers(d_seqs, n, seq_len, enc_byte, words_per_kmer, d_words, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_encode_windows(kmx_ctx* ctx, const kmx_reads* reads, uint32_t k, uint8_t enc_byte, uint32_t words_per_kmer,
                       uint64_t* d_words) {
    if (!ctx || !reads_ok(reads) || reads->d_offsets) return KMX_E_ARG;
    if (!enc_ok(enc_byte) || words_per_kmer < 1 || words_per_kmer > 4) return KMX_E_ARG;
    if (k < 1) return KMX_E_K_RANGE;
    if (k > 32u * words_per_kmer) return KMX_E_TOO_LONG;
    if (reads->n_reads == 0 || reads->read_len < k) return KMX_OK;
    if (!d_words) return KMX_E_ARG;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_encode_windows(reads->d_bases, reads->n_reads, reads->read_len, k, enc_byte, words_per_kmer,
                                            d_words, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_encoding_rev_comp(kmx_ctx* ctx, const uint64_t* d_in, uint64_t n, uint32_t K, uint8_t enc_byte,
                          uint32_t words_per_kmer, uint64_t* d_out) {
    if (!ctx || (n && (!d_in || !d_out))) return KMX_E_ARG;
    if (!enc_ok(enc_byte) || words_per_kmer < 1 || words_per_kmer > 4) return KMX_E_ARG;
    if (K < 2 || K > 32u * words_per_kmer) return KMX_E_K_RANGE;  // K=1 underflows usize in the reference (naive.rs:140,150)
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_encoding_rev_comp(d_in, n, K, comp_lut_for(enc_byte), words_per_kmer, d_out, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_encoding_decode(kmx_ctx* ctx, const uint64_t* d_in, uint64_t n, uint8_t enc_byte, uint32_t words_per_kmer,
                        uint8_t* d_seqs) {
    if (!ctx || (n && (!d_in || !d_seqs))) return KMX_E_ARG;
    if (!enc_ok(enc_byte) || words_per_kmer < 1 || words_per_kmer > 4) return KMX_E_ARG;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_encoding_decode(d_in, n, nuc_lut_for(enc_byte), words_per_kmer, d_seqs, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

/* ------------------------------------------------ decode / display (f3) ---- */

int kmx_sub_kmer_words(kmx_ctx* ctx, const uint64_t* d_words, uint64_t n, uint32_t k, uint32_t pos, uint32_t width,
                       uint64_t* d_out) {
    if (!ctx || (n && (!d_words || !d_out))) return KMX_E_ARG;
    if (k < 1 || k > 32) return KMX_E_K_RANGE;
    if (!(pos < k) || !(pos + width <= k)) return KMX_E_ARG;   // the reference's two asserts (kmer.rs:157-158)
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_sub_kmer_words(d_words, n, pos, width, d_out, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_kmers_to_strings(kmx_ctx* ctx, const uint64_t* d_words, uint64_t n, uint32_t k, uint8_t* d_out) {
    if (!ctx || (n && k && (!d_words || !d_out))) return KMX_E_ARG;
    if (k > 32) return KMX_E_K_RANGE;
    if (n == 0 || k == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_kmers_to_bytes(d_words, n, k, false, d_out, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_bitmers_to_bytes(kmx_ctx* ctx, const uint64_t* d_mers, uint64_t n, uint32_t len, uint8_t* d_out) {
    if (!ctx || (n && len && (!d_mers || !d_out))) return KMX_E_ARG;
    if (len > 32) return KMX_E_K_RANGE;
    if (n == 0 || len == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_kmers_to_bytes(d_mers, n, len, true, d_out, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

/* ------------------------------------------------ Encoding<P, B>, any utils::Data P ---- */

static int p_shape_ok(uint32_t word_bits, uint32_t words_per_kmer, uint32_t* nb) {
    if (word_bits != 8 && word_bits != 16 && word_bits != 32 && word_bits != 64 && word_bits != 128) return KMX_E_ARG;
    if (words_per_kmer < 1) return KMX_E_ARG;
    const uint64_t bytes = (uint64_t)(word_bits / 8u) * words_per_kmer;
    if (bytes > 64) return KMX_E_ARG;
    *nb = (uint32_t)bytes;
    return KMX_OK;
}

int kmx_encode_kmers_p(kmx_ctx* ctx, const uint8_t* d_seqs, uint64_t n, uint32_t seq_len, uint8_t enc_byte, uint32_t word_bits,
                       uint32_t words_per_kmer, void* d_arrays) {
    uint32_t nb = 0;
    if (!ctx || (n && ((seq_len && !d_seqs) || !d_arrays))) return KMX_E_ARG;
    if (!enc_ok(enc_byte)) return KMX_E_ARG;
    if (int st = p_shape_ok(word_bits, words_per_kmer, &nb)) return st;
    if (seq_len > 4u * nb) return KMX_E_TOO_LONG;   // bit_field set_bits would panic (naive.rs:120)
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_encode_kmers_bytes(d_seqs, n, seq_len, enc_byte, nb, static_cast<uint8_t*>(d_arrays), ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_encoding_rev_comp_p(kmx_ctx* ctx, const void* d_in, uint64_t n, uint32_t K, uint8_t enc_byte, uint32_t word_bits,
                            uint32_t words_per_kmer, void* d_out) {
    uint32_t nb = 0;
    if (!ctx || (n && (!d_in || !d_out))) return KMX_E_ARG;
    if (!enc_ok(enc_byte)) return KMX_E_ARG;
    if (int st = p_shape_ok(word_bits, words_per_kmer, &nb)) return st;
    if (K < 2 || K > 4u * nb) return KMX_E_K_RANGE;   // K=1 underflows usize in the reference (naive.rs:140,150)
    if (n == 0) return KMX_OK;
    if (d_in == d_out) return KMX_E_ARG;              // base i reads base K-1-i of the input
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_encoding_rev_comp_bytes(static_cast<const uint8_t*>(d_in), n, K, comp_lut_for(enc_byte), nb,
                                                     static_cast<uint8_t*>(d_out), ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_encoding_decode_p(kmx_ctx* ctx, const void* d_in, uint64_t n, uint8_t enc_byte, uint32_t word_bits,
                          uint32_t words_per_kmer, uint8_t* d_seqs) {
    uint32_t nb = 0;
    if (!ctx || (n && (!d_in || !d_seqs))) return KMX_E_ARG;
    if (!enc_ok(enc_byte)) return KMX_E_ARG;
    if (int st = p_shape_ok(word_bits, words_per_kmer, &nb)) return st;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_encoding_decode_bytes(static_cast<const uint8_t*>(d_in), n * (uint64_t)nb, nuc_lut_for(enc_byte), d_seqs,
                                                   ctx->n_cu, ctx->stream));
    return KMX_OK;
}

/* ------------------------------------------------ measurement helper ---- */

int kmx_calib_stream_read(kmx_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint64_t* d_out) {
    if (!ctx || !d_out || (nbytes && !d_buf)) return KMX_E_ARG;
    if (reinterpret_cast<uintptr_t>(d_buf) & 15u) return KMX_E_ARG;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipMemsetAsync(d_out, 0, 8, ctx->stream));
    if (nbytes < 16) return KMX_OK;
    KMX_HIP(ctx, kmx::launch_calib_stream_read(d_buf, nbytes, reinterpret_cast<unsigned long long*>(d_out), ctx->n_cu, ctx->stream));
    return KMX_OK;
}

/* ------------------------------------------------------------- SeqVector ---- */

int kmx_seqvec_push_chars(kmx_ctx* ctx, uint64_t* d_words, uint64_t n_bases_before, const uint8_t* d_bytes, uint64_t n,
                          uint64_t* h_first_bad) {
    if (!ctx || (n && (!d_words || !d_bytes))) return KMX_E_ARG;
    if (h_first_bad) *h_first_bad = ~0ull;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0xFF, 8, ctx->stream));
    KMX_HIP(ctx, kmx::launch_seqvec_push(d_words, n_bases_before, d_bytes, n, ctx->d_scratch, ctx->n_cu, ctx->stream));
    unsigned long long bad = 0;
    KMX_HIP(ctx, hipMemcpyAsync(&bad, ctx->d_scratch, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bad != ~0ull) {
        if (h_first_bad) *h_first_bad = bad;
        return KMX_E_INVALID_BASE;
    }
    return KMX_OK;
}

int kmx_seqvec_to_bytes(kmx_ctx* ctx, const uint64_t* d_words, uint64_t n_bases, uint8_t* d_bytes) {
    if (!ctx || (n_bases && (!d_words || !d_bytes))) return KMX_E_ARG;
    if (n_bases == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_seqvec_to_bytes(d_words, n_bases, d_bytes, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_seqvec_get_kmers(kmx_ctx* ctx, const uint64_t* d_words, uint64_t n_bases, const uint64_t* d_pos, uint64_t n,
                         uint32_t k, uint64_t* d_out) {
    if (!ctx || (n && (!d_words || !d_pos || !d_out))) return KMX_E_ARG;
    if (k < 1 || k > 32) return KMX_E_K_RANGE;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0xFF, 8, ctx->stream));
    KMX_HIP(ctx, kmx::launch_seqvec_get_kmers(d_words, n_bases, d_pos, n, k, d_out, ctx->d_scratch, ctx->n_cu, ctx->stream));
    unsigned long long bad = 0;
    KMX_HIP(ctx, hipMemcpyAsync(&bad, ctx->d_scratch, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bad != ~0ull) {
        std::snprintf(ctx->last_error, sizeof ctx->last_error, "kmx_seqvec_get_kmers: element %llu lies outside the vector", bad);
        return KMX_E_ARG;
    }
    return KMX_OK;
}

int kmx_seqvec_iter_kmers(kmx_ctx* ctx, const uint64_t* d_words, uint64_t n_bases, uint64_t start, uint64_t end, uint32_t k,
                          uint64_t* d_out) {
    if (!ctx || start > end || end > n_bases) return KMX_E_ARG;
    if (k < 1 || k > 32) return KMX_E_K_RANGE;
    if (end - start < k) return KMX_OK;   // the reference's `len - k + 1` underflows here; an empty iteration is the only sane answer
    if (!d_words || !d_out) return KMX_E_ARG;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_seqvec_iter_kmers(d_words, n_bases, start, end - start - k + 1u, k, d_out, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_seqvec_canonical_reduce(kmx_ctx* ctx, const uint64_t* d_words, uint64_t n_reads, uint32_t read_len, uint32_t k,
                                uint32_t hasher, uint32_t hasher_k, uint32_t flags, kmx_summary* d_out) {
    if (!ctx || !d_out || (n_reads && !d_words)) return KMX_E_ARG;
    if (k < 1 || k > 31) return KMX_E_K_RANGE;   // rolling with MASK_TABLE[32] == 0 is unusable (kmer.rs:617)
    if (hasher > KMX_HASH_IDENTITY) return KMX_E_ARG;
    if (hasher == KMX_HASH_LEX && (hasher_k < 1 || hasher_k > 32)) return KMX_E_K_RANGE;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(kmx_summary), ctx->stream));
    if (n_reads == 0 || read_len < k) return KMX_OK;
    const bool want_hash = hasher != KMX_HASH_NONE, want_sumfw = (flags & KMX_REDUCE_SUM_FW) != 0;
    const bool fix_fold = want_hash && !(hasher == KMX_HASH_LEX && hasher_k == k);   // (as in kmx_canonical_reduce)
    bool handled = false;
    KMX_HIP(ctx, queue_clear(ctx, 32 * 128));
    KMX_HIP(ctx, kmx::launch_scan_bitsliced_packed(d_words, n_reads, read_len, k, want_hash, want_sumfw, d_out,
                                                   ctx->d_scratch + 16, ctx->n_cu, ctx->stream, &handled));
    if (!handled)
        KMX_HIP(ctx, kmx::launch_reduce_packed_generic(d_words, n_reads, read_len, k, want_hash, want_sumfw, d_out, ctx->n_cu, ctx->stream));
    if (fix_fold) KMX_HIP(ctx, kmx::launch_fix_hash_fold(d_out, k, hasher, hasher_k, ctx->stream));
    return KMX_OK;
}

/* ------------------------------------------------------------- minimizers ---- */

static int mm_hasher_ok(uint32_t hasher, uint32_t hasher_k) {
    if (hasher == KMX_HASH_IDENTITY) return KMX_OK;
    if (hasher != KMX_HASH_LEX) return KMX_E_ARG;
    return (hasher_k >= 1 && hasher_k <= 32) ? KMX_OK : KMX_E_K_RANGE;
}

int kmx_minimizer_words(kmx_ctx* ctx, const uint64_t* d_words, uint64_t n, uint32_t k, uint32_t width, uint32_t hasher,
                        uint32_t hasher_k, uint64_t* d_mmer, uint32_t* d_offset) {
    if (!ctx || (n && (!d_words || !d_mmer || !d_offset))) return KMX_E_ARG;
    if (k < 1 || k > 32 || width < 1 || width > k) return KMX_E_K_RANGE;   // sub_kmer_word asserts pos + width <= k
    if (int st = mm_hasher_ok(hasher, hasher_k)) return st;
    if (n == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_minimizer_words(d_words, n, k, width, hasher, hasher_k, d_mmer, d_offset, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_seqvec_minimizers(kmx_ctx* ctx, const uint64_t* d_words, uint64_t n_reads, uint32_t read_len, uint32_t k, uint32_t w,
                          uint32_t hasher, uint32_t hasher_k, uint64_t* d_word, uint32_t* d_pos) {
    if (!ctx || (n_reads && (!d_words || !d_word || !d_pos))) return KMX_E_ARG;
    if (k < 1 || w < 1 || w > k || w > 32) return KMX_E_K_RANGE;
    if (read_len < k) return KMX_E_ARG;   // SeqVecMinimizerIter::new: assert!(sv.len() >= k)
    if (int st = mm_hasher_ok(hasher, hasher_k)) return st;
    if (n_reads == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    KMX_HIP(ctx, kmx::launch_seqvec_minimizers(d_words, n_reads, read_len, k, w, hasher, hasher_k, d_word, d_pos, ctx->n_cu, ctx->stream));
    return KMX_OK;
}

int kmx_minimizers(kmx_ctx* ctx, const kmx_reads* reads, const uint64_t* d_win_offsets, uint32_t k, uint32_t w, uint32_t hasher,
                   uint32_t hasher_k, uint64_t* d_word, uint32_t* d_pos, uint64_t* h_first_bad) {
    if (!ctx || !reads_ok(reads)) return KMX_E_ARG;
    if (k < 1 || w < 1 || w > k || w > 32) return KMX_E_K_RANGE;
    if (int st = mm_hasher_ok(hasher, hasher_k)) return st;
    if (h_first_bad) *h_first_bad = ~0ull;
    if (reads->n_reads == 0) return KMX_OK;
    if (!d_word || !d_pos) return KMX_E_ARG;
    if (reads->d_offsets && !d_win_offsets) return KMX_E_ARG;
    if (!reads->d_offsets && reads->read_len < k) return KMX_E_ARG;   // SeqVecMinimizerIter::new: assert!(sv.len() >= k) (a ragged read shorter than k owns no slot)
    DeviceGuard g(ctx->device);
    uint64_t total_bytes = reads->n_reads * (uint64_t)reads->read_len;
    uint32_t bound = reads->read_len;
    if (reads->d_offsets) {
        // the batch's last byte and its longest read (the bound of kmx_reads is a hint; the key of the sliding minimum holds 8 bits
        // of position): one host round trip
        uint32_t lo = 0, hi = 0;
        if (int st = kmx_reads_length_range(ctx, reads->d_offsets, reads->n_reads, &lo, &hi)) return st;
        KMX_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, reads->d_offsets + reads->n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
        KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        total_bytes = ctx->h_pinned[0];
        bound = hi;
        if (hi < k) return KMX_OK;      // no read holds a k-mer
    }
    KMX_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0xFF, 8, ctx->stream));
    bool tiled = false;
    KMX_HIP(ctx, kmx::launch_minimizers_reads(reads->d_bases, total_bytes, reads->d_offsets, d_win_offsets, reads->n_reads, reads->read_len, bound, k, w,
                                              hasher, hasher_k, d_word, d_pos, ctx->d_scratch, ctx->n_cu, ctx->stream, &tiled));
    if (!h_first_bad) return KMX_OK;
    KMX_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_pinned[0] != ~0ull) {
        *h_first_bad = ctx->h_pinned[0];
        return KMX_E_INVALID_BASE;
    }
    return KMX_OK;
}

int kmx_fastx_parse(kmx_ctx* ctx, const uint8_t* d_text, uint64_t n_bytes, uint32_t format, uint8_t* d_bases,
                    uint64_t* d_offsets, uint64_t max_reads, uint64_t* h_n_reads, uint64_t* h_n_bases) {
    const bool same_text = (format & KMX_FASTX_SAME_TEXT) != 0u;
    format &= ~KMX_FASTX_SAME_TEXT;
    if (!ctx || format > KMX_FASTX_FASTA || (n_bytes && !d_text) || (!d_bases != !d_offsets)) return KMX_E_ARG;
    if (reinterpret_cast<uintptr_t>(d_text) & 15u) return KMX_E_ARG;
    if (h_n_reads) *h_n_reads = 0;
    if (h_n_bases) *h_n_bases = 0;
    DeviceGuard g(ctx->device);
    if (n_bytes == 0) {
        if (d_offsets) {
            KMX_HIP(ctx, hipMemsetAsync(d_offsets, 0, 8, ctx->stream));
            KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        return KMX_OK;
    }
    // (the emit call of a counts-then-emit pair: the counting call looked at the first byte and settled the format)
    const bool reuse = same_text && ctx->fx_valid && ctx->fx_text == d_text && ctx->fx_bytes == n_bytes &&
                       (format == KMX_FASTX_AUTO || ctx->fx_fasta == (format == KMX_FASTX_FASTA ? 1u : 0u));
    if (reuse) {
        format = ctx->fx_fasta ? KMX_FASTX_FASTA : KMX_FASTX_FASTQ;
    } else {
        uint8_t first = 0;
        KMX_HIP(ctx, hipMemcpyAsync(&first, d_text, 1, hipMemcpyDeviceToHost, ctx->stream));
        KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (format == KMX_FASTX_AUTO) format = first == '@' ? KMX_FASTX_FASTQ : first == '>' ? KMX_FASTX_FASTA : 3u;
        if (format == 3u || first != (format == KMX_FASTX_FASTQ ? '@' : '>')) {
            std::snprintf(ctx->last_error, sizeof ctx->last_error, "kmx_fastx_parse: the text starts with byte 0x%02x, not with '@' (FASTQ) or '>' (FASTA)", first);
            return KMX_E_ARG;
        }
    }
    const bool fasta = format == KMX_FASTX_FASTA;
    void* scratch = big_scratch(ctx, kmx::fastx_scratch_bytes(n_bytes));
    if (!scratch) return KMX_E_NOMEM;
    unsigned long long* d_totals = ctx->d_scratch + 2;
    unsigned long long totals[2] = {0, 0};
    if (reuse && scratch == ctx->d_big) {
        // the chunk prefixes of the counting call are still in the work buffer, its totals in d_scratch[2..3]
        totals[0] = ctx->fx_totals[0];
        totals[1] = ctx->fx_totals[1];
    } else {
        ctx->fx_valid = false;
        KMX_HIP(ctx, kmx::launch_fastx_count(d_text, n_bytes, fasta, scratch, d_totals, ctx->stream));
        KMX_HIP(ctx, hipMemcpyAsync(totals, d_totals, 16, hipMemcpyDeviceToHost, ctx->stream));
        KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->fx_text = d_text;
        ctx->fx_bytes = n_bytes;
        ctx->fx_fasta = fasta ? 1u : 0u;
        ctx->fx_totals[0] = totals[0];
        ctx->fx_totals[1] = totals[1];
        ctx->fx_valid = true;
    }
    if (h_n_reads) *h_n_reads = totals[0];
    if (h_n_bases) *h_n_bases = totals[1];
    if (!d_bases) return KMX_OK;
    if (totals[0] > max_reads) {
        std::snprintf(ctx->last_error, sizeof ctx->last_error, "kmx_fastx_parse: %llu reads, room for %llu", totals[0],
                      (unsigned long long)max_reads);
        return KMX_E_NOMEM;
    }
    KMX_HIP(ctx, kmx::launch_fastx_emit(d_text, n_bytes, fasta, scratch, d_totals, d_bases, d_offsets, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KMX_OK;
}

int kmx_reads_length_range(kmx_ctx* ctx, const uint64_t* d_offsets, uint64_t n_reads, uint32_t* h_min_len, uint32_t* h_max_len) {
    if (!ctx || (n_reads && !d_offsets)) return KMX_E_ARG;
    if (h_min_len) *h_min_len = 0;
    if (h_max_len) *h_max_len = 0;
    if (n_reads == 0) return KMX_OK;
    DeviceGuard g(ctx->device);
    uint32_t* d_out = reinterpret_cast<uint32_t*>(ctx->d_scratch + 4);   // (d_scratch[4]: free between calls)
    const uint32_t init[2] = {0xFFFFFFFFu, 0u};
    uint32_t got[2] = {0, 0};
    KMX_HIP(ctx, hipMemcpyAsync(d_out, init, 8, hipMemcpyHostToDevice, ctx->stream));
    KMX_HIP(ctx, kmx::launch_length_range(d_offsets, n_reads, d_out, ctx->n_cu, ctx->stream));
    KMX_HIP(ctx, hipMemcpyAsync(got, d_out, 8, hipMemcpyDeviceToHost, ctx->stream));
    KMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_min_len) *h_min_len = got[0];
    if (h_max_len) *h_max_len = got[1];
    if (got[1] >= 0x80000000u) {   // (lengths are clamped to 32 bits by the kernel: anything from 2^31 up lands here)
        std::snprintf(ctx->last_error, sizeof ctx->last_error, "kmx_reads_length_range: a read of 2^31 bases or more (the scans skip such reads)");
        return KMX_E_ARG;
    }
    return KMX_OK;
}

}  // extern "C"
