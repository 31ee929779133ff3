// kmx_hist.hip -- bucket histograms of hash(canonical k-mer) over the word-domain scan kernel (kmx_scan_kernel.h): the sinks
// (device-scope atomics, block-private LDS tables, the two-pass partitioned histogram), pass 2 of the latter, and
// launch_hist_uniform.  BUILD-DEFINED (the reference has no histogram): include/kmx.h kmx_histogram, oracle kmo_histogram.
#include "kmx_hist_part.h"

namespace kmx {

struct HistParams {
    u64* counts;
    u32 hasher, hk, log2_buckets;
};
// d_counts[bucket(hash(canonical k-mer))] += 1 with device-scope u64 atomics.  Measured ~24 G atomics/s on MI355X
// independent of the bucket count (2^12..2^26) and of the atomic scope (XCD-private copies updated with
// workgroup-scope atomics ran at the same rate), i.e. bound by the atomic issue rate, not by contention.
struct SinkHist {
    u64* counts;
    u64 maskk;
    u32 hasher, hk, k, b;
    static constexpr u32 kLdsDwordsPerWave = 0;
    __device__ __forceinline__ void block_done(u64, u32, u32) {}
    static constexpr bool kRagged = true;
    static constexpr bool kMarksDirty = true;   // (a tile with an invalid byte: fast path + marks, kmx_scan_kernel.h; launch_hist_uniform sweeps behind every scan)
    static u32 block_lds_dwords(const HistParams&) { return 0; }
    __device__ SinkHist(const HistParams& p, u32 k_, u32, u32*, u32, u32*, u32)
        : counts(p.counts), maskk(mask2k(k_)), hasher(p.hasher), hk(p.hk), k(k_), b(p.log2_buckets) {}
    __device__ __forceinline__ void emit(u64 fw, u64 rc) {
        const u64 canon = fw < rc ? fw : rc;
        u64 h;
        if (hasher == KMX_HASH_LEX) h = (hk == k) ? (maskk ^ fw ^ rc ^ canon) : lex_hash(canon, hk);
        else h = canon;  // identity: write_u64(data), hash.rs:4-8
        atomicAdd((unsigned long long*)&counts[bucket_of(h, b)], 1ull);
    }
    __device__ __forceinline__ void fast(u32, u64 fw, u64 rc) { emit(fw, rc); }
    __device__ __forceinline__ void slow(u32, u64 fw, u64 rc) { emit(fw, rc); }
    __device__ __forceinline__ void begin_read(u64) {}
    __device__ __forceinline__ void slow_block(u32) {}   // a rolled tile: the wave has completed 16 more windows per read
    __device__ __forceinline__ void tile_slow_begin(u64 read) { begin_read(read); }
    __device__ __forceinline__ void tile_slow_emit(u32 pos, u64 fw, u64 rc) { slow(pos, fw, rc); }
    __device__ __forceinline__ void tile_slow_end() { end_read(); }
    __device__ __forceinline__ void end_read() {}
    __device__ __forceinline__ void tile_fast_done(u32) {}
    __device__ __forceinline__ void finish(const HistParams&) {}
};

// Histogram, 2^b <= 2^14 buckets: block-private u32 table in LDS (ds_add_u32, no return), merged into d_counts
// with one u64 atomic per non-empty bucket per block when the block retires.  The global-atomic sink above is bound
// by the atomic rate (24 G/s => 0.5 s per 1e8 reads); LDS atomics are not.
struct SinkHistLds {
    u64* counts;
    u32* tab;
    u64 maskk;
    u32 hasher, hk, k, b, tid;
    static constexpr u32 kLdsDwordsPerWave = 0;
    static constexpr bool kRagged = true;
    static constexpr bool kMarksDirty = true;   // (a tile with an invalid byte: fast path + marks, kmx_scan_kernel.h; launch_hist_uniform sweeps behind every scan)
    static u32 block_lds_dwords(const HistParams& p) { return 1u << p.log2_buckets; }
    __device__ SinkHistLds(const HistParams& p, u32 k_, u32, u32*, u32, u32* block_lds, u32 tid_)
        : counts(p.counts), tab(block_lds), maskk(mask2k(k_)), hasher(p.hasher), hk(p.hk), k(k_), b(p.log2_buckets), tid(tid_) {
        for (u32 j = tid; j < (1u << b); j += 256u) tab[j] = 0;
        __syncthreads();
    }
    __device__ __forceinline__ void block_done(u64, u32, u32) {}
    __device__ __forceinline__ void emit(u64 fw, u64 rc) {
        const u64 canon = fw < rc ? fw : rc;
        u64 h;
        if (hasher == KMX_HASH_LEX) h = (hk == k) ? (maskk ^ fw ^ rc ^ canon) : lex_hash(canon, hk);
        else h = canon;
        atomicAdd(&tab[(u32)bucket_of(h, b)], 1u);
    }
    __device__ __forceinline__ void fast(u32, u64 fw, u64 rc) { emit(fw, rc); }
    __device__ __forceinline__ void slow(u32, u64 fw, u64 rc) { emit(fw, rc); }
    __device__ __forceinline__ void begin_read(u64) {}
    __device__ __forceinline__ void slow_block(u32) {}   // a rolled tile: the wave has completed 16 more windows per read
    __device__ __forceinline__ void tile_slow_begin(u64 read) { begin_read(read); }
    __device__ __forceinline__ void tile_slow_emit(u32 pos, u64 fw, u64 rc) { slow(pos, fw, rc); }
    __device__ __forceinline__ void tile_slow_end() { end_read(); }
    __device__ __forceinline__ void end_read() {}
    __device__ __forceinline__ void tile_fast_done(u32) {}
    __device__ __forceinline__ void finish(const HistParams&) {
        __syncthreads();
        for (u32 j = tid; j < (1u << b); j += 256u) {
            const u32 c = tab[j];
            if (c) atomicAdd((unsigned long long*)&counts[j], (unsigned long long)c);
        }
    }
};

// Histogram, 2^15..2^22 buckets, pass 1 of 2: scatter the bucket ids into 64 partitions (the top 6 bits of the bucket).
// Every wave owns a private segment of every partition's stream, so no global cursor and no global atomic is
// involved.  The low b-6 bits of an id are staged in a ROW-entry ring per partition in the wave's LDS slice: ONE
// ds_add_rtn_u32 on a packed {entries appended : 16 | entries written out : 16} word returns the slot and tells
// whether the ring has room, one ds_write_b16 stores the id.  After every block of 16 windows the rings holding
// >= ROW/2 ids write one half row each to their segment, four partitions at a time (one per quarter-wave).
// Pass 2 (hist_part_reduce_kernel) builds each partition's 2^(b-6)-bucket table in LDS.  Ids that find their ring or
// their segment full (adversarial input: everything in one partition) go straight to the global table, so the result
// is exact for every input.
// buckets this block counts; both halves read the whole id stream of the partition.
template <int THREADS, int SUB_BITS = 0>
__global__ void __launch_bounds__(THREADS)
hist_part_reduce_kernel(const uint16_t* __restrict__ stream, const u32* __restrict__ seg_len, u32 cap, u32 n_waves,
                        u32 log2_buckets, u64* __restrict__ counts, u64 top_stream_stride, u32 top_len_stride) {
    // blockIdx.z = (top partition T << SUB_BITS) | sub.  One level (2^15..2^22 buckets): T = 0.  Two levels (2^23..2^28): T = the
    // first-level partition whose 64 second-level streams this launch counts; log2_buckets is then the bucket bits BELOW
    // the first level (b - 6), the streams and lengths of T start T strides in, its counters at counts + (T << log2_buckets).
    extern __shared__ __attribute__((aligned(16))) u32 tab[];
    const u32 lowbits = log2_buckets - 6u, idm = (1u << lowbits) - 1u;
    const u32 tb = lowbits - (u32)SUB_BITS, nb = 1u << tb, sub = SUB_BITS ? (blockIdx.z & ((1u << SUB_BITS) - 1u)) : 0u;
    const u32 top = blockIdx.z >> SUB_BITS;
    stream += (u64)top * top_stream_stride;
    seg_len += (u64)top * top_len_stride;
    counts += (u64)top << log2_buckets;
    const u32 q = blockIdx.x;
    for (u32 j = threadIdx.x; j < nb; j += THREADS) tab[j] = 0;
    __syncthreads();
    // every WAVE of the block walks its own segments (w == its index mod the waves of the partition's blocks): a segment is
    // ~25 KB, too short for 512 threads to keep several loads each in flight
    const u32 wv = threadIdx.x >> 6, ln = threadIdx.x & 63u, nwv = THREADS / 64u;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    // (an id carries the low bits of its partition above bit lowbits: the scan writes the bucket's low 16 bits as they are)
    auto count1 = [&](u32 id) {
        id &= idm;
        if constexpr (SUB_BITS == 0) atomicAdd(&tab[id], 1u);
        else if ((id >> tb) == sub) atomicAdd(&tab[id & (nb - 1u)], 1u);
    };
    auto count8 = [&](const u32x4 v) {
        count1(v.x); count1(v.x >> 16); count1(v.y); count1(v.y >> 16);
        count1(v.z); count1(v.z >> 16); count1(v.w); count1(v.w >> 16);
    };
    const u32 w0 = blockIdx.y * nwv + wv, wstep = gridDim.y * nwv;
    u32 len_next = w0 < n_waves ? seg_len[(u64)w0 * 64u + q] : 0u;
    for (u32 w = w0; w < n_waves; w += wstep) {
        const u32 len = len_next;
        if (w + wstep < n_waves) len_next = seg_len[(u64)(w + wstep) * 64u + q];   // (one segment ahead)
        const uint16_t* __restrict__ sp = stream + ((u64)w * 64u + q) * (u64)cap;   // cap is a multiple of 64: 128-byte aligned
        const u32x4* __restrict__ sp8 = reinterpret_cast<const u32x4*>(sp);
        // four 16-byte loads per lane in flight: with one load per thread the pass ran at the latency of its loads (3.8 TB/s
        // of ids, the LDS 39 % busy)
        const u32 n16 = len / 8u;
        u32 i = ln;
        for (; i + 192u < n16; i += 256u) {
            const u32x4 v0 = __builtin_nontemporal_load(sp8 + i);
            const u32x4 v1 = __builtin_nontemporal_load(sp8 + i + 64u);
            const u32x4 v2 = __builtin_nontemporal_load(sp8 + i + 128u);
            const u32x4 v3 = __builtin_nontemporal_load(sp8 + i + 192u);
            count8(v0);
            count8(v1);
            count8(v2);
            count8(v3);
        }
        for (; i < n16; i += 64u) count8(__builtin_nontemporal_load(sp8 + i));
        for (u32 j = (len & ~7u) + ln; j < len; j += 64u) count1(sp[j]);
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < nb; j += THREADS) {
        const u32 c = tab[j];
        if (c) atomicAdd((unsigned long long*)&counts[((u64)q << lowbits) | ((u64)sub << tb) | j], (unsigned long long)c);
    }
}

// 2^23..2^28 buckets, the pass between: the u32 stream of first-level partition T (blockIdx.y; the ids of all waves of pass
// 1) is split by the NEXT six bits of the bucket into 64 uint16_t streams -- through the same rings and rows as pass 1 (the
// sink is reused as it is: an id is pushed as the 32-bit mix whose top bits are the bucket below the first level).
// Every wave owns a segment of each of T's 64 second-level streams: stream2[T][wave][64][cap2], seg_len2[T][wave][64].
__global__ void __launch_bounds__(256)
hist_repartition_kernel(const u32* __restrict__ stream1, const u32* __restrict__ seg_len1, u32 cap1, u32 n_waves1, u32 log2_buckets,
                        u64* __restrict__ counts, uint16_t* __restrict__ stream2, u32* __restrict__ seg_len2, u32 cap2) {
    typedef SinkHistPartT<1, uint16_t> Sink;
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 T = blockIdx.y, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32 n_waves2 = gridDim.x * 4u;
    const u32 low1 = log2_buckets - 6u;                       // bucket bits below the first level (17..22)
    HistPartParams p2{counts + ((u64)T << low1), KMX_HASH_IDENTITY, 0u, low1,
                      stream2 + (u64)T * n_waves2 * 64u * (u64)cap2, seg_len2 + (u64)T * n_waves2 * 64u, cap2};
    // LDS: the four waves' rings first (32 KB, 8 KB-aligned as the sink wants them), then each wave's words
    u32* const rings = lds;
    u32* const mine = lds + Sink::kRingDwords + wv * Sink::kLdsDwordsPerWave;
    Sink sink(p2, 0u, 0u, mine, lane, rings, threadIdx.x);
    const u32 up = 32u - low1;                                // id -> mix: the bucket below the first level in the top bits
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32 w0 = blockIdx.x * 4u + wv;
    for (u32 w = w0; w < n_waves1; w += n_waves2) {
        const u32 len = seg_len1[(u64)w * 64u + T];
        const u32* __restrict__ sp = stream1 + ((u64)w * 64u + T) * (u64)cap1;
        const u32x4* __restrict__ sp4 = reinterpret_cast<const u32x4*>(sp);
        const u32 n16 = len / 4u;                             // 16-byte pieces
        u32 i = 0;
        // 16 ids per lane and round (four loads in flight), then the rings holding a half row write it out -- the rhythm of
        // pass 1 (16 windows per read, then flush_rows)
        for (; i + 256u <= n16; i += 256u) {
            const u32x4 v0 = __builtin_nontemporal_load(sp4 + i + lane);
            const u32x4 v1 = __builtin_nontemporal_load(sp4 + i + 64u + lane);
            const u32x4 v2 = __builtin_nontemporal_load(sp4 + i + 128u + lane);
            const u32x4 v3 = __builtin_nontemporal_load(sp4 + i + 192u + lane);
            sink.push_mix(0, v0.x << up); sink.push_mix(1, v0.y << up); sink.push_mix(2, v0.z << up); sink.push_mix(3, v0.w << up);
            sink.push_mix(4, v1.x << up); sink.push_mix(5, v1.y << up); sink.push_mix(6, v1.z << up); sink.push_mix(7, v1.w << up);
            sink.push_mix(8, v2.x << up); sink.push_mix(9, v2.y << up); sink.push_mix(10, v2.z << up); sink.push_mix(11, v2.w << up);
            sink.push_mix(12, v3.x << up); sink.push_mix(13, v3.y << up); sink.push_mix(14, v3.z << up); sink.push_mix(15, v3.w << up);
            sink.flush_rows();
        }
        // the tail of the segment: id by id, a quarter of a round at a time
        for (u32 j = 4u * i; j < len; j += 256u) {
#pragma unroll
            for (u32 t = 0; t < 4u; ++t) {
                const u32 e = j + 64u * t + lane;
                if (e < len) {
                    const u32 bucket = (sp[e] << up) >> (32u - low1);
                    sink.place(bucket, sink.take_slot(bucket));
                }
            }
            sink.flush_rows();
        }
    }
    sink.finish(p2);
}

hipError_t launch_sweep_hist(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 hasher, u32 hk, u32 log2_buckets, u64* counts,
                             unsigned long long* queue, int n_cu, hipStream_t stream, const u64* offsets);

// Larger tables, or no scratch: device-scope u64 atomics (SinkHist).
// Every scan is followed by the sweep that takes the windows with an invalid byte back out of the counters (round 6: the sinks mark
// the reads of a dirty tile instead of rolling it -- kmx_scan_kernel.h, SinkMarksDirty; a no-op on clean input).
hipError_t launch_hist_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 hasher, u32 hk, u32 log2_buckets,
                               u64* counts, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                               void* (*get_scratch)(void*, size_t), void* user, size_t scratch_budget, const u64* offsets) {
    *handled = offsets ? scan_domain_ragged(bases, L, k) : scan_domain(bases, n_reads, L, k);
    if (!*handled) return hipSuccess;
    const HistParams p{counts, hasher, hk, log2_buckets};
    auto sweep = [&](const uint8_t* b, u64 n, const u64* o) { return launch_sweep_hist(b, n, L, k, hasher, hk, log2_buckets, counts, queue, n_cu, stream, o); };
    auto atomic_scan = [&](const uint8_t* b, u64 n, const u64* o) -> hipError_t {
        hipError_t e = dispatch<SinkHist>(b, n, L, k, p, queue, n_cu, stream, NoPre(), o);
        return e != hipSuccess ? e : sweep(b, n, o);
    };
    if (log2_buckets <= 14u) {
        hipError_t e = dispatch<SinkHistLds>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets);
        return e != hipSuccess ? e : sweep(bases, n_reads, offsets);
    }
    if (log2_buckets >= 23u && log2_buckets <= 28u && get_scratch != nullptr && n_reads >= 4096u) {
        // Two levels of 64 partitions each (round 3; these sizes took device atomics before: 0.5 s per 1e8 reads).  Pass 1 as
        // below but with the whole bucket per id (u32 entries); hist_repartition_kernel splits each partition's stream by the
        // next six bits into uint16_t streams; hist_part_reduce_kernel counts the 64 x 64 streams in LDS tables of 2^(b-12)
        // entries (2^16 at b = 28: two halves).  Scratch per window: 1.5 x (4 + 2) bytes.
        const u32 Lb = offsets ? (L ? L : 256u) : L;
        const u64 W = Lb >= k ? Lb - k + 1u : 1u;
        // scratch per window: 1.5 x (4 + 2) bytes, plus the fixed pads of the segments (<= 319 entries for each of the 64 x waves
        // first-level and 64 x 64 x waves second-level segments, ~0.8 GB on 256 CUs) -- taken off the budget first, as the
        // one-level path does: a request above the budget grows the context's buffer past it, and the next call asks for
        // slightly more again (a re-allocation per call)
        const u64 fixed_max = (u64)n_cu * 16u * 64u * 319u * 4u + 64ull * (u64)((n_cu * 4 + 63) / 64 > 0 ? (n_cu * 4 + 63) / 64 : 1) * 4u * 64u * 319u * 2u + (64u << 20);
        const u64 fixed = fixed_max < scratch_budget / 4u ? fixed_max : scratch_budget / 4u;
        u64 chunk = (scratch_budget - fixed) / (9u * W);
        if (chunk > n_reads) chunk = n_reads;
        chunk &= ~63ull;
        if (chunk >= 4096u) {
            const u32 low1 = log2_buckets - 6u;
            for (u64 first = 0; first < n_reads; first += chunk) {
                const u64 n = n_reads - first < chunk ? n_reads - first : chunk;
                HistPartParams pp{counts, hasher, hk, log2_buckets, nullptr, nullptr, 0};
                u32 n_waves = 0, n_waves2 = 0, cap2 = 0;
                uint16_t* stream2 = nullptr;
                u32* seg_len2 = nullptr;
                const u32 gx2 = (u32)((n_cu * 4 + 63) / 64 > 0 ? (n_cu * 4 + 63) / 64 : 1);   // blocks per first-level partition in the pass between
                auto pre = [&](u64 grid) -> bool {
                    n_waves = (u32)(grid * 4u);
                    const u64 per_seg = (n * W * 3u / 2u) / ((u64)n_waves * 64u) + 256u;
                    pp.cap = (u32)((per_seg + 63u) & ~63ull);
                    if (pp.cap > (1u << 24)) return false;
                    n_waves2 = gx2 * 4u;
                    const u64 per_seg2 = (n * W * 3u / 2u) / (64ull * n_waves2 * 64u) + 256u;
                    cap2 = (u32)((per_seg2 + 63u) & ~63ull);
                    const size_t s1 = (size_t)n_waves * 64u * pp.cap * 4u, l1 = (size_t)n_waves * 64u * 4u;
                    const size_t s2 = (size_t)64u * n_waves2 * 64u * cap2 * 2u, l2 = (size_t)64u * n_waves2 * 64u * 4u;
                    char* buf = static_cast<char*>(get_scratch(user, s1 + l1 + s2 + l2));
                    if (!buf) return false;
                    pp.stream = buf;
                    pp.seg_len = reinterpret_cast<u32*>(buf + s1);
                    stream2 = reinterpret_cast<uint16_t*>(buf + s1 + l1);
                    seg_len2 = reinterpret_cast<u32*>(buf + s1 + l1 + s2);
                    return true;
                };
                if (first != 0) {
                    hipError_t e = hipMemsetAsync(queue, 0, 32 * 128 + 8, stream);   // (the heads, and the count of marked reads behind them)
                    if (e != hipSuccess) return e;
                }
                const uint8_t* cb = offsets ? bases : bases + first * (u64)L;
                const u64* co = offsets ? offsets + first : nullptr;
                hipError_t e = dispatch_part_u32(cb, n, L, k, pp, queue, n_cu, stream, make_hist_pre(pre), co);
                if (e == hipErrorOutOfMemory) {   // no scratch: the atomic sink handles the rest
                    (void)hipGetLastError();
                    return atomic_scan(cb, n_reads - first, co);
                }
                if (e != hipSuccess) return e;
                e = sweep(cb, n, co);
                if (e != hipSuccess) return e;
                typedef SinkHistPartT<1, uint16_t> Sink2;
                const HistPartParams dummy{};
                const size_t lds2 = ((size_t)Sink2::block_lds_dwords(dummy) + 4u * Sink2::kLdsDwordsPerWave) * 4u;
                hipLaunchKernelGGL(hist_repartition_kernel, dim3(gx2, 64), dim3(256), lds2, stream, static_cast<const u32*>(pp.stream), pp.seg_len,
                                   pp.cap, n_waves, log2_buckets, counts, stream2, seg_len2, cap2);
                e = hipGetLastError();
                if (e != hipSuccess) return e;
                const bool halves = low1 == 22u;   // 2^16 buckets per second-level partition: two blocks of 2^15 each
                const u32 nb_bytes = 4u << (low1 - 6u - (halves ? 1u : 0u));
                auto red = halves ? hist_part_reduce_kernel<512, 1> : hist_part_reduce_kernel<512, 0>;
                if (nb_bytes > 64u * 1024u) {
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(red), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_bytes);
                    if (e != hipSuccess) return e;
                }
                const u32 groups = n_waves2 < 4u ? n_waves2 : 4u;
                hipLaunchKernelGGL(red, dim3(64, groups, 64u * (halves ? 2u : 1u)), dim3(512), nb_bytes, stream, stream2, seg_len2, cap2, n_waves2,
                                   low1, counts, (u64)n_waves2 * 64u * (u64)cap2, n_waves2 * 64u);
                e = hipGetLastError();
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
    }
    if (log2_buckets <= 22u && get_scratch != nullptr && n_reads >= 4096u) {
        // windows per read the segments are sized for.  Ragged reads: from the caller's bound of the lengths (the frame's 256 if
        // there is none); a read that is longer after all only fills its wave's segments sooner, and what finds a segment full
        // goes to the global table (exact, slow).
        const u32 Lb = offsets ? (L ? L : 256u) : L;
        const u64 W = Lb >= k ? Lb - k + 1u : 1u;
        // scratch per read: 1.5x slack on 2 bytes per window, plus the fixed per-segment pad (at most 319 entries for each of the 64
        // segments of at most 8 blocks per CU of four waves, and their lengths); chunk the reads so that a request never EXCEEDS
        // the budget -- a request above it made the context's buffer larger than the budget, the next call was handed that size as
        // its budget (kmx_api.hip), asked for a little more again, and every call re-allocated a 36 GB buffer (1.1 s; round 3)
        const u64 fixed_max = (u64)n_cu * 8u * 4u * 64u * (319u * 2u + 4u);
        const u64 fixed = fixed_max < scratch_budget / 4u ? fixed_max : scratch_budget / 4u;
        u64 chunk = (scratch_budget - fixed) / (3u * W);
        if (chunk > n_reads) chunk = n_reads;
        chunk &= ~63ull;
        if (chunk >= 4096u) {
            for (u64 first = 0; first < n_reads; first += chunk) {
                const u64 n = n_reads - first < chunk ? n_reads - first : chunk;
                HistPartParams pp{counts, hasher, hk, log2_buckets, nullptr, nullptr, 0};
                u32 n_waves = 0;
                auto pre = [&](u64 grid) -> bool {
                    n_waves = (u32)(grid * 4u);
                    const u64 per_seg = (n * W * 3u / 2u) / ((u64)n_waves * 64u) + 256u;
                    pp.cap = (u32)((per_seg + 63u) & ~63ull);
                    if (pp.cap > (1u << 24)) return false;   // 64 * cap must stay below 2^31 (SinkHistPart::dest)
                    const size_t stream_bytes = (size_t)n_waves * 64u * pp.cap * 2u;
                    const size_t len_bytes = (size_t)n_waves * 64u * 4u;
                    char* buf = static_cast<char*>(get_scratch(user, stream_bytes + len_bytes));
                    if (!buf) return false;
                    pp.stream = buf;
                    pp.seg_len = reinterpret_cast<u32*>(buf + stream_bytes);
                    return true;
                };
                if (first != 0) {
                    hipError_t e = hipMemsetAsync(queue, 0, 32 * 128 + 8, stream);   // (the heads, and the count of marked reads behind them)
                    if (e != hipSuccess) return e;
                }
                // the hook fills pp through the reference captured above; dispatch takes its params by value, so hand it
                // a proxy that copies the finished pp at launch time
                // (ragged reads: the offsets are absolute, the chunk is a window into them)
                const uint8_t* cb = offsets ? bases : bases + first * (u64)L;
                const u64* co = offsets ? offsets + first : nullptr;
                hipError_t e = dispatch_part<HistPartPre>(cb, n, L, k, pp, queue, n_cu, stream, make_hist_pre(pre), co);
                if (e == hipErrorOutOfMemory) {   // no scratch: the atomic sink handles the rest
                    (void)hipGetLastError();
                    return atomic_scan(cb, n_reads - first, co);
                }
                if (e != hipSuccess) return e;
                e = sweep(cb, n, co);
                if (e != hipSuccess) return e;
                const bool halves = log2_buckets == 22u;   // 2^16 buckets per partition: two blocks of 2^15 each
                const u32 nb_bytes = 4u << (log2_buckets - 6u - (halves ? 1u : 0u));
                auto red = halves ? hist_part_reduce_kernel<512, 1> : hist_part_reduce_kernel<512, 0>;
                if (nb_bytes > 64u * 1024u) {
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(red), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_bytes);
                    if (e != hipSuccess) return e;
                }
                // (eight blocks per partition: 16 measured the same at 1.25e8 reads and 35 us slower per call up to 4e6 -- every block flushes a whole
                // table --, one or two slower again: profiles/r05_small_batches.txt)
                const u32 groups = n_waves < 8u ? n_waves : 8u;
                hipLaunchKernelGGL(red, dim3(64, groups, halves ? 2 : 1), dim3(512), nb_bytes, stream, static_cast<const uint16_t*>(pp.stream), pp.seg_len,
                                   pp.cap, n_waves, log2_buckets, counts, (u64)0, 0u);
                e = hipGetLastError();
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
    }
    return atomic_scan(bases, n_reads, offsets);
}

}  // namespace kmx
