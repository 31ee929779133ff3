// kmx_segments.hip -- long ragged reads (round 4): reads behind an offsets array that do not fit a frame of the tiled scan
// (PacBio / ONT reads, contigs: thousands of bases, every length different) used to roll one lane per read (~0.4 TB/s).  Here
// each read is cut into overlapping SEGMENTS of at most `t_max` windows -- read r of W_r = len_r - k + 1 windows into
// c_r = ceil(W_r / t_max) segments of T_r = ceil(W_r / c_r) windows (the last one shorter), segment j = bases
// [j T_r, j T_r + T_r + k - 1) of the read: every window belongs to exactly one segment, neighbours share k - 1 bases.  The
// segments' starts and ends go to two arrays and the ragged bit-sliced kernel scans them as reads of their own
// (scan_bitsliced_kernel<.., RAGGED>: `offsets` = starts, `ends` = ends).  A window depends on its own k bases only, so the
// summaries are the reads' (canonical_kmer_iterator.rs:42-70 per read; the sums and the xor fold are over windows).
//
// Three small kernels, no host round trip between them: per-block segment counts; an exclusive scan of the block counts (one
// block); the fill (block-local scan + the block's base).  The arrays are sized from an upper bound S_max >= S (the host knows
// the total number of bases); the entries past the last segment are empty segments at the end of the buffer.
#include "kmx_device.h"

namespace kmx {

namespace {

constexpr u32 SEG_THREADS = 256, SEG_PER_THREAD = 4, SEG_PER_BLOCK = SEG_THREADS * SEG_PER_THREAD;

__device__ __forceinline__ u32 seg_count(u64 len, u32 k, u32 t_max) {
    if (len < k) return 0u;
    if (len > 0x7FFFFFFFull) return 0u;   // (a read of 2^31 bases or more is skipped by every scan and reported: kmx.h "Limits")
    const u32 w = (u32)len - k + 1u;
    return (w + t_max - 1u) / t_max;
}

__global__ void __launch_bounds__(SEG_THREADS) seg_count_kernel(const u64* __restrict__ offsets, u64 n_reads, u32 k, u32 t_max,
                                                               u64* __restrict__ block_sums, unsigned long long* __restrict__ too_long) {
    __shared__ u64 part[SEG_THREADS / 64];
    const u64 r0 = ((u64)blockIdx.x * SEG_THREADS + threadIdx.x) * SEG_PER_THREAD;
    u64 c = 0;
#pragma unroll
    for (u32 i = 0; i < SEG_PER_THREAD; ++i) {
        const u64 r = r0 + i;
        if (r < n_reads) {
            const u64 len = offsets[r + 1u] - offsets[r];
            if (len > 0x7FFFFFFFull) *too_long = 1ull;
            c += seg_count(len, k, t_max);
        }
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the block counts, in place, by ONE block (n_blocks <= n_reads / 1024: a few thousand for long reads)
__global__ void __launch_bounds__(1024) seg_scan_blocks_kernel(u64* __restrict__ block_sums, u64 n_blocks, u64* __restrict__ total) {
    __shared__ u64 warp_tot[16];
    __shared__ u64 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u64 base = 0; base < n_blocks; base += 1024u) {
        const u64 i = base + threadIdx.x;
        const u64 v = i < n_blocks ? block_sums[i] : 0ull;
        // inclusive scan within the wave (shuffles), then across the 16 waves
        u64 x = v;
#pragma unroll
        for (u32 d = 1; d < 64u; d <<= 1) {
            const u64 y = __shfl_up(x, d, WAVE);
            if ((threadIdx.x & 63u) >= d) x += y;
        }
        if ((threadIdx.x & 63u) == 63u) warp_tot[threadIdx.x >> 6] = x;
        __syncthreads();
        u64 before = carry_s;
        for (u32 w = 0; w < (threadIdx.x >> 6); ++w) before += warp_tot[w];
        if (i < n_blocks) block_sums[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023u) carry_s = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

__global__ void __launch_bounds__(SEG_THREADS) seg_fill_kernel(const u64* __restrict__ offsets, u64 n_reads, u32 k, u32 t_max,
                                                              const u64* __restrict__ block_base, u64* __restrict__ starts,
                                                              u64* __restrict__ ends) {
    __shared__ u64 wave_tot[SEG_THREADS / 64];
    const u64 r0 = ((u64)blockIdx.x * SEG_THREADS + threadIdx.x) * SEG_PER_THREAD;
    u32 c[SEG_PER_THREAD];
    u64 o0[SEG_PER_THREAD], len[SEG_PER_THREAD];
    u64 mine = 0;
#pragma unroll
    for (u32 i = 0; i < SEG_PER_THREAD; ++i) {
        const u64 r = r0 + i;
        c[i] = 0;
        o0[i] = 0;
        len[i] = 0;
        if (r < n_reads) {
            o0[i] = offsets[r];
            len[i] = offsets[r + 1u] - o0[i];
            c[i] = seg_count(len[i], k, t_max);
        }
        mine += c[i];
    }
    u64 x = mine;   // inclusive scan over the block's threads
#pragma unroll
    for (u32 d = 1; d < 64u; d <<= 1) {
        const u64 y = __shfl_up(x, d, WAVE);
        if ((threadIdx.x & 63u) >= d) x += y;
    }
    if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = x;
    __syncthreads();
    u64 at = block_base[blockIdx.x] + x - mine;
    for (u32 w = 0; w < (threadIdx.x >> 6); ++w) at += wave_tot[w];
#pragma unroll
    for (u32 i = 0; i < SEG_PER_THREAD; ++i) {
        if (c[i] == 0u) continue;
        const u32 w = (u32)len[i] - k + 1u;
        const u32 t = (w + c[i] - 1u) / c[i];          // windows per segment, balanced; the last segment takes what is left
        const u64 read_end = o0[i] + len[i];
        for (u32 j = 0; j < c[i]; ++j) {
            const u64 s = o0[i] + (u64)j * t;
            const u64 e = s + t + (k - 1u);
            starts[at + j] = s;
            ends[at + j] = e < read_end ? e : read_end;
        }
        at += c[i];
    }
}

// the entries between the last segment and the array's capacity: empty segments at the end of the buffer
__global__ void seg_pad_kernel(const u64* __restrict__ total, u64 capacity, u64 end_of_bases, u64* __restrict__ starts, u64* __restrict__ ends) {
    const u64 first = *total;
    for (u64 i = first + (u64)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += (u64)gridDim.x * blockDim.x) {
        starts[i] = end_of_bases;
        ends[i] = end_of_bases;
    }
}

}  // namespace

// scratch the segment arrays of a batch need: two u64 per segment of the upper bound + the block counts + the total
size_t segments_scratch_bytes(u64 n_reads, u64 seg_capacity) {
    const u64 n_blocks = (n_reads + SEG_PER_BLOCK - 1u) / SEG_PER_BLOCK;
    return (size_t)(2u * seg_capacity + n_blocks + 8u) * 8u;
}

// S_max for reads of `total_bases` bases in all: read r has at most len_r / t_max + 1 segments
u64 segments_capacity(u64 n_reads, u64 total_bases, u32 t_max) { return ((total_bases / t_max + n_reads + 63u) & ~63ull) + 64u; }

// Builds starts[] / ends[] (seg_capacity entries each, inside `scratch`) on `stream`.  `end_of_bases` = offsets[n_reads].
hipError_t launch_segments_build(const u64* offsets, u64 n_reads, u32 k, u32 t_max, u64 seg_capacity, u64 end_of_bases, void* scratch,
                                 const u64** starts_out, const u64** ends_out, unsigned long long* too_long, hipStream_t stream) {
    const u64 n_blocks = (n_reads + SEG_PER_BLOCK - 1u) / SEG_PER_BLOCK;
    if (n_blocks == 0 || n_blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    u64* starts = static_cast<u64*>(scratch);
    u64* ends = starts + seg_capacity;
    u64* block_sums = ends + seg_capacity;
    u64* total = block_sums + n_blocks;
    hipLaunchKernelGGL(seg_count_kernel, dim3((unsigned)n_blocks), dim3(SEG_THREADS), 0, stream, offsets, n_reads, k, t_max, block_sums, too_long);
    hipLaunchKernelGGL(seg_scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, block_sums, n_blocks, total);
    hipLaunchKernelGGL(seg_fill_kernel, dim3((unsigned)n_blocks), dim3(SEG_THREADS), 0, stream, offsets, n_reads, k, t_max, block_sums, starts, ends);
    hipLaunchKernelGGL(seg_pad_kernel, dim3(64), dim3(256), 0, stream, total, seg_capacity, end_of_bases, starts, ends);
    *starts_out = starts;
    *ends_out = ends;
    return hipGetLastError();
}

}  // namespace kmx
