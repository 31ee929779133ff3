// kmx_segments.hip -- long ragged reads (round 4): reads behind an offsets array that do not fit a frame of the tiled scan
// (PacBio / ONT reads, contigs: thousands of bases, every length different) used to roll one lane per read (~0.4 TB/s).  Here
// each read is cut into overlapping SEGMENTS of at most `t_max` windows -- read r of W_r = len_r - k + 1 windows into
// c_r = ceil(W_r / t_max) segments of T_r = ceil(W_r / c_r) windows (the last one shorter), segment j = bases
// [j T_r, j T_r + T_r + k - 1) of the read: every window belongs to exactly one segment, neighbours share k - 1 bases.  The
// segments' starts and ends go to two arrays and the ragged bit-sliced kernel scans them as reads of their own
// (scan_bitsliced_kernel<.., RAGGED>: `offsets` = starts, `ends` = ends).  A window depends on its own k bases only, so the
// summaries are the reads' (canonical_kmer_iterator.rs:42-70 per read; the sums and the xor fold are over windows).
//
// Three small kernels: per-block segment counts; an exclusive scan of the block counts (one block); the fill (block-local scan +
// the block's base, the block's threads writing its run of the arrays together).  The arrays are sized from an upper bound
// S_max >= S (the host knows the total number of bases); the host reads S back and scans exactly S segments.
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

// The block's 1024 reads own a contiguous run of the output, [block_base, block_base + block total).  The threads fill it
// TOGETHER, one segment per thread and step (coalesced 8-byte stores): which read a segment belongs to is a binary search over
// the block's per-read first-segment indices in LDS.  (One thread writing all segments of "its" reads took 0.5 ms for 2e5 reads
// of 30 kbp: ~230 scattered stores per read, in series.)  A block whose reads are long has hundreds of thousands of segments to
// write: gridDim.y blocks share the run (each repeats the cheap block-local scan and fills its slice).
// (materialise: `win_offsets` != nullptr -- the first output slot of every read -- and `wins` takes every SEGMENT's first slot, the
// slot array the ragged materialise kernels expect: a read's segments write consecutive slots)
__global__ void __launch_bounds__(SEG_THREADS) seg_fill_kernel(const u64* __restrict__ offsets, u64 n_reads, u32 k, u32 t_max,
                                                              const u64* __restrict__ block_base, u64* __restrict__ starts,
                                                              u64* __restrict__ ends, const u64* __restrict__ win_offsets, u64* __restrict__ wins,
                                                              u64 seg_capacity) {
    __shared__ u64 wave_tot[SEG_THREADS / 64];
    __shared__ u32 first[SEG_PER_BLOCK + 1];     // first segment of read i of the block, relative to the block's base
    __shared__ u64 r_o0[SEG_PER_BLOCK];
    __shared__ u64 r_w0[SEG_PER_BLOCK];
    __shared__ u32 r_len[SEG_PER_BLOCK];
    const u64 r0 = ((u64)blockIdx.x * SEG_THREADS + threadIdx.x) * SEG_PER_THREAD;
    u32 c[SEG_PER_THREAD];
    u64 mine = 0;
#pragma unroll
    for (u32 i = 0; i < SEG_PER_THREAD; ++i) {
        const u64 r = r0 + i;
        u64 o0 = 0, len = 0;
        c[i] = 0;
        if (r < n_reads) {
            o0 = offsets[r];
            len = offsets[r + 1u] - o0;
            c[i] = seg_count(len, k, t_max);
        }
        r_o0[threadIdx.x * SEG_PER_THREAD + i] = o0;
        r_w0[threadIdx.x * SEG_PER_THREAD + i] = (win_offsets != nullptr && r < n_reads) ? win_offsets[r] : 0ull;
        r_len[threadIdx.x * SEG_PER_THREAD + i] = c[i] ? (u32)len : 0u;
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
    u64 at = x - mine;
    for (u32 w = 0; w < (threadIdx.x >> 6); ++w) at += wave_tot[w];
#pragma unroll
    for (u32 i = 0; i < SEG_PER_THREAD; ++i) {
        first[threadIdx.x * SEG_PER_THREAD + i] = (u32)at;     // (a block holds fewer than 2^32 segments: 1024 reads of < 2^31 bases)
        at += c[i];
    }
    const u32 block_total = (u32)(wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3]);
    if (threadIdx.x == 0) first[SEG_PER_BLOCK] = block_total;
    __syncthreads();
    const u64 base = block_base[blockIdx.x];
    const u32 slice = (block_total + gridDim.y - 1u) / gridDim.y;
    const u32 s_end = slice * (blockIdx.y + 1u) < block_total ? slice * (blockIdx.y + 1u) : block_total;
    for (u32 s = slice * blockIdx.y + threadIdx.x; s < s_end; s += SEG_THREADS) {
        // the last read i with first[i] <= s (reads without a segment share their successor's index and are skipped over)
        u32 lo = 0, hi = SEG_PER_BLOCK;
        while (hi - lo > 1u) {
            const u32 mid = (lo + hi) >> 1;
            if (first[mid] <= s) lo = mid; else hi = mid;
        }
        const u32 j = s - first[lo];
        const u32 len = r_len[lo], w = len - k + 1u;
        const u32 cnt = first[lo + 1u] - first[lo];
        const u32 t = (w + cnt - 1u) / cnt;          // windows per segment, balanced; the last segment takes what is left
        const u64 o0 = r_o0[lo], st = o0 + (u64)j * t, e = st + t + (k - 1u), read_end = o0 + len;
        // (offsets that do not increase give lengths whose segments outnumber the bound the arrays were sized from: nothing is
        // written past them, and the host, which compares the total with the bound before it scans, fails the call)
        if (base + s >= seg_capacity) continue;
        starts[base + s] = st;
        ends[base + s] = e < read_end ? e : read_end;
        if (wins != nullptr) wins[base + s] = r_w0[lo] + (u64)j * t;
    }
}

}  // namespace

// scratch the segment arrays of a batch need: two u64 per segment of the upper bound + the block counts + the total
size_t segments_scratch_bytes(u64 n_reads, u64 seg_capacity, bool with_wins) {
    const u64 n_blocks = (n_reads + SEG_PER_BLOCK - 1u) / SEG_PER_BLOCK;
    return (size_t)(2u * seg_capacity + n_blocks + 8u + (with_wins ? seg_capacity + 8u : 0u)) * 8u;
}

// S_max for reads of `total_bases` bases in all: read r has at most len_r / t_max + 1 segments
u64 segments_capacity(u64 n_reads, u64 total_bases, u32 t_max) { return ((total_bases / t_max + n_reads + 63u) & ~63ull) + 64u; }

// Builds starts[] / ends[] (room for seg_capacity entries each, inside `scratch`) on `stream`; *total_out (device) = the number of
// segments written.
hipError_t launch_segments_build(const u64* offsets, u64 n_reads, u32 k, u32 t_max, u64 seg_capacity, void* scratch,
                                 const u64** starts_out, const u64** ends_out, const u64** total_out, unsigned long long* too_long,
                                 hipStream_t stream, const u64* win_offsets, const u64** wins_out) {
    const u64 n_blocks = (n_reads + SEG_PER_BLOCK - 1u) / SEG_PER_BLOCK;
    if (n_blocks == 0 || n_blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    u64* starts = static_cast<u64*>(scratch);
    u64* ends = starts + seg_capacity;
    u64* block_sums = ends + seg_capacity;
    u64* total = block_sums + n_blocks;
    u64* wins = win_offsets ? total + 8 : nullptr;      // (materialise: seg_capacity + 1 slots behind everything else)
    hipLaunchKernelGGL(seg_count_kernel, dim3((unsigned)n_blocks), dim3(SEG_THREADS), 0, stream, offsets, n_reads, k, t_max, block_sums, too_long);
    hipLaunchKernelGGL(seg_scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, block_sums, n_blocks, total);
    // (slices of ~4096 segments: by the bound, the average block of 1024 reads holds seg_capacity / n_blocks of them)
    u64 splits = seg_capacity / n_blocks / 4096u;
    splits = splits < 1u ? 1u : splits > 64u ? 64u : splits;
    hipLaunchKernelGGL(seg_fill_kernel, dim3((unsigned)n_blocks, (unsigned)splits), dim3(SEG_THREADS), 0, stream, offsets, n_reads, k, t_max, block_sums, starts, ends, win_offsets, wins, seg_capacity);
    *starts_out = starts;
    *ends_out = ends;
    *total_out = total;
    if (wins_out) *wins_out = wins;
    return hipGetLastError();
}

// ---- uniform reads longer than a frame, materialise (round 4): read r of L bases (W = L - k + 1 windows) as J = ceil(W / T) segments
// of T windows (the last one what is left), each a "read" of its own for the ragged materialise kernels: segment g = r J + j
// starts at byte r L + j T, ends k - 1 bases behind its last window's start, and its windows go to the output slots
// r W + j T ... -- consecutive segments, consecutive slots: `wins` is the exclusive prefix sum the kernels expect.
__global__ void __launch_bounds__(256) seg_plan_uniform_kernel(u64 n_seg, u32 L, u32 k, u32 T, u32 J, u64* __restrict__ starts, u64* __restrict__ ends,
                                                               u64* __restrict__ wins) {
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    const u32 W = L - k + 1u;
    if (g < n_seg) {
        const u64 r = g / J;
        const u32 j = (u32)(g - r * J);
        const u32 nw = W - j * T < T ? W - j * T : T;
        starts[g] = r * (u64)L + (u64)j * T;
        ends[g] = r * (u64)L + (u64)j * T + nw + (k - 1u);
        wins[g] = r * (u64)W + (u64)j * T;
    }
    if (g == n_seg) wins[g] = (n_seg / J) * (u64)W;
}

size_t uniform_segments_scratch_bytes(u64 n_seg) { return (size_t)(3u * n_seg + 8u) * 8u; }

hipError_t launch_uniform_segments_plan(u64 n_reads, u32 L, u32 k, u32 T, void* scratch, const u64** starts, const u64** ends, const u64** wins, u64* n_seg_out,
                                        hipStream_t stream) {
    const u32 W = L - k + 1u, J = (W + T - 1u) / T;
    const u64 n_seg = n_reads * J;
    u64* a = static_cast<u64*>(scratch);
    hipLaunchKernelGGL(seg_plan_uniform_kernel, dim3((unsigned)((n_seg + 256u) / 256u)), dim3(256), 0, stream, n_seg, L, k, T, J, a, a + n_seg, a + 2u * n_seg);
    *starts = a;
    *ends = a + n_seg;
    *wins = a + 2u * n_seg;
    *n_seg_out = n_seg;
    return hipGetLastError();
}

}  // namespace kmx
