// kmx_generic.hip -- reference-shaped kernels for everything outside the fast uniform scan:
// ragged reads (offsets), k in {1,17}, mis-aligned buffers, per-window materialisation,
// the [u64;2] (k in 33..64) extension and the bucket histogram.  One lane walks one read
// exactly like CanonicalKmerIterator::find_next (src/naive_impl/canonical_kmer_iterator.rs:42-70);
// correctness-first, these are not the roofline kernels.
#include "kmx_device.h"

namespace kmx {

struct ReadsView {
    const uint8_t* bases;
    u64 n_reads;
    u32 read_len;
    const u64* offsets;
    __device__ __forceinline__ void span(u64 r, const uint8_t*& s, u32& len) const {
        if (offsets) {
            const u64 a = offsets[r], b = offsets[r + 1];
            s = bases + a;
            len = (u32)(b - a);
        } else {
            s = bases + r * (u64)read_len;
            len = read_len;
        }
    }
};

__device__ __forceinline__ u64 hash_word(u64 canon, u32 hasher, u32 hk) {
    return hasher == KMX_HASH_LEX ? lex_hash(canon, hk) : canon;  // identity: write_u64(data), hash.rs:4-8
}

// ---- reduce, any layout, k in [1,31]
__global__ void __launch_bounds__(256)
reduce_generic_kernel(ReadsView rv, u32 k, u32 hasher, u32 hk, u32 want_sumfw, kmx_summary* __restrict__ out) {
    Acc acc;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < rv.n_reads; r += stride) {
        const uint8_t* s;
        u32 len;
        rv.span(r, s, len);
        roll_read(s, len, k, [&](u32, u64 fw, u64 rc) {
            const u64 canon = fw < rc ? fw : rc;
            acc.n_valid += 1;
            acc.sum_canon += canon;
            if (hasher != KMX_HASH_NONE) acc.xor_hash ^= hash_word(canon, hasher, hk);
            if (want_sumfw) acc.sum_fw += fw;
        });
    }
    flush_acc(acc, out, hasher != KMX_HASH_NONE, want_sumfw != 0);
}

// ---- materialise per-window state, any layout, k in [1,31]
__global__ void __launch_bounds__(256)
windows_generic_kernel(ReadsView rv, const u64* __restrict__ win_offsets, u32 k, u64* __restrict__ o_fw,
                       u64* __restrict__ o_rc, u64* __restrict__ o_canon, uint8_t* __restrict__ o_flags) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < rv.n_reads; r += stride) {
        const uint8_t* s;
        u32 len;
        rv.span(r, s, len);
        const u32 nwin = len >= k ? len - k + 1u : 0u;
        const u64 base = win_offsets ? win_offsets[r] : r * (u64)(rv.read_len >= k ? rv.read_len - k + 1u : 0u);
        u32 next = 0;  // first slot not yet written
        auto zero_to = [&](u32 end) {
            for (; next < end; ++next) {
                if (o_fw) o_fw[base + next] = 0;
                if (o_rc) o_rc[base + next] = 0;
                if (o_canon) o_canon[base + next] = 0;
                if (o_flags) o_flags[base + next] = 0;
            }
        };
        roll_read(s, len, k, [&](u32 pos, u64 fw, u64 rc) {
            zero_to(pos);
            const bool lt = fw < rc;
            if (o_fw) o_fw[base + pos] = fw;
            if (o_rc) o_rc[base + pos] = rc;
            if (o_canon) o_canon[base + pos] = lt ? fw : rc;
            if (o_flags) o_flags[base + pos] = (uint8_t)(KMX_WIN_VALID | (lt ? KMX_WIN_FW_CANONICAL : 0u));
            next = pos + 1u;
        });
        zero_to(nwin);
    }
}

// ---- bucket histogram, any layout
__global__ void __launch_bounds__(256)
histogram_generic_kernel(ReadsView rv, u32 k, u32 hasher, u32 hk, u32 log2_buckets, u64* __restrict__ counts) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < rv.n_reads; r += stride) {
        const uint8_t* s;
        u32 len;
        rv.span(r, s, len);
        roll_read(s, len, k, [&](u32, u64 fw, u64 rc) {
            const u64 canon = fw < rc ? fw : rc;
            const u64 h = hasher == KMX_HASH_NONE ? canon : hash_word(canon, hasher, hk);
            atomicAdd((unsigned long long*)&counts[bucket_of(h, log2_buckets)], 1ull);
        });
    }
}

// ---------------------------------------------------------------- [u64;2] k-mers (U128, roll_read2: kmx_device.h)
__global__ void __launch_bounds__(256)
reduce2_generic_kernel(ReadsView rv, u32 k, u32 with_hash, kmx_summary2* __restrict__ out) {
    u64 n = 0, slo = 0, shi = 0, xlo = 0, xhi = 0;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < rv.n_reads; r += stride) {
        const uint8_t* s;
        u32 len;
        rv.span(r, s, len);
        roll_read2(s, len, k, [&](u32, U128 fw, U128 rc) {
            const U128 c = lt128(fw, rc) ? fw : rc;
            n += 1;
            slo += c.lo;
            shi += c.hi;
            if (with_hash) {
                const U128 h = lex_hash128(c, k);
                xlo ^= h.lo;
                xhi ^= h.hi;
            }
        });
    }
    n = wave_sum(n);
    slo = wave_sum(slo);
    shi = wave_sum(shi);
    xlo = wave_xor(xlo);
    xhi = wave_xor(xhi);
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd((unsigned long long*)&out->n_valid, (unsigned long long)n);
        atomicAdd((unsigned long long*)&out->sum_lo, (unsigned long long)slo);
        atomicAdd((unsigned long long*)&out->sum_hi, (unsigned long long)shi);
        atomicXor((unsigned long long*)&out->xor_lo, (unsigned long long)xlo);
        atomicXor((unsigned long long*)&out->xor_hi, (unsigned long long)xhi);
    }
}

__global__ void __launch_bounds__(256)
windows2_generic_kernel(ReadsView rv, const u64* __restrict__ win_offsets, u32 k, u64* __restrict__ o_fw,
                        u64* __restrict__ o_rc, u64* __restrict__ o_canon, uint8_t* __restrict__ o_flags) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < rv.n_reads; r += stride) {
        const uint8_t* s;
        u32 len;
        rv.span(r, s, len);
        const u32 nwin = len >= k ? len - k + 1u : 0u;
        const u64 base = win_offsets ? win_offsets[r] : r * (u64)(rv.read_len >= k ? rv.read_len - k + 1u : 0u);
        u32 next = 0;
        auto zero_to = [&](u32 end) {
            for (; next < end; ++next) {
                const u64 j = 2u * (base + next);
                if (o_fw) o_fw[j] = o_fw[j + 1] = 0;
                if (o_rc) o_rc[j] = o_rc[j + 1] = 0;
                if (o_canon) o_canon[j] = o_canon[j + 1] = 0;
                if (o_flags) o_flags[base + next] = 0;
            }
        };
        roll_read2(s, len, k, [&](u32 pos, U128 fw, U128 rc) {
            zero_to(pos);
            const bool lt = lt128(fw, rc);
            const U128 c = lt ? fw : rc;
            const u64 j = 2u * (base + pos);
            if (o_fw) { o_fw[j] = fw.lo; o_fw[j + 1] = fw.hi; }
            if (o_rc) { o_rc[j] = rc.lo; o_rc[j + 1] = rc.hi; }
            if (o_canon) { o_canon[j] = c.lo; o_canon[j + 1] = c.hi; }
            if (o_flags) o_flags[base + pos] = (uint8_t)(KMX_WIN_VALID | (lt ? KMX_WIN_FW_CANONICAL : 0u));
            next = pos + 1u;
        });
        zero_to(nwin);
    }
}

// ------------------------------------------------------------------ launchers

static inline unsigned grid_for(u64 n, int n_cu) {
    u64 g = (n + 255u) / 256u;
    const u64 cap = (u64)n_cu * 8u;
    if (g > cap) g = cap;
    return (unsigned)(g ? g : 1);
}

hipError_t launch_reduce_generic(const kmx_reads* r, u32 k, u32 hasher, u32 hk, u32 want_sumfw, kmx_summary* out,
                                 int n_cu, hipStream_t st) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets};
    hipLaunchKernelGGL(reduce_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, k, hasher, hk,
                       want_sumfw, out);
    return hipGetLastError();
}

hipError_t launch_windows_generic(const kmx_reads* r, const u64* win_off, u32 k, u64* fw, u64* rc, u64* canon,
                                  uint8_t* flags, int n_cu, hipStream_t st) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets};
    hipLaunchKernelGGL(windows_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, win_off, k, fw,
                       rc, canon, flags);
    return hipGetLastError();
}

hipError_t launch_histogram_generic(const kmx_reads* r, u32 k, u32 hasher, u32 hk, u32 log2_buckets, u64* counts,
                                    int n_cu, hipStream_t st) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets};
    hipLaunchKernelGGL(histogram_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, k, hasher, hk,
                       log2_buckets, counts);
    return hipGetLastError();
}

hipError_t launch_reduce2_generic(const kmx_reads* r, u32 k, u32 with_hash, kmx_summary2* out, int n_cu,
                                  hipStream_t st) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets};
    hipLaunchKernelGGL(reduce2_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, k, with_hash,
                       out);
    return hipGetLastError();
}

hipError_t launch_windows2_generic(const kmx_reads* r, const u64* win_off, u32 k, u64* fw, u64* rc, u64* canon,
                                   uint8_t* flags, int n_cu, hipStream_t st) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets};
    hipLaunchKernelGGL(windows2_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, win_off, k, fw,
                       rc, canon, flags);
    return hipGetLastError();
}

}  // namespace kmx
