// kmx_seqvec.hip -- SeqVector: the 2-bit packed sequence container of the reference
// (src/naive_impl/seq_vector.rs) as batch device operations.
//
// Layout (seq_vector.rs:230-242 builds it from Kmer::from of 32-base chunks; RawVector is LSB-first): base i occupies
// flat bits [2i, 2i+1] of a little-endian u64 word array, codes A0 C1 G2 T3; bits past 2*len are zero.
//   push_chars / From<&[u8]>   seq_vector.rs:141-161, 230-242   -> seqvec_push_kernel (strict: Kmer::from panics on a bad base)
//   get_kmer_u64 / get_base    seq_vector.rs:96-103            -> seqvec_get_kmers_kernel
//   iter_kmers                 seq_vector.rs:117-124, 341-357   -> seqvec_iter_kmers_kernel
//   String::from(&SeqVector)   seq_vector.rs:171-182            -> seqvec_to_bytes_kernel
//   Kmer::minimizer_word       kmer.rs:170-192                  -> minimizer_words_kernel
//   SeqVecMinimizerIter        seq_vector/minimizers.rs:39-141  -> seqvec_minimizers_kernel
// and, for reads stored back to back as L-base slices (SeqVector::slice, :226-234), the canonical k-mer scan of
// every slice for (k, L) outside the bit-sliced kernel: reduce_packed_generic_kernel (one lane walks one read with
// CanonicalKmer::append_base, canonical_kmer.rs:90-94).
#include "kmx_device.h"

#include <type_traits>

// (the per-window stores of the sliding-minimum kernel carry the nt hint (+2 ... +6 %: profiles/r03_nt_stores.txt)
namespace kmx {

// the 2k-bit field at base position pos (RawVector::int(pos*2, k*2)), k in [1,32]; words past the end read as 0
__device__ __forceinline__ u64 seqvec_field(const u64* __restrict__ words, u64 n_words, u64 pos, u32 k) {
    const u64 wi = pos >> 5;
    const u32 sh = 2u * (u32)(pos & 31u);
    const u64 lo = words[wi];
    const u64 hi = (sh != 0u && wi + 1u < n_words) ? words[wi + 1u] : 0ull;
    const u64 v = sh ? ((lo >> sh) | (hi << (64u - sh))) : lo;
    return k >= 32u ? v : (v & ((1ull << (2u * k)) - 1ull));
}

// one thread per output word: bases [first, first+n) arrive as ASCII in bytes[0..n)
__global__ void __launch_bounds__(256)
seqvec_push_kernel(u64* __restrict__ words, u64 first, const uint8_t* __restrict__ bytes, u64 n,
                   unsigned long long* __restrict__ first_bad) {
    const u64 w0 = first >> 5, w1 = (first + n + 31u) >> 5;   // words [w0, w1) receive bases
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 w = w0 + (u64)blockIdx.x * blockDim.x + threadIdx.x; w < w1; w += stride) {
        const u64 b_lo = w * 32u > first ? w * 32u : first;
        const u64 b_hi = (w + 1u) * 32u < first + n ? (w + 1u) * 32u : first + n;
        u64 v = (b_lo > w * 32u) ? (words[w] & ((1ull << (2u * (u32)(b_lo - w * 32u))) - 1ull)) : 0ull;   // keep what is already there
        for (u64 b = b_lo; b < b_hi; ++b) {
            const u32 c = encode_base(bytes[b - first]);
            if (c >= 4u) atomicMin(first_bad, (unsigned long long)(b - first));
            v |= (u64)(c & 3u) << (2u * (u32)(b & 31u));
        }
        words[w] = v;
    }
}

__global__ void __launch_bounds__(256)
seqvec_to_bytes_kernel(const u64* __restrict__ words, u64 n, uint8_t* __restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (uint8_t)("ACGT"[(words[i >> 5] >> (2u * (u32)(i & 31u))) & 3u]);
}

__global__ void __launch_bounds__(256)
seqvec_get_kmers_kernel(const u64* __restrict__ words, u64 n_bases, const u64* __restrict__ pos, u64 n, u32 k,
                        u64* __restrict__ out, unsigned long long* __restrict__ first_bad) {
    const u64 n_words = (n_bases + 31u) >> 5;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const u64 p = pos[e];
        if (p >= n_bases || p + k > n_bases) {   // assert!(pos < self.len()), seq_vector.rs:97; the field must lie inside the vector
            atomicMin(first_bad, (unsigned long long)e);
            out[e] = 0;
        } else {
            out[e] = seqvec_field(words, n_words, p, k);
        }
    }
}

__global__ void __launch_bounds__(256)
seqvec_iter_kmers_kernel(const u64* __restrict__ words, u64 n_bases, u64 start, u64 count, u32 k, u64* __restrict__ out) {
    const u64 n_words = (n_bases + 31u) >> 5;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride)
        out[e] = seqvec_field(words, n_words, start + e, k);
}

// canonical scan of read r = bases [r*L, (r+1)*L) of the vector, one lane per read (any k in [1,31], any L)
__global__ void __launch_bounds__(256)
reduce_packed_generic_kernel(const u64* __restrict__ words, u64 n_reads, u32 L, u32 k, u32 want_hash, u32 want_sumfw,
                             kmx_summary* __restrict__ out) {
    Acc acc;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += stride) {
        roll_read_packed(words, r * (u64)L, L, k, [&](u32, u64 fw, u64 rc) {
            const u64 canon = fw < rc ? fw : rc;
            acc.n_valid += 1;
            acc.sum_canon += canon;
            acc.xor_hash ^= lex_hash(canon, k);
            acc.sum_fw += fw;
        });
    }
    flush_acc(acc, out, want_hash != 0u, want_sumfw != 0u);
}

// ---------------------------------------------------------------- minimizers
// hash_one(state, lmer) for the hashers with pinned outputs: LexHasher(hk) (hash.rs:60-71) or identity (write_u64(data))
__device__ __forceinline__ u64 mm_hash(u64 lmer, u32 hasher, u32 hk) { return hasher == KMX_HASH_LEX ? lex_hash(lmer, hk) : lmer; }

// Kmer::minimizer_word (kmer.rs:170-192): leftmost minimum (strict `<` starting from u64::MAX) over the k-w+1 sub-words
__global__ void __launch_bounds__(256)
minimizer_words_kernel(const u64* __restrict__ in, u64 n, u32 k, u32 w, u32 hasher, u32 hk, u64* __restrict__ out_mm,
                       u32* __restrict__ out_off) {
    const u64 mask = mask2k(w);
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const u64 word = in[e];
        u64 best = word & mask, best_h = ~0ull;
        u32 off = 0;
        for (u32 pos = 0; pos + w <= k; ++pos) {
            const u64 mm = (word >> (2u * pos)) & mask;   // sub_kmer_word, kmer.rs:156-162
            const u64 h = mm_hash(mm, hasher, hk);
            if (h < best_h) {
                best = mm;
                best_h = h;
                off = pos;
            }
        }
        out_mm[e] = best;
        out_off[e] = off;
    }
}

// SeqVecMinimizerIter (seq_vector/minimizers.rs:39-141) for every read slice [r*L, (r+1)*L): the monotone deque yields,
// for k-mer i, the leftmost minimum-hash l-mer among positions i..i+k-w (`backmer.hash <= dqmer.hash` keeps the
// earlier of equals); one thread per k-mer evaluates that window directly.
__global__ void __launch_bounds__(256)
seqvec_minimizers_kernel(const u64* __restrict__ words, u64 n_reads, u32 L, u32 k, u32 w, u32 hasher, u32 hk,
                         u64* __restrict__ out_word, u32* __restrict__ out_pos) {
    const u64 W = L - k + 1u, total = n_reads * W;
    const u64 n_words = (n_reads * (u64)L + 31u) >> 5;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const u64 r = e / W;
        const u32 i = (u32)(e - r * W);
        const u64 base = r * (u64)L;
        u64 best = 0, best_h = 0;
        u32 best_pos = i;
        for (u32 pos = i; pos <= i + k - w; ++pos) {
            const u64 mm = seqvec_field(words, n_words, base + pos, w);
            const u64 h = mm_hash(mm, hasher, hk);
            if (pos == i || h < best_h) {
                best = mm;
                best_h = h;
                best_pos = pos;
            }
        }
        out_word[e] = best;
        out_pos[e] = best_pos;
    }
}

// The same function with the l-mer hashes shared through LDS: a block hashes every l-mer of its RB reads once
// (thread per l-mer), then a thread per k-mer scans the k-w+1 staged hashes of its window (leftmost minimum) and
// re-extracts the winning l-mer.  ~4x fewer instructions than hashing each window's l-mers again.
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
seqvec_minimizers_lds_kernel(const u64* __restrict__ words, u64 n_reads, u32 L, u32 k, u32 w, u32 hasher, u32 hk, u32 rb,
                             u64* __restrict__ out_word, u32* __restrict__ out_pos) {
    extern __shared__ __attribute__((aligned(16))) u64 hs[];   // [rb][L-w+1]
    const u32 NL = L - w + 1u, W = L - k + 1u, span = k - w + 1u;
    const u64 n_words = (n_reads * (u64)L + 31u) >> 5;
    for (u64 r0 = (u64)blockIdx.x * rb; r0 < n_reads; r0 += (u64)gridDim.x * rb) {
        const u32 nr = (u32)(n_reads - r0 < rb ? n_reads - r0 : rb);
        for (u32 e = threadIdx.x; e < nr * NL; e += THREADS) {
            const u32 r = e / NL, p = e - r * NL;
            hs[e] = mm_hash(seqvec_field(words, n_words, (r0 + r) * (u64)L + p, w), hasher, hk);
        }
        __syncthreads();
        for (u32 e = threadIdx.x; e < nr * W; e += THREADS) {
            const u32 r = e / W, i = e - r * W;
            const u64* __restrict__ h = hs + r * NL + i;
            u64 best_h = h[0];
            u32 best = 0;
            for (u32 j = 1; j < span; ++j) {
                const u64 v = h[j];
                if (v < best_h) {   // strict: the earlier of equal hashes stays (minimizers.rs:71 keeps `backmer.hash <= dqmer.hash`)
                    best_h = v;
                    best = j;
                }
            }
            const u64 slot = (r0 + r) * (u64)W + i;
            out_word[slot] = seqvec_field(words, n_words, (r0 + r) * (u64)L + i + best, w);
            out_pos[slot] = i + best;
        }
        __syncthreads();
    }
}

// The same function as a sliding-window minimum (hashes of at most 56 bits, L <= 256): every l-mer becomes the key
// (hash << 8) | position -- the minimum of keys is the leftmost minimum-hash l-mer, minimizers.rs:71 -- and the
// minimum over the k-w+1 keys of a window is min(M[i], M[i + span - len]) with M = minima over len = 2^J <= span
// consecutive keys, built by J doubling passes over the block's reads in LDS (ping-pong buffers).  log2(span) + 1
// passes instead of span compares per k-mer: 17 -> 5 at k = 31, w = 15.
// The reads themselves are staged too, realigned to bit 0 (FW) and -- for LexHasher(w) on w-base l-mers, whose hash is
// the l-mer with its bases in reverse order -- base-reversed (RV): an l-mer and its hash are then two bit-field reads
// from LDS instead of two unaligned fetches from global memory and five 64-bit swap stages.
// MODE 0: identity hasher, 1: LexHasher(hk == w), 2: LexHasher(any hk).
__device__ __forceinline__ u64 lds_field(const u32* __restrict__ a, u32 bitoff, u32 nbits /* <= 56 */) {
    const u32 q = bitoff >> 5, sh = bitoff & 31u;
    const u64 lo = (u64)a[q] | ((u64)a[q + 1u] << 32);
    const u64 v = sh ? ((lo >> sh) | ((u64)a[q + 2u] << (64u - sh))) : lo;
    return v & ((1ull << nbits) - 1ull);
}
// ... and a field of at most 25 bits: two dwords, one funnel shift
__device__ __forceinline__ u32 lds_field32(const u32* __restrict__ a, u32 bitoff, u32 nbits /* <= 25 */) {
    const u32 q = bitoff >> 5, sh = bitoff & 31u;
    return __builtin_amdgcn_alignbit(a[q + 1u], a[q], sh) & ((1u << nbits) - 1u);
}
// K32 (round 6): hash and position fit one dword (hash bits + 8 <= 32, l-mers of up to 12 bases): u32 keys -- see kmx_minimizers.hip
template <int THREADS, int RB, int MODE, bool K32 = false>
__global__ void __launch_bounds__(THREADS)
seqvec_minimizers_slide_kernel(const u64* __restrict__ words, u64 n_reads, u32 L, u32 k, u32 w, u32 hk,
                               u64* __restrict__ out_word, u32* __restrict__ out_pos) {
    static_assert(THREADS == 16 * RB, "16 threads per read");
    using key_t = std::conditional_t<K32, u32, u64>;
    extern __shared__ __attribute__((aligned(16))) u64 hs[];   // keys [2][RB][NL] (8 bytes each reserved, K32 uses half), then FW [RB][ND], RV [RB][ND] (u32)
    const u32 NL = L - w + 1u, W = L - k + 1u, span = k - w + 1u;
    const u32 ND = ((2u * L + 31u) >> 5) + 2u;                  // dwords of a staged read (+2: the field reads look ahead)
    u32* FW = reinterpret_cast<u32*>(hs + 2u * RB * NL);
    u32* RV = FW + RB * ND;
    const u64 n_words = (n_reads * (u64)L + 31u) >> 5;
    const u32 r = threadIdx.x >> 4, j16 = threadIdx.x & 15u;
    const u32 recipW = W > 1u ? (u32)(0x100000000ull / W) + 1u : 0u;   // e / W = umulhi(e, recipW) for e < RB * W <= 2^12 (W = 1: e itself)
    for (u64 r0 = (u64)blockIdx.x * RB; r0 < n_reads; r0 += (u64)gridDim.x * RB) {
        const u32 nr = (u32)(n_reads - r0 < RB ? n_reads - r0 : RB);
        key_t* A = reinterpret_cast<key_t*>(hs);
        key_t* B = reinterpret_cast<key_t*>(hs + RB * NL);
        // the read, realigned: dword d = its bits [32d, 32d+32) (whatever follows the read in the vector comes along: never looked at)
        if (r < nr) {
            const u64 bit0 = 2u * (r0 + r) * (u64)L;
            for (u32 d = j16; d < ND; d += 16u) {
                const u64 b = bit0 + 32u * d;
                const u64 q = b >> 6;
                const u32 sh = (u32)(b & 63u);
                const u64 lo = q < n_words ? words[q] : 0ull, hi = (sh && q + 1u < n_words) ? words[q + 1u] : 0ull;
                FW[r * ND + d] = (u32)(sh ? ((lo >> sh) | (hi << (64u - sh))) : lo);
            }
        }
        __syncthreads();
        if (MODE == 1 && r < nr) {
            // base-reversed copy: dword d holds bases L-1-16d-j (j = 0..15) = the 32 bits at base offset L-16d-16, group-reversed
            for (u32 d = j16; d < ND; d += 16u) {
                const int off = (int)L - 16 * (int)d - 16;      // may be negative: the read has fewer bases left
                u32 x;
                if (off >= 0) x = (u32)lds_field(FW + r * ND, 2u * (u32)off, 32u);
                else x = off > -16 ? FW[r * ND] << (2u * (u32)(-off)) : 0u;
                x = __builtin_bitreverse32(x);
                RV[r * ND + d] = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);   // bit-reversed pairs back in order
            }
        }
        if (MODE == 1) __syncthreads();
        if (r < nr) {
            for (u32 p = j16; p < NL; p += 16u) {
                if constexpr (K32) {
                    u32 h;
                    if (MODE == 1) h = lds_field32(RV + r * ND, 2u * (L - p - w), 2u * w);
                    else {
                        const u32 lm = lds_field32(FW + r * ND, 2u * p, 2u * w);
                        h = MODE == 0 ? lm : revgroups32(lm) >> (32u - 2u * hk);      // (= lex_hash on a value of one dword)
                    }
                    A[r * NL + p] = (h << 8) | p;
                } else {
                    u64 h;
                    if (MODE == 1) h = lds_field(RV + r * ND, 2u * (L - p - w), 2u * w);
                    else {
                        const u64 lm = lds_field(FW + r * ND, 2u * p, 2u * w);
                        h = MODE == 0 ? lm : lex_hash(lm, hk);
                    }
                    A[r * NL + p] = (h << 8) | p;
                }
            }
        }
        __syncthreads();
        // minima over len consecutive keys: len x 4 per pass while that fits the span (two passes at span = 17 where doubling took four:
        // half the LDS writes and block barriers), then one doubling if there is room.  Past the read's last key: that key once more --
        // a minimum does not mind, and one v_min_u32 on the index is a third of a compare and two selects on a 64-bit key.  (Round 6,
        // as kmx_minimizers.hip: the kernel is bound by VALU issue, profiles/r06_pmc_minimizers.txt.)
        u32 len = 1;
        while (4u * len <= span) {
            if (r < nr) {
                for (u32 p = j16; p < NL; p += 16u) {
                    const key_t* const ap = A + r * NL + p;
                    const u32 room = NL - 1u - p;
                    key_t m = ap[0];
                    const key_t b = ap[len < room ? len : room], c = ap[2u * len < room ? 2u * len : room], d = ap[3u * len < room ? 3u * len : room];
                    m = m < b ? m : b;
                    const key_t m2 = c < d ? c : d;
                    B[r * NL + p] = m < m2 ? m : m2;
                }
            }
            __syncthreads();
            key_t* t = A; A = B; B = t;
            len *= 4u;
        }
        if (2u * len <= span) {
            if (r < nr) {
                for (u32 p = j16; p < NL; p += 16u) {
                    const u32 room = NL - 1u - p;
                    const key_t a = A[r * NL + p];
                    const key_t b = A[r * NL + p + (len < room ? len : room)];
                    B[r * NL + p] = a < b ? a : b;
                }
            }
            __syncthreads();
            key_t* t = A; A = B; B = t;
            len *= 2u;
        }
        const u32 second = span - len;      // the window [i, i+span) = [i, i+len) u [i+second, i+second+len)
        // (the block's slots are one run from a wave-uniform first slot: the stores take a 32-bit index)
        u64* const ow = out_word + r0 * (u64)W;
        u32* const op = out_pos + r0 * (u64)W;
        for (u32 e = threadIdx.x; e < nr * W; e += THREADS) {
            const u32 rr = W > 1u ? __umulhi(e, recipW) : e, i = e - rr * W;
            const key_t a = A[rr * NL + i], b = A[rr * NL + i + second];
            const u32 pos = (u32)((a < b ? a : b) & 0xFFu);
            const u64 lmer = K32 ? (u64)lds_field32(FW + rr * ND, 2u * pos, 2u * w) : lds_field(FW + rr * ND, 2u * pos, 2u * w);
            __builtin_nontemporal_store(lmer, &ow[e]);
            __builtin_nontemporal_store(pos, &op[e]);
        }
        __syncthreads();
    }
}

static inline unsigned sgrid(u64 n, int n_cu) {
    u64 g = (n + 255u) / 256u;
    const u64 cap = (u64)n_cu * 16u;
    if (g > cap) g = cap;
    return (unsigned)(g ? g : 1);
}

hipError_t launch_seqvec_push(u64* words, u64 first, const uint8_t* bytes, u64 n, unsigned long long* first_bad, int n_cu,
                              hipStream_t st) {
    hipLaunchKernelGGL(seqvec_push_kernel, dim3(sgrid((n + 31u) / 32u + 1u, n_cu)), dim3(256), 0, st, words, first, bytes, n, first_bad);
    return hipGetLastError();
}
hipError_t launch_seqvec_to_bytes(const u64* words, u64 n, uint8_t* out, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(seqvec_to_bytes_kernel, dim3(sgrid(n, n_cu)), dim3(256), 0, st, words, n, out);
    return hipGetLastError();
}
hipError_t launch_seqvec_get_kmers(const u64* words, u64 n_bases, const u64* pos, u64 n, u32 k, u64* out,
                                   unsigned long long* first_bad, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(seqvec_get_kmers_kernel, dim3(sgrid(n, n_cu)), dim3(256), 0, st, words, n_bases, pos, n, k, out, first_bad);
    return hipGetLastError();
}
hipError_t launch_seqvec_iter_kmers(const u64* words, u64 n_bases, u64 start, u64 count, u32 k, u64* out, int n_cu,
                                    hipStream_t st) {
    hipLaunchKernelGGL(seqvec_iter_kmers_kernel, dim3(sgrid(count, n_cu)), dim3(256), 0, st, words, n_bases, start, count, k, out);
    return hipGetLastError();
}
hipError_t launch_reduce_packed_generic(const u64* words, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw,
                                        kmx_summary* out, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(reduce_packed_generic_kernel, dim3(sgrid(n_reads, n_cu)), dim3(256), 0, st, words, n_reads, L, k,
                       want_hash ? 1u : 0u, want_sumfw ? 1u : 0u, out);
    return hipGetLastError();
}

hipError_t launch_minimizer_words(const u64* in, u64 n, u32 k, u32 w, u32 hasher, u32 hk, u64* out_mm, u32* out_off, int n_cu,
                                  hipStream_t st) {
    hipLaunchKernelGGL(minimizer_words_kernel, dim3(sgrid(n, n_cu)), dim3(256), 0, st, in, n, k, w, hasher, hk, out_mm, out_off);
    return hipGetLastError();
}
hipError_t launch_seqvec_minimizers(const u64* words, u64 n_reads, u32 L, u32 k, u32 w, u32 hasher, u32 hk, u64* out_word,
                                    u32* out_pos, int n_cu, hipStream_t st) {
    const u32 NL = L - w + 1u;
    const u32 hash_bits = hasher == KMX_HASH_LEX ? 2u * hk : 2u * w;
    if (hash_bits <= 56u && w <= 28u && L <= 256u && k > w) {   // (hash, position) keys fit a u64: sliding-window minimum
        constexpr int RB = 16;
        u64 grid = (n_reads + RB - 1u) / RB;
        const u64 cap = (u64)n_cu * 8u;
        if (grid > cap) grid = cap;
        const u32 ND = ((2u * L + 31u) >> 5) + 2u;
        const size_t lds = (size_t)2u * RB * NL * 8u + (size_t)2u * RB * ND * 4u;
        const int mode = hasher != KMX_HASH_LEX ? 0 : (hk == w ? 1 : 2);
        const bool k32 = hash_bits + 8u <= 32u && 2u * w + 8u <= 32u;   // hash and position in one dword (and the l-mer itself in 24 bits)
        auto k0 = k32 ? seqvec_minimizers_slide_kernel<256, RB, 0, true> : seqvec_minimizers_slide_kernel<256, RB, 0, false>;
        auto k1 = k32 ? seqvec_minimizers_slide_kernel<256, RB, 1, true> : seqvec_minimizers_slide_kernel<256, RB, 1, false>;
        auto k2 = k32 ? seqvec_minimizers_slide_kernel<256, RB, 2, true> : seqvec_minimizers_slide_kernel<256, RB, 2, false>;
        auto kern = mode == 0 ? k0 : mode == 1 ? k1 : k2;
        if (lds > 64u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid ? grid : 1)), dim3(256), lds, st, words, n_reads, L, k, w, hk, out_word, out_pos);
        return hipGetLastError();
    }
    if ((size_t)NL * 8u <= 48u * 1024u) {   // the staged hashes of at least one read fit: LDS-shared kernel
        u32 rb = (48u * 1024u) / (NL * 8u);
        if (rb > 16u) rb = 16u;
        u64 grid = (n_reads + rb - 1u) / rb;
        const u64 cap = (u64)n_cu * 8u;
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL(seqvec_minimizers_lds_kernel<256>, dim3((unsigned)(grid ? grid : 1)), dim3(256), (size_t)rb * NL * 8u, st, words,
                           n_reads, L, k, w, hasher, hk, rb, out_word, out_pos);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(seqvec_minimizers_kernel, dim3(sgrid(n_reads * (u64)(L - k + 1u), n_cu)), dim3(256), 0, st, words, n_reads,
                       L, k, w, hasher, hk, out_word, out_pos);
    return hipGetLastError();
}

}  // namespace kmx
