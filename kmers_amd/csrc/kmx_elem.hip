// kmx_elem.hip -- K0 synthetic read generator + K5 element-wise batch operations
// (one element per lane).  Each kernel cites the reference function it restates.
#include "kmx_device.h"

namespace kmx {

// ---------------------------------------------------------------- K0 generator
// byte g of the stream = "ACGT"[(splitmix64(seed + g/32) >> 2*(g%32)) & 3]   (BUILD-DEFINED)
__device__ __forceinline__ u32 expand4(u32 b) {  // 4 two-bit codes -> 4 ASCII letters
    const u32 sel = (b & 0x03u) | ((b & 0x0Cu) << 6) | ((b & 0x30u) << 12) | ((b & 0xC0u) << 18);
    return __builtin_amdgcn_perm(0u, 0x54474341u /* 'A','C','G','T' */, sel);
}

// fast path: first_byte % 16 == 0 and out 16-byte aligned; one 16-byte chunk per thread
__global__ void __launch_bounds__(256) gen_reads_vec_kernel(u64 seed, u64 first_byte, uint4* __restrict__ out, u64 n_chunks) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const u64 pos = first_byte + 16u * c;
        const u64 z = splitmix64(seed + (pos >> 5));
        const u32 h = (u32)(z >> (2u * (u32)(pos & 31u)));
        uint4 v;
        v.x = expand4(h & 0xFFu);
        v.y = expand4((h >> 8) & 0xFFu);
        v.z = expand4((h >> 16) & 0xFFu);
        v.w = expand4(h >> 24);
        out[c] = v;
    }
}

__global__ void __launch_bounds__(256) gen_reads_byte_kernel(u64 seed, u64 first_byte, uint8_t* __restrict__ out, u64 n) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 g = first_byte + i;
        const u64 z = splitmix64(seed + (g >> 5));
        out[i] = (uint8_t)((0x54474341u >> (8u * (u32)((z >> (2u * (u32)(g & 31u))) & 3u))) & 0xFFu);
    }
}

// ------------------------------------------------------- naive_impl element ops

// Kmer::from(&[u8]) (src/naive_impl/kmer.rs:234-251): bytes reversed, w = (w<<2)|code; strict:
// a non-ACGTacgt byte is a panic in the reference (mod.rs:35) -> record the lowest offending index
__global__ void __launch_bounds__(256)
kmers_from_bytes_kernel(const uint8_t* __restrict__ seqs, u64 n, u32 k, u64* __restrict__ words,
                        unsigned long long* __restrict__ first_bad) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const uint8_t* s = seqs + e * (u64)k;
        u64 w = 0;
        for (int i = (int)k - 1; i >= 0; --i) {
            const u32 b = encode_base(s[i]);
            if (b >= 4u) atomicMin(first_bad, (unsigned long long)(e * (u64)k + (u64)i));
            w = (w << 2) | (u64)(b & 3u);
        }
        words[e] = w;
    }
}

// Streaming u64 -> u64 maps: two words per lane and step (16-byte nt loads and stores when both arrays allow it), the odd
// element and unaligned arrays one word at a time.  The three calls below ran at the rate of a plain device copy (4.3 TB/s of
// traffic); what they add to it is nothing the memory system notices.
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
template <class F>
__device__ __forceinline__ void map_words(const u64* __restrict__ in, u64 n, u64* __restrict__ out, F f) {
    const u64 stride = (u64)gridDim.x * blockDim.x, tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 done = 0;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0u) {
        const u64 pairs = n >> 1;
        const u64x2* in2 = reinterpret_cast<const u64x2*>(in);
        u64x2* out2 = reinterpret_cast<u64x2*>(out);
        for (u64 e = tid; e < pairs; e += stride) {
            const u64x2 v = __builtin_nontemporal_load(in2 + e);
            const u64x2 r = {f(v.x), f(v.y)};
            __builtin_nontemporal_store(r, out2 + e);
        }
        done = pairs << 1;
    }
    for (u64 e = done + tid; e < n; e += stride) out[e] = f(in[e]);
}

// Kmer::to_reverse_complement (kmer.rs:124-136)
__global__ void __launch_bounds__(256) revcomp_words_kernel(const u64* __restrict__ in, u64 n, u32 k, u64* __restrict__ out) {
    map_words(in, n, out, [k](u64 w) { return revcomp_word(w, k); });
}

// Kmer::to_canonical / is_canonical (kmer.rs:55-74): canonical <=> data <= rc.data
__global__ void __launch_bounds__(256)
canonical_words_kernel(const u64* __restrict__ in, u64 n, u32 k, u64* __restrict__ canon, uint8_t* __restrict__ is_canon) {
    if (canon && !is_canon) {     // the words alone: the streaming map
        map_words(in, n, canon, [k](u64 w) { const u64 rc = revcomp_word(w, k); return w <= rc ? w : rc; });
        return;
    }
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const u64 w = in[e], rc = revcomp_word(w, k);
        const bool c = w <= rc;
        if (canon) canon[e] = c ? w : rc;
        if (is_canon) is_canon[e] = c ? 1 : 0;
    }
}

// hash_one(&LexHasherState::new(hk), kmer) (hash.rs:10-20,60-71) / identity (hash.rs:4-8)
__global__ void __launch_bounds__(256)
hash_words_kernel(const u64* __restrict__ in, u64 n, u32 hasher, u32 hk, u64* __restrict__ out) {
    if (hasher == KMX_HASH_LEX) map_words(in, n, out, [hk](u64 w) { return lex_hash(w, hk); });
    else map_words(in, n, out, [](u64 w) { return w; });
}

// hash_one(&DefaultHasher / RandomState, kmer): SipHash-1-3 of the word's 8 little-endian bytes (kmx.h; hash.rs:4-20).  One full
// message block m = w, then the final block b = 8 << 56 (length in the top byte, no tail bytes).
__device__ __forceinline__ u64 rotl64(u64 x, int b) { return (x << b) | (x >> (64 - b)); }
__device__ __forceinline__ void sip_round(u64& v0, u64& v1, u64& v2, u64& v3) {
    v0 += v1; v1 = rotl64(v1, 13); v1 ^= v0; v0 = rotl64(v0, 32);
    v2 += v3; v3 = rotl64(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl64(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl64(v1, 17); v1 ^= v2; v2 = rotl64(v2, 32);
}
__device__ __forceinline__ u64 siphash13_u64(u64 w, u64 k0, u64 k1) {
    u64 v0 = k0 ^ 0x736f6d6570736575ull, v1 = k1 ^ 0x646f72616e646f6dull, v2 = k0 ^ 0x6c7967656e657261ull, v3 = k1 ^ 0x7465646279746573ull;
    v3 ^= w; sip_round(v0, v1, v2, v3); v0 ^= w;
    const u64 b = 8ull << 56;
    v3 ^= b; sip_round(v0, v1, v2, v3); v0 ^= b;
    v2 ^= 0xffull;
    sip_round(v0, v1, v2, v3); sip_round(v0, v1, v2, v3); sip_round(v0, v1, v2, v3);
    return v0 ^ v1 ^ v2 ^ v3;
}
__global__ void __launch_bounds__(256)
hash_words_sip13_kernel(const u64* __restrict__ in, u64 n, u64 k0, u64 k1, u64* __restrict__ out) {
    map_words(in, n, out, [k0, k1](u64 w) { return siphash13_u64(w, k0, k1); });
}

// CanonicalKmer::get_word_equivalency (canonical_kmer.rs:152-161)
__global__ void __launch_bounds__(256)
match_words_kernel(const u64* __restrict__ fw, const u64* __restrict__ rc, const u64* __restrict__ other, u64 n,
                   uint8_t* __restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const u64 o = other[e];
        out[e] = fw[e] == o ? KMX_IDENTITY_MATCH : (rc[e] == o ? KMX_TWIN_MATCH : KMX_NO_MATCH);
    }
}

// CanonicalKmer::append_base (canonical_kmer.rs:90-94) / prepend_base (:97-101)
template <bool APPEND>
__global__ void __launch_bounds__(256)
ck_shift_kernel(u64* __restrict__ fw, u64* __restrict__ rc, const uint8_t* __restrict__ bases, u64 n, u32 k,
                uint8_t* __restrict__ dropped) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 mask = mask2k(k);
    const u32 top = 2u * k - 2u;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const u64 b = bases[e] & 3u, cb = 3u - b;
        u64 f = fw[e], r = rc[e];
        u64 d;
        if (APPEND) {
            d = f & 3u;                          // kmer.rs:98-102 on fw
            f = (f >> 2) | (b << top);
            r = mask & ((r << 2) | cb);          // kmer.rs:91-95 on rc
        } else {
            d = (f >> top) & 3u;                 // kmer.rs:91-95 on fw
            f = mask & ((f << 2) | b);
            r = (r >> 2) | (cb << top);          // kmer.rs:98-102 on rc
        }
        fw[e] = f;
        rc[e] = r;
        if (dropped) dropped[e] = (uint8_t)d;
    }
}

// ------------------------------------------------- encoding::{Naive,Xor10} ops

// code of a nucleotide under enc: nuc2bits (src/encoding/naive.rs:78-85, :14-16)
__device__ __forceinline__ u32 nuc2bits(u32 enc, u32 nuc) { return (enc >> (6u - 2u * ((nuc >> 1) & 3u))) & 3u; }

// Encoding::encode (naive.rs:116-124 / xor10.rs:52-60): base idx -> flat bits 2idx..2idx+1
__global__ void __launch_bounds__(256)
encode_kmers_kernel(const uint8_t* __restrict__ seqs, u64 n, u32 seq_len, u32 enc, u32 B, u64* __restrict__ words) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const uint8_t* s = seqs + e * (u64)seq_len;
        for (u32 wd = 0; wd < B; ++wd) {
            u64 w = 0;
            const u32 lo = wd * 32u;
            const u32 hi = seq_len < lo + 32u ? seq_len : lo + 32u;
            for (u32 i = lo; i < hi; ++i) w |= (u64)nuc2bits(enc, s[i]) << (2u * (i - lo));
            words[e * B + wd] = w;
        }
    }
}

// b.windows(K).map(|x| Kmer::new(x,&enc)) (benches/simple_benchmark.rs:24-34), uniform reads
__global__ void __launch_bounds__(256)
encode_windows_kernel(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 k, u32 enc, u32 B, u64* __restrict__ words) {
    const u32 nwin = L - k + 1u;
    const u64 total = n_reads * (u64)nwin;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const u64 r = e / nwin;
        const u32 p = (u32)(e - r * nwin);
        const uint8_t* s = bases + r * (u64)L + p;
        for (u32 wd = 0; wd < B; ++wd) {
            u64 w = 0;
            const u32 lo = wd * 32u;
            const u32 hi = k < lo + 32u ? k : lo + 32u;
            for (u32 i = lo; i < hi; ++i) w |= (u64)nuc2bits(enc, s[i]) << (2u * (i - lo));
            words[e * B + wd] = w;
        }
    }
}

// Encoding::rev_comp::<K> (naive.rs:138-154): out base i = complement(in base K-1-i) for i<K,
// bits >= 2K untouched.  comp_lut: 4 x 2-bit complement table for this enc (host-computed from
// naive.rs:98-109), packed in one byte.
__global__ void __launch_bounds__(256)
encoding_rev_comp_kernel(const u64* __restrict__ in, u64 n, u32 K, u32 comp_lut, u32 B, u64* __restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const u64* a = in + e * B;
        u64 res[4];
        for (u32 wd = 0; wd < B; ++wd) res[wd] = a[wd];
        for (u32 i = 0; i < K; ++i) {
            const u32 j = K - 1u - i;
            const u32 c = (u32)(a[j >> 5] >> (2u * (j & 31u))) & 3u;
            const u64 cc = (comp_lut >> (2u * c)) & 3u;
            const u32 sh = 2u * (i & 31u);
            res[i >> 5] = (res[i >> 5] & ~(3ull << sh)) | (cc << sh);
        }
        for (u32 wd = 0; wd < B; ++wd) out[e * B + wd] = res[wd];
    }
}

// Encoding::decode (naive.rs:126-136): ALL 32*B slots; nuc_lut = 4 letters indexed by code
__global__ void __launch_bounds__(256)
encoding_decode_kernel(const u64* __restrict__ in, u64 n, u32 nuc_lut, u32 B, uint8_t* __restrict__ seqs) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        for (u32 wd = 0; wd < B; ++wd) {
            u64 w = in[e * B + wd];
            uint8_t* o = seqs + (e * B + wd) * 32u;
            for (u32 i = 0; i < 32u; ++i) {
                o[i] = (uint8_t)((nuc_lut >> (8u * (u32)(w & 3u))) & 0xFFu);
                w >>= 2;
            }
        }
    }
}

// ------------------------------------------------- decode / display direction

// Kmer::sub_kmer_word (src/naive_impl/kmer.rs:156-162): (word >> 2*pos) & MASK_TABLE[width]  (MASK_TABLE[32] == 0, :617)
__global__ void __launch_bounds__(256)
sub_kmer_words_kernel(const u64* __restrict__ in, u64 n, u32 pos, u32 width, u64* __restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 mask = width >= 32u ? 0ull : ((1ull << (2u * width)) - 1ull);
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) out[e] = (in[e] >> (2u * pos)) & mask;
}

// String::from(Kmer) (kmer.rs:196-207, lower case) / bitmer_to_bytes (src/kmer.rs:71-91, upper case): letter i = table[(w >> 2i) & 3].
// One thread per 4 letters (one dword store when the output is aligned, bytes otherwise).
__global__ void __launch_bounds__(256)
kmers_to_bytes_kernel(const u64* __restrict__ in, u64 n, u32 k, u32 letters /* 4 bytes indexed by code */, uint8_t* __restrict__ out) {
    const u32 q = (k + 3u) >> 2;                 // groups of 4 letters per k-mer
    const u64 total = n * (u64)q;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const u64 e = t / q;
        const u32 g = (u32)(t - e * q);
        const u32 bits = (u32)(in[e] >> (8u * g)) & 0xFFu;
        const u32 sel = (bits & 0x03u) | ((bits & 0x0Cu) << 6) | ((bits & 0x30u) << 12) | ((bits & 0xC0u) << 18);
        const u32 four = __builtin_amdgcn_perm(0u, letters, sel);
        uint8_t* o = out + e * (u64)k + 4u * g;
        const u32 cnt = k - 4u * g < 4u ? k - 4u * g : 4u;
        if (cnt == 4u && (reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
            *reinterpret_cast<u32*>(o) = four;
        } else {
            for (u32 i = 0; i < cnt; ++i) o[i] = (uint8_t)(four >> (8u * i));
        }
    }
}

// ------------------------------------------------- Encoding<P, B> on the byte image of [P; B] (utils::Data: u8 .. u128)
// flat bit i of a [P; B] = byte i / 8, bit i % 8 of its little-endian image, whatever P (bit_field BitArray)

// Encoding::encode (naive.rs:116-124): nb = B * size_of::<P>() bytes per k-mer, unused high bits zero
__global__ void __launch_bounds__(256)
encode_kmers_bytes_kernel(const uint8_t* __restrict__ seqs, u64 n, u32 seq_len, u32 enc, u32 nb, uint8_t* __restrict__ arrays) {
    const u64 total = n * (u64)nb;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const u64 e = t / nb;
        const u32 by = (u32)(t - e * nb);
        const uint8_t* s = seqs + e * (u64)seq_len + 4u * by;
        u32 v = 0;
        for (u32 i = 0; i < 4u && 4u * by + i < seq_len; ++i) v |= nuc2bits(enc, s[i]) << (2u * i);
        arrays[t] = (uint8_t)v;
    }
}

// Encoding::rev_comp::<K> (naive.rs:138-154): out base i = complement(in base K-1-i) for i < K, bits >= 2K unchanged
__global__ void __launch_bounds__(256)
encoding_rev_comp_bytes_kernel(const uint8_t* __restrict__ in, u64 n, u32 K, u32 comp_lut, u32 nb, uint8_t* __restrict__ out) {
    const u64 total = n * (u64)nb;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const u64 e = t / nb;
        const u32 by = (u32)(t - e * nb);
        const uint8_t* a = in + e * (u64)nb;
        u32 v = a[by];
        for (u32 i = 0; i < 4u; ++i) {
            const u32 b = 4u * by + i;
            if (b >= K) break;
            const u32 j = K - 1u - b;
            const u32 c = ((u32)a[j >> 2] >> (2u * (j & 3u))) & 3u;
            v = (v & ~(3u << (2u * i))) | (((comp_lut >> (2u * c)) & 3u) << (2u * i));
        }
        out[t] = (uint8_t)v;
    }
}

// Encoding::decode (naive.rs:126-136): ALL 4 * nb letters
__global__ void __launch_bounds__(256)
encoding_decode_bytes_kernel(const uint8_t* __restrict__ in, u64 total_bytes, u32 nuc_lut, uint8_t* __restrict__ seqs) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total_bytes; t += stride) {
        const u32 bits = in[t];
        const u32 sel = (bits & 0x03u) | ((bits & 0x0Cu) << 6) | ((bits & 0x30u) << 12) | ((bits & 0xC0u) << 18);
        const u32 four = __builtin_amdgcn_perm(0u, nuc_lut, sel);
        uint8_t* o = seqs + 4u * t;
        if ((reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
            *reinterpret_cast<u32*>(o) = four;
        } else {
            for (u32 i = 0; i < 4u; ++i) o[i] = (uint8_t)(four >> (8u * i));
        }
    }
}

// ------------------------------------------------- hash fold of a reduce pass under another hasher
// The scan kernels fold xor of LexHasher(k)(canonical word) (for the bit-sliced kernels it is the parity of counters they
// keep anyway).  Every hasher offered is a GF(2)-linear map of the word -- LexHasher(hk): the 2-bit groups reversed and
// shifted (hash.rs:60-71); identity: the word itself (hash.rs:4-8) -- so the xor over all k-mers of H(word) is H(xor of
// the words), and Lex(k) is an involution on k-base words: the fold under any other hasher follows from the Lex(k) fold.
__global__ void fix_hash_fold_kernel(kmx_summary* __restrict__ out, u32 k, u32 hasher, u32 hk) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const u64 x = lex_hash(out->xor_hash, k);   // xor of the canonical words themselves
    out->xor_hash = hasher == KMX_HASH_LEX ? lex_hash(x, hk) : x;
}

// ------------------------------------------------- measurement helper: read-only stream with the scan's load shape
// each wave streams whole 9600-byte tiles (600 16-byte chunks), tiles striped over the waves, next tile requested before
// the current one is folded
__global__ void __launch_bounds__(256) calib_stream_read_kernel(const uint8_t* __restrict__ buf, u64 n_tiles, u64 tail_chunks,
                                                                unsigned long long* __restrict__ out) {
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    constexpr int TILE16 = 600, IT = (TILE16 + 63) / 64;
    const u32x4* __restrict__ p = reinterpret_cast<const u32x4*>(buf);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = (u64)gridDim.x * 4u;
    u32 acc = 0;
    u32x4 w[IT];
    auto issue = [&](u64 t) {
        const u32x4* tb = p + t * TILE16;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            u32 c = it * 64u + lane;
            c = c < TILE16 ? c : TILE16 - 1;
            w[it] = __builtin_nontemporal_load(tb + c);
        }
    };
    if (wave < n_tiles) issue(wave);
    for (u64 t = wave; t < n_tiles; t += n_waves) {
        u32 a = 0;
#pragma unroll
        for (int it = 0; it < IT; ++it) a ^= w[it].x ^ w[it].y ^ w[it].z ^ w[it].w;
        acc ^= a;
        issue(t + n_waves < n_tiles ? t + n_waves : t);
    }
    if (wave == 0) {   // the chunks behind the last whole tile
        for (u64 c = lane; c < tail_chunks; c += 64u) {
            const u32x4 v = p[n_tiles * TILE16 + c];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    acc = (u32)wave_xor((u64)acc);
    if (lane == 0 && acc != 0u) atomicXor(out, (unsigned long long)acc);
}

// ------------------------------------------------------------------ launchers

static inline unsigned egrid(u64 n, int n_cu) {
    u64 g = (n + 255u) / 256u;
    const u64 cap = (u64)n_cu * 16u;
    if (g > cap) g = cap;
    return (unsigned)(g ? g : 1);
}

hipError_t launch_gen_reads(u64 seed, u64 first_byte, uint8_t* out, u64 nbytes, int n_cu, hipStream_t st) {
    if (nbytes == 0) return hipSuccess;
    if (((first_byte & 15u) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
        const u64 n_chunks = nbytes >> 4;
        if (n_chunks)
            hipLaunchKernelGGL(gen_reads_vec_kernel, dim3(egrid(n_chunks, n_cu)), dim3(256), 0, st, seed, first_byte,
                               reinterpret_cast<uint4*>(out), n_chunks);
        const u64 tail = nbytes & 15u;
        if (tail)
            hipLaunchKernelGGL(gen_reads_byte_kernel, dim3(1), dim3(64), 0, st, seed, first_byte + (n_chunks << 4),
                               out + (n_chunks << 4), tail);
    } else {
        hipLaunchKernelGGL(gen_reads_byte_kernel, dim3(egrid(nbytes, n_cu)), dim3(256), 0, st, seed, first_byte, out, nbytes);
    }
    return hipGetLastError();
}

hipError_t launch_kmers_from_bytes(const uint8_t* seqs, u64 n, u32 k, u64* words, unsigned long long* first_bad, int n_cu,
                                   hipStream_t st) {
    hipLaunchKernelGGL(kmers_from_bytes_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, seqs, n, k, words, first_bad);
    return hipGetLastError();
}
hipError_t launch_revcomp_words(const u64* in, u64 n, u32 k, u64* out, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(revcomp_words_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, in, n, k, out);
    return hipGetLastError();
}
hipError_t launch_canonical_words(const u64* in, u64 n, u32 k, u64* canon, uint8_t* is_canon, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(canonical_words_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, in, n, k, canon, is_canon);
    return hipGetLastError();
}
hipError_t launch_hash_words(const u64* in, u64 n, u32 hasher, u32 hk, u64* out, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(hash_words_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, in, n, hasher, hk, out);
    return hipGetLastError();
}
hipError_t launch_hash_words_sip13(const u64* in, u64 n, u64 k0, u64 k1, u64* out, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(hash_words_sip13_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, in, n, k0, k1, out);
    return hipGetLastError();
}
hipError_t launch_match_words(const u64* fw, const u64* rc, const u64* other, u64 n, uint8_t* out, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(match_words_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, fw, rc, other, n, out);
    return hipGetLastError();
}
hipError_t launch_ck_shift(bool append, u64* fw, u64* rc, const uint8_t* bases, u64 n, u32 k, uint8_t* dropped, int n_cu,
                           hipStream_t st) {
    if (append)
        hipLaunchKernelGGL(ck_shift_kernel<true>, dim3(egrid(n, n_cu)), dim3(256), 0, st, fw, rc, bases, n, k, dropped);
    else
        hipLaunchKernelGGL(ck_shift_kernel<false>, dim3(egrid(n, n_cu)), dim3(256), 0, st, fw, rc, bases, n, k, dropped);
    return hipGetLastError();
}
hipError_t launch_encode_kmers(const uint8_t* seqs, u64 n, u32 seq_len, u32 enc, u32 B, u64* words, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(encode_kmers_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, seqs, n, seq_len, enc, B, words);
    return hipGetLastError();
}
hipError_t launch_encode_windows(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 enc, u32 B, u64* words, int n_cu,
                                 hipStream_t st) {
    hipLaunchKernelGGL(encode_windows_kernel, dim3(egrid(n_reads * (u64)(L - k + 1u), n_cu)), dim3(256), 0, st, bases,
                       n_reads, L, k, enc, B, words);
    return hipGetLastError();
}
hipError_t launch_encoding_rev_comp(const u64* in, u64 n, u32 K, u32 comp_lut, u32 B, u64* out, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(encoding_rev_comp_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, in, n, K, comp_lut, B, out);
    return hipGetLastError();
}
hipError_t launch_encoding_decode(const u64* in, u64 n, u32 nuc_lut, u32 B, uint8_t* seqs, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(encoding_decode_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, in, n, nuc_lut, B, seqs);
    return hipGetLastError();
}

hipError_t launch_sub_kmer_words(const u64* in, u64 n, u32 pos, u32 width, u64* out, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(sub_kmer_words_kernel, dim3(egrid(n, n_cu)), dim3(256), 0, st, in, n, pos, width, out);
    return hipGetLastError();
}
hipError_t launch_kmers_to_bytes(const u64* in, u64 n, u32 k, bool upper, uint8_t* out, int n_cu, hipStream_t st) {
    const u32 letters = upper ? 0x54474341u /* A C G T */ : 0x74676361u /* a c g t */;
    hipLaunchKernelGGL(kmers_to_bytes_kernel, dim3(egrid(n * (u64)((k + 3u) >> 2), n_cu)), dim3(256), 0, st, in, n, k, letters, out);
    return hipGetLastError();
}
hipError_t launch_encode_kmers_bytes(const uint8_t* seqs, u64 n, u32 seq_len, u32 enc, u32 nb, uint8_t* arrays, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(encode_kmers_bytes_kernel, dim3(egrid(n * (u64)nb, n_cu)), dim3(256), 0, st, seqs, n, seq_len, enc, nb, arrays);
    return hipGetLastError();
}
hipError_t launch_encoding_rev_comp_bytes(const uint8_t* in, u64 n, u32 K, u32 comp_lut, u32 nb, uint8_t* out, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(encoding_rev_comp_bytes_kernel, dim3(egrid(n * (u64)nb, n_cu)), dim3(256), 0, st, in, n, K, comp_lut, nb, out);
    return hipGetLastError();
}
hipError_t launch_encoding_decode_bytes(const uint8_t* in, u64 total_bytes, u32 nuc_lut, uint8_t* seqs, int n_cu, hipStream_t st) {
    hipLaunchKernelGGL(encoding_decode_bytes_kernel, dim3(egrid(total_bytes, n_cu)), dim3(256), 0, st, in, total_bytes, nuc_lut, seqs);
    return hipGetLastError();
}
hipError_t launch_fix_hash_fold(kmx_summary* out, u32 k, u32 hasher, u32 hk, hipStream_t st) {
    hipLaunchKernelGGL(fix_hash_fold_kernel, dim3(1), dim3(64), 0, st, out, k, hasher, hk);
    return hipGetLastError();
}
hipError_t launch_calib_stream_read(const uint8_t* buf, u64 nbytes, unsigned long long* out, int n_cu, hipStream_t st) {
    const u64 n_tiles = nbytes / 9600u, tail_chunks = (nbytes - n_tiles * 9600u) >> 4;
    u64 grid = (u64)n_cu * 3u;   // the scan's 3 blocks per CU
    if (grid > (n_tiles + 3u) / 4u) grid = (n_tiles + 3u) / 4u;
    hipLaunchKernelGGL(calib_stream_read_kernel, dim3((unsigned)(grid ? grid : 1)), dim3(256), 0, st, buf, n_tiles, tail_chunks, out);
    return hipGetLastError();
}

// min / max of offsets[r+1] - offsets[r] (clamped to 32 bits) into out[0] / out[1] (preset to ~0 / 0 by the caller)
__global__ void __launch_bounds__(256) length_range_kernel(const u64* __restrict__ offsets, u64 n_reads, u32* __restrict__ out) {
    u32 mn = 0xFFFFFFFFu, mx = 0u;
    for (u64 r = (u64)blockIdx.x * 256u + threadIdx.x; r < n_reads; r += (u64)gridDim.x * 256u) {
        const u64 d = offsets[r + 1u] - offsets[r];
        const u32 len = d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)d;
        mn = len < mn ? len : mn;
        mx = len > mx ? len : mx;
    }
    mx = wave_max_u32(mx);
    mn = ~wave_max_u32(~mn);
    // One pair of atomics per block, and only where it would change something: the 2 x 8192 per-wave atomics on one cache line
    // (~6 ns apiece, all at the end) were half of this kernel's 0.2 ms for 1.7e7 reads.  (A stale look at out[] only costs an atomic.)
    __shared__ u32 part[4][2];
    if ((threadIdx.x & 63u) == 0u) { part[threadIdx.x >> 6][0] = mn; part[threadIdx.x >> 6][1] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (u32 w = 1; w < (blockDim.x >> 6); ++w) {
            mn = part[w][0] < mn ? part[w][0] : mn;
            mx = part[w][1] > mx ? part[w][1] : mx;
        }
        if (mn < __hip_atomic_load(&out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&out[0], mn);
        if (mx > __hip_atomic_load(&out[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&out[1], mx);
    }
}

// Are the reads behind `offsets` all of ONE length L0, k <= L0 <= bound, starting at byte 0?  *gate stays 1 ("run the uniform kernels") if
// so and becomes 2 ("run the ragged kernels") at the first read that is not -- kmx_canonical_reduce launches both scans behind it,
// each returns at once when the gate names the other (kmx_bitslice_kernel.h), and the host never waits for the answer.  gate[1] takes
// L0: the uniform scan was launched for the bound (frame and windows per lane) and reads the length it scans with from there (round 5:
// until then the length had to EQUAL the caller's bound -- untrimmed 150-base reads handed over with a bound of 160, or with none, took
// the ragged kernel: 3.0 instead of 2.6 ms per 1e8).
__global__ void __launch_bounds__(256) offsets_uniform_gate_kernel(const u64* __restrict__ offsets, u64 n_reads, u32 bound, u32 k, u32* __restrict__ gate) {
    bool bad = false, stop = false;
    // (every thread reads the first two offsets: one cache line, and the length is needed before anything can be compared)
    const u64 o0 = offsets[0], len0 = offsets[1] - o0;
    const u32 L = (u32)len0;
    if (o0 != 0ull || len0 < k || len0 > bound) {
        if (blockIdx.x == 0 && threadIdx.x == 0u) atomicMax(gate, 2u);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0u) gate[1] = L;
    // two offsets per 16-byte load (hipMalloc'ed arrays are 256-byte aligned; an odd tail is looked at by itself)
    const bool al16 = (reinterpret_cast<uintptr_t>(offsets) & 15u) == 0u;
    const u64 n_off = n_reads + 1u, pairs = al16 ? n_off >> 1 : 0u;
    u32 it = 0;
    for (u64 j = (u64)blockIdx.x * 256u + threadIdx.x; j < pairs; j += (u64)gridDim.x * 256u, ++it) {
        // (the verdict is in as soon as ANY read differs -- in trimmed FASTQ that is within the first few hundred reads: every wave
        // looks at the gate word now and then and stops reading offsets nobody needs any more; 0.8 GB for 1e8 reads otherwise)
        if ((it & 3u) == 3u && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u) { stop = true; break; }
        const ulonglong2 v = reinterpret_cast<const ulonglong2*>(offsets)[j];
        bad |= v.x != (2u * j) * (u64)L || v.y != (2u * j + 1u) * (u64)L;
        if (__any(bad)) break;
    }
    if (!stop)
        for (u64 i = 2u * pairs + (u64)blockIdx.x * 256u + threadIdx.x; i < n_off; i += (u64)gridDim.x * 256u) bad |= offsets[i] != i * (u64)L;
    // ONE write per block, and none once the word says 2: on ragged input every wave of the grid finds a mismatch in its first
    // step, and 8192 plain stores of the same constant to the same word took 0.28 ms -- whatever the number of reads
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (any_bad && threadIdx.x == 0u && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 2u) atomicMax(gate, 2u);
}

hipError_t launch_offsets_uniform_gate(const u64* offsets, u64 n_reads, u32 bound, u32 k, u32* gate, int n_cu, hipStream_t st) {
    u64 grid = (u64)n_cu * 8u;
    const u64 need = (n_reads / 2u + 256u) / 256u;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(offsets_uniform_gate_kernel, dim3((unsigned)(grid ? grid : 1)), dim3(256), 0, st, offsets, n_reads, bound, k, gate);
    return hipGetLastError();
}

hipError_t launch_length_range(const u64* offsets, u64 n_reads, u32* out, int n_cu, hipStream_t st) {
    u64 grid = (n_reads + 255u) / 256u;
    if (grid > (u64)n_cu * 8u) grid = (u64)n_cu * 8u;
    hipLaunchKernelGGL(length_range_kernel, dim3((unsigned)(grid ? grid : 1)), dim3(256), 0, st, offsets, n_reads, out);
    return hipGetLastError();
}

}  // namespace kmx
