// kmx_minimizers.hip -- SeqVecMinimizerIter (seq_vector/minimizers.rs:39-141) over READS (round 6): what
// SeqVector::from(read).iter_minimizers(k, w, hasher) yields for every read of a batch -- ASCII, one length or behind an offsets
// array -- without building the SeqVector: the reads kmx_fastx_parse hands over go straight in.
//
// The algorithm is the sliding-window minimum of kmx_seqvec.hip (seqvec_minimizers_slide_kernel): every l-mer of a read becomes
// the key (hash << 8) | position -- the minimum of keys is the leftmost minimum-hash l-mer, the tie rule of the reference's
// monotone deque (minimizers.rs:61-81: `backmer.hash <= dqmer.hash` keeps the earlier of equals) -- and the minimum over the
// k - w + 1 keys of a k-mer's window is min(M[i], M[i + span - len]) with M the minima over len = 2^J <= span consecutive
// keys, built by J doubling passes in LDS.  What differs is either end:
//   * in: a block packs its RB reads from ASCII (16 bases per dword, encode16; a byte outside ACGTacgt -- the reference's
//     SeqVector::from panics on it, seq_vector.rs:230-242 / kmer.rs:45-60 -- leaves the read's index in *first_bad), each read
//     from its own start: reads of one length, reads behind offsets, or the SEGMENTS a read longer than 256 bases is cut into
//     (positions are 8 bits of the key);
//   * out: the k-mers of a block's reads are consecutive slots -- one run of whole 128-byte lines when the batch is uniform --
//     written by consecutive threads.
#include "kmx_device.h"

#include <type_traits>

namespace kmx {

__device__ __forceinline__ u64 mmr_field(const u32* __restrict__ a, u32 bitoff, u32 nbits /* <= 56 */) {
    const u32 q = bitoff >> 5, sh = bitoff & 31u;
    const u64 lo = (u64)a[q] | ((u64)a[q + 1u] << 32);
    const u64 v = sh ? ((lo >> sh) | ((u64)a[q + 2u] << (64u - sh))) : lo;
    return v & ((1ull << nbits) - 1ull);
}

// ... and a field of at most 25 bits: two dwords, one funnel shift
__device__ __forceinline__ u32 mmr_field32(const u32* __restrict__ a, u32 bitoff, u32 nbits /* <= 25 */) {
    const u32 q = bitoff >> 5, sh = bitoff & 31u;
    return __builtin_amdgcn_alignbit(a[q + 1u], a[q], sh) & ((1u << nbits) - 1u);
}

// how the reads of a batch lie: read R of the kernel's index space = piece (R % J) of read R / J (J = 1: the reads themselves)
struct MmrGeom {
    const u64* offsets;      // ragged: read r = bases[offsets[r], offsets[r+1]); nullptr: uniform
    const u64* win_offsets;  // ragged: slot of k-mer 0 of read r
    u64 n_reads;             // reads (not pieces)
    u32 L;                   // uniform: bases per read
    u32 J, T;                // pieces per read, k-mers per piece (the last one: what is left)
};

// MODE 0: identity hasher, 1: LexHasher(hk == w), 2: LexHasher(any hk).  RAGGED: reads behind offsets (J == 1).
// K32 (round 6): hash and position fit ONE dword (hash bits + 8 <= 32: l-mers of up to 12 bases) -- the keys are u32, a radix-4 pass is
// four dword reads, a v_min3_u32 and a v_min_u32 where 64-bit keys cost three compares and six selects, and a field is two dwords and
// one funnel shift: the kernel is bound by VALU issue (profiles/r06_pmc_minimizers.txt), k = 21 / w = 11 3.3 -> see there.
template <int THREADS, int RB, int MODE, bool RAGGED, bool K32 = false>
__global__ void __launch_bounds__(THREADS)
minimizers_reads_kernel(const uint8_t* __restrict__ bases, u64 total_bytes, const MmrGeom geo, u32 Lmax, u32 k, u32 w, u32 hk,
                        u64* __restrict__ out_word, u32* __restrict__ out_pos, unsigned long long* __restrict__ first_bad) {
    static_assert(THREADS == 16 * RB, "16 threads per read");   // (and a piece is at most 256 bases = 16 + 2 dwords: two per thread)
    using key_t = std::conditional_t<K32, u32, u64>;
    extern __shared__ __attribute__((aligned(16))) u64 hs[];   // keys [2][RB][NLS] (8 bytes each reserved, K32 uses half), then FW [RB][ND], RV [RB][ND] (u32), then the reads' geometry
    const u32 NLS = Lmax - w + 1u, span = k - w + 1u;
    const u32 ND = ((2u * Lmax + 31u) >> 5) + 2u;               // dwords of a staged read (+2: the field reads look ahead)
    u32* FW = reinterpret_cast<u32*>(hs + 2u * RB * NLS);
    u32* RV = FW + RB * ND;
    u64* SLOT = reinterpret_cast<u64*>(RV + RB * ND + ((RB * ND) & 1u));   // [RB] slot of the piece's first k-mer
    u32* LEN = reinterpret_cast<u32*>(SLOT + RB);               // [RB] bases of the piece
    u32* PBASE = LEN + RB;                                      // [RB] position of the piece inside its read
    u32* CUM = PBASE + RB;                                      // [RB + 1] k-mers of the pieces before
    const u32 r = threadIdx.x >> 4, j16 = threadIdx.x & 15u;
    const u64 n_pieces = geo.n_reads * geo.J;
    const uint8_t* const buf_end = bases + total_bytes;
    // where piece R lies (this thread's: r of the block's RB), and -- j16 == 0 -- its slot and position for the block
    struct Piece { const uint8_t* sp; u32 len; u64 slot; u32 pbase; };
    auto piece_of = [&](u64 R) -> Piece {
        Piece q{bases, 0u, 0ull, 0u};
        if (R >= n_pieces) return q;
        if constexpr (RAGGED) {
            const u64 o0 = geo.offsets[R], o1 = geo.offsets[R + 1u];
            q.sp = bases + o0;
            q.len = (u32)(o1 - o0 > (u64)Lmax ? 0u : o1 - o0);        // (a read above the bound: the caller took another kernel)
            q.slot = geo.win_offsets[R];
        } else {
            const u64 rd = geo.J == 1u ? R : R / geo.J;      // (whole reads: no 64-bit division per thread and iteration)
            const u32 j = (u32)(R - rd * geo.J);
            q.sp = bases + rd * (u64)geo.L + (u64)j * geo.T;
            const u32 left = geo.L - j * geo.T;                         // bases from the piece's first
            q.len = left < geo.T + k - 1u ? left : geo.T + k - 1u;
            q.slot = rd * (u64)(geo.L - k + 1u) + (u64)j * geo.T;
            q.pbase = j * geo.T;
        }
        return q;
    };
    // A thread packs the dwords j16 and j16 + 16 of its piece (a piece is at most 18 dwords).  Their bytes are REQUESTED one block
    // iteration ahead -- dword-aligned loads: a 16-byte load from an odd address is taken apart by the memory pipeline -- so that a
    // block never starts by waiting for memory (the 16 reads of an iteration are 2.4 KB: nothing else hides the latency).
    u32 xa[2][5];
    auto request = [&](const Piece& q) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32 d = j16 + 16u * h;
            const uint8_t* p = q.sp + 16u * d;
            const u32 rsh = (u32)(reinterpret_cast<uintptr_t>(p) & 3u);
            const bool fast = 16u * d < q.len && p + 20 <= buf_end && p - rsh >= bases;
            const u32* a4 = reinterpret_cast<const u32*>(fast ? p - rsh : bases - (reinterpret_cast<uintptr_t>(bases) & 3u) + 4u);
#pragma unroll
            for (int i = 0; i < 5; ++i) xa[h][i] = a4[i];
        }
    };
    Piece cur = piece_of((u64)blockIdx.x * RB + r);
    request(cur);
    for (u64 r0 = (u64)blockIdx.x * RB; r0 < n_pieces; r0 += (u64)gridDim.x * RB) {
        const u32 nr = (u32)(n_pieces - r0 < RB ? n_pieces - r0 : RB);
        key_t* A = reinterpret_cast<key_t*>(hs);
        key_t* B = reinterpret_cast<key_t*>(hs + RB * NLS);
        const uint8_t* const sp = cur.sp;
        const u32 len = cur.len;
        if (j16 == 0u) {
            SLOT[r] = cur.slot;
            PBASE[r] = cur.pbase;
            LEN[r] = len;
        }
        // ---- the piece, packed: dword d = its bases [16 d, 16 d + 16) (what follows it in the buffer comes along: never looked at)
        if (r < nr) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32 d = j16 + 16u * h;
                if (d >= ND) continue;
                u32 code = 0u;
                if (16u * d < len) {
                    const uint8_t* p = sp + 16u * d;
                    uint4 v;
                    const u32 rsh = (u32)(reinterpret_cast<uintptr_t>(p) & 3u);
                    if (p + 20 <= buf_end && p - rsh >= bases) {
                        v = make_uint4(__builtin_amdgcn_alignbyte(xa[h][1], xa[h][0], rsh), __builtin_amdgcn_alignbyte(xa[h][2], xa[h][1], rsh),
                                       __builtin_amdgcn_alignbyte(xa[h][3], xa[h][2], rsh), __builtin_amdgcn_alignbyte(xa[h][4], xa[h][3], rsh));
                    } else {                      // the batch's last bytes, one by one
                        u32 t[4] = {0u, 0u, 0u, 0u};
                        for (u32 b = 0; b < 16u && p + b < buf_end; ++b) t[b >> 2] |= (u32)p[b] << (8u * (b & 3u));
                        v = make_uint4(t[0], t[1], t[2], t[3]);
                    }
                    u32 inv16;
                    code = encode16_inv(v, inv16);
                    const u32 inside = len - 16u * d >= 16u ? 0xFFFFu : (1u << (len - 16u * d)) - 1u;
                    if ((inv16 & inside) != 0u) atomicMin(first_bad, (unsigned long long)((r0 + r) / geo.J));
                }
                FW[r * ND + d] = code;
            }
        }
        cur = piece_of(r0 + (u64)gridDim.x * RB + r);      // the next iteration's piece: its bytes are on their way while this one is worked on
        request(cur);
        __syncthreads();
        if (threadIdx.x == 0) {          // k-mers of the pieces before (RB additions)
            u32 c = 0;
            for (u32 i = 0; i < RB; ++i) {
                CUM[i] = c;
                const u32 l = LEN[i];
                c += l >= k ? l - k + 1u : 0u;
            }
            CUM[RB] = c;
        }
        if (MODE == 1 && r < nr) {
            // base-reversed copy: dword d holds bases len-1-16d-j (j = 0..15) = the 32 bits at base offset len-16d-16, group-reversed
            for (u32 d = j16; d < ND; d += 16u) {
                const int off = (int)len - 16 * (int)d - 16;      // may be negative: the piece has fewer bases left
                u32 x;
                if (off >= 0) x = (u32)mmr_field(FW + r * ND, 2u * (u32)off, 32u);
                else x = off > -16 ? FW[r * ND] << (2u * (u32)(-off)) : 0u;
                RV[r * ND + d] = revgroups32(x);
            }
        }
        if (MODE == 1) __syncthreads();
        const u32 NL = len >= w ? len - w + 1u : 0u;             // l-mers of the piece
        if (r < nr) {
            for (u32 p = j16; p < NL; p += 16u) {
                if constexpr (K32) {
                    u32 h;
                    if (MODE == 1) h = mmr_field32(RV + r * ND, 2u * (len - p - w), 2u * w);
                    else {
                        const u32 lm = mmr_field32(FW + r * ND, 2u * p, 2u * w);
                        h = MODE == 0 ? lm : revgroups32(lm) >> (32u - 2u * hk);      // (= lex_hash on a value of one dword)
                    }
                    A[r * NLS + p] = (h << 8) | p;
                } else {
                    u64 h;
                    if (MODE == 1) h = mmr_field(RV + r * ND, 2u * (len - p - w), 2u * w);
                    else {
                        const u64 lm = mmr_field(FW + r * ND, 2u * p, 2u * w);
                        h = MODE == 0 ? lm : lex_hash(lm, hk);
                    }
                    A[r * NLS + p] = (h << 8) | p;
                }
            }
        }
        __syncthreads();
        // minima over lenw consecutive keys: lenw x 4 per pass while that fits the span (two passes at span = 17 where doubling took four:
        // half the LDS writes and block barriers), then one doubling if there is room for it
        u32 lenw = 1;
        while (4u * lenw <= span) {
            if (r < nr) {
                for (u32 p = j16; p < NL; p += 16u) {
                    // (past the piece's last key: that key once more -- a minimum does not mind, and one v_min_u32 on the index is a third of
                    // a compare and two selects on a 64-bit key)
                    const key_t* const ap = A + r * NLS + p;
                    const u32 room = NL - 1u - p;
                    key_t m = ap[0];
                    const key_t b = ap[lenw < room ? lenw : room], c = ap[2u * lenw < room ? 2u * lenw : room], d = ap[3u * lenw < room ? 3u * lenw : room];
                    m = m < b ? m : b;
                    const key_t m2 = c < d ? c : d;
                    B[r * NLS + p] = m < m2 ? m : m2;
                }
            }
            __syncthreads();
            key_t* t = A; A = B; B = t;
            lenw *= 4u;
        }
        if (2u * lenw <= span) {
            if (r < nr) {
                for (u32 p = j16; p < NL; p += 16u) {
                    const key_t a = A[r * NLS + p];
                    const u32 room = NL - 1u - p;
                    const key_t b = A[r * NLS + p + (lenw < room ? lenw : room)];
                    B[r * NLS + p] = a < b ? a : b;
                }
            }
            __syncthreads();
            key_t* t = A; A = B; B = t;
            lenw *= 2u;
        }
        const u32 second = span - lenw;      // the window [i, i+span) = [i, i+lenw) u [i+second, i+second+lenw)
        const u32 total = CUM[RB];
        bool done = false;
        if constexpr (!RAGGED) {
            if (geo.J == 1u) {
                // reads of ONE length, whole: k-mer e of the block is k-mer e % Wk of its read e / Wk (a multiply instead of a search through
                // CUM), and the block's slots are one run from a wave-uniform first slot -- the stores take a 32-bit index (round 6: the
                // kernel is bound by VALU issue, 289 instructions per read: profiles/r06_pmc_minimizers.txt)
                const u32 Wk = geo.L - k + 1u, magic = Wk > 1u ? (u32)(0x100000000ull / Wk) + 1u : 0u;   // e < 16 x 256: e / Wk = umulhi(e, magic) (Wk = 1: e itself)
                u64* const ow = out_word + r0 * (u64)Wk;
                u32* const op = out_pos + r0 * (u64)Wk;
                for (u32 e = threadIdx.x; e < total; e += THREADS) {
                    const u32 rr = Wk > 1u ? __umulhi(e, magic) : e, i = e - rr * Wk;
                    const key_t a = A[rr * NLS + i], b = A[rr * NLS + i + second];
                    const key_t key = a < b ? a : b;
                    const u32 pos = (u32)(key & 0xFFu);
                    u64 word;
                    if constexpr (K32) word = MODE == 0 ? (u64)(key >> 8) : MODE == 1 ? (u64)(revgroups32((u32)(key >> 8)) >> (32u - 2u * w)) : (u64)mmr_field32(FW + rr * ND, 2u * pos, 2u * w);
                    else word = MODE == 0 ? (u64)(key >> 8) : MODE == 1 ? lex_hash((u64)(key >> 8), w) : mmr_field(FW + rr * ND, 2u * pos, 2u * w);
                    __builtin_nontemporal_store(word, &ow[e]);
                    __builtin_nontemporal_store(pos, &op[e]);
                }
                done = true;
            }
        }
        for (u32 e = threadIdx.x; !done && e < total; e += THREADS) {
            u32 rr = 0;      // the piece that holds k-mer e of the block (CUM is non-decreasing: four halvings)
#pragma unroll
            for (u32 step = RB / 2u; step != 0u; step >>= 1) rr += e >= CUM[rr + step] ? step : 0u;
            const u32 i = e - CUM[rr];
            const key_t a = A[rr * NLS + i], b = A[rr * NLS + i + second];
            const key_t key = a < b ? a : b;
            const u32 pos = (u32)(key & 0xFFu);
            const u64 slot = SLOT[rr] + i;
            // the l-mer itself: the key's hash IS it (identity), or it with its bases reversed (LexHasher(w) on w bases)
            u64 word;
            if constexpr (K32) word = MODE == 0 ? (u64)(key >> 8) : MODE == 1 ? (u64)(revgroups32((u32)(key >> 8)) >> (32u - 2u * w)) : (u64)mmr_field32(FW + rr * ND, 2u * pos, 2u * w);
            else word = MODE == 0 ? (u64)(key >> 8) : MODE == 1 ? lex_hash((u64)(key >> 8), w) : mmr_field(FW + rr * ND, 2u * pos, 2u * w);
            __builtin_nontemporal_store(word, &out_word[slot]);
            __builtin_nontemporal_store(pos + PBASE[rr], &out_pos[slot]);
        }
        __syncthreads();
    }
}

// Any read, any (k, w, hasher): a wave per read, a lane per k-mer, the window's l-mers evaluated one after the other from the bytes
// (leftmost minimum).  What the tiled kernel does not take: reads above 256 bases behind offsets, hashes above 56 bits.
__global__ void __launch_bounds__(256)
minimizers_reads_generic_kernel(const uint8_t* __restrict__ bases, const MmrGeom geo, u32 k, u32 w, u32 hasher, u32 hk,
                                u64* __restrict__ out_word, u32* __restrict__ out_pos, unsigned long long* __restrict__ first_bad) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 R = wave; R < geo.n_reads; R += n_waves) {
        u64 o0 = R * (u64)geo.L, len = geo.L, slot = R * (u64)(geo.L - k + 1u);
        if (geo.offsets) {
            o0 = geo.offsets[R];
            len = geo.offsets[R + 1u] - o0;
            slot = geo.win_offsets[R];
        }
        const uint8_t* s = bases + o0;
        bool bad = false;
        for (u64 i = lane; i < len; i += 64u) bad |= encode_base(s[i]) >= 4u;
        if (__any(bad) && lane == 0) atomicMin(first_bad, (unsigned long long)R);
        if (len < k) continue;
        for (u64 i = lane; i + k <= len; i += 64u) {
            u64 best = 0, best_h = 0;
            u32 best_p = 0;
            for (u32 p = 0; p + w <= k; ++p) {
                u64 lm = 0;
                for (u32 b = 0; b < w; ++b) {
                    const u32 c = s[i + p + b];
                    const u32 ic = (c >> 1) & 3u;
                    lm |= (u64)(ic ^ (ic >> 1)) << (2u * b);
                }
                const u64 h = hasher == KMX_HASH_LEX ? lex_hash(lm, hk) : lm;
                if (p == 0u || h < best_h) {
                    best = lm;
                    best_h = h;
                    best_p = p;
                }
            }
            out_word[slot + i] = best;
            out_pos[slot + i] = (u32)(i + best_p);
        }
    }
}

// bound: the longest read (uniform: L).  *tiled: whether the sliding-minimum kernel ran (the caller's diagnostics / tests)
hipError_t launch_minimizers_reads(const uint8_t* bases, u64 total_bytes, const u64* offsets, const u64* win_offsets, u64 n_reads, u32 L,
                                   u32 bound, u32 k, u32 w, u32 hasher, u32 hk, u64* out_word, u32* out_pos,
                                   unsigned long long* first_bad, int n_cu, hipStream_t st, bool* tiled) {
    MmrGeom geo{offsets, win_offsets, n_reads, L, 1u, 0u};
    const u32 hash_bits = hasher == KMX_HASH_LEX ? 2u * hk : 2u * w;
    *tiled = false;
    const bool keys_fit = hash_bits <= 56u && w <= 28u && k > w;
    u32 Lmax = bound;
    if (!offsets && L > 256u) {          // pieces of at most 256 bases: T k-mers each, the same for every read
        geo.T = 256u - k + 1u;
        geo.J = (L - k + 1u + geo.T - 1u) / geo.T;
        Lmax = 256u;
    } else if (!offsets) {
        geo.T = L - k + 1u;
    }
    if (keys_fit && Lmax <= 256u && Lmax >= k && n_reads < (1ull << 40)) {
        constexpr int RB = 16;
        const u64 n_pieces = n_reads * geo.J;
        u64 grid = (n_pieces + RB - 1u) / RB;
        const u64 cap = (u64)n_cu * 8u;
        if (grid > cap) grid = cap;
        const u32 NLS = Lmax - w + 1u, ND = ((2u * Lmax + 31u) >> 5) + 2u;
        const size_t lds = (size_t)2u * RB * NLS * 8u + (size_t)(2u * RB * ND + ((RB * ND) & 1u)) * 4u + (size_t)RB * 8u + (size_t)(3u * RB + 1u) * 4u + 16u;
        const int mode = hasher != KMX_HASH_LEX ? 0 : (hk == w ? 1 : 2);
        const bool k32 = hash_bits + 8u <= 32u && 2u * w + 8u <= 32u;   // hash and position in one dword (and the l-mer itself in 24 bits)
#define KMX_MMR_LAUNCH(M, RG)                                                                                                          \
    do {                                                                                                                               \
        auto kern = k32 ? minimizers_reads_kernel<256, RB, M, RG, true> : minimizers_reads_kernel<256, RB, M, RG, false>;              \
        if (lds > 64u * 1024u) {                                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                                             \
        }                                                                                                                              \
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid ? grid : 1)), dim3(256), lds, st, bases, total_bytes, geo, Lmax, k, w, hk, out_word, \
                           out_pos, first_bad);                                                                                        \
    } while (0)
        if (offsets) {
            if (mode == 0) KMX_MMR_LAUNCH(0, true); else if (mode == 1) KMX_MMR_LAUNCH(1, true); else KMX_MMR_LAUNCH(2, true);
        } else {
            if (mode == 0) KMX_MMR_LAUNCH(0, false); else if (mode == 1) KMX_MMR_LAUNCH(1, false); else KMX_MMR_LAUNCH(2, false);
        }
#undef KMX_MMR_LAUNCH
        *tiled = true;
        return hipGetLastError();
    }
    u64 grid = (n_reads + 3u) / 4u;
    const u64 cap = (u64)n_cu * 16u;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(minimizers_reads_generic_kernel, dim3((unsigned)(grid ? grid : 1)), dim3(256), 0, st, bases, geo, k, w, hasher, hk, out_word,
                       out_pos, first_bad);
    return hipGetLastError();
}

}  // namespace kmx
