// kmx_scan.hip -- K1: fused encode + sliding window + reverse complement + canonical (+hash)
// + reduce over uniform-length reads.  The headline kernel (BASELINE.json: canonical
// k-mers/s at k=31 on 150 bp reads).
//
// Replaces, per read, the reference's streaming loop
//   CanonicalKmerIterator::find_next      src/naive_impl/canonical_kmer_iterator.rs:42-70
//   CanonicalKmer::append_base            src/naive_impl/canonical_kmer.rs:90-94
//   Kmer::append_base / prepend_base      src/naive_impl/kmer.rs:91-102
//   CanonicalKmer::get_canonical_word     src/naive_impl/canonical_kmer.rs:113-119
//   encode_binary_u8                      src/naive_impl/mod.rs:40-50
//   LexHasher::write_u64 (optional)       src/naive_impl/hash.rs:60-71
//
// gfx950 design (one wave owns a tile of 64 reads, no block-level barrier anywhere):
//   1. the wave streams its tile (64*L contiguous bytes) from HBM with 16 B/lane
//      global_load_dwordx4 (1 KiB per wave instruction, fully coalesced);
//   2. each 16-byte chunk is packed to 16 two-bit bases (one dword) with v_dot4_u32_u8
//      and checked for non-ACGTacgt bytes with v_perm_b32 (exact); the packed tile
//      (4x smaller) is staged in the wave's private LDS slice;
//   3. each lane pulls ITS read's packed words back from LDS (ds_read_b32, stride ~L/16
//      dwords, conflict-light) and realigns them with v_alignbit_b32 into a forward word
//      array F and -- via v_bfrev_b32 -- a reverse-complement array G, both in VGPRs;
//   4. every window is then two funnel shifts per strand with compile-time shift amounts
//      (v_alignbit_b32), one 64-bit compare, two v_cndmask and a 64-bit add: ~11 VALU
//      lane-ops per canonical k-mer.  No per-base rolling, no per-window branches.
//   A tile that contains any invalid byte (or the final partial tile) takes the
//   reference-shaped per-lane rolling path instead (roll_read) -- rare on real reads,
//   and bit-exact with the iterator's skip semantics.
//
// Roofline: HBM-read bound by construction (L bytes read per read, ~0 written); the VALU
// budget at 5.6 TB/s is ~17 lane-ops per k-mer, which is what step 4 is sized for.
#include "kmx_device.h"

namespace kmx {

// One window, two-dword k-mer (k in 18..31).  SF/SR: forward / reverse funnel-shift amounts.
template <bool FULL>
__device__ __forceinline__ void window2(u32 f0, u32 f1, u32 f2, u32 g0, u32 g1, u32 g2, u32 sf, u32 sr, u32 mhi,
                                        u64 maskk, Acc& acc) {
    const u32 fw_lo = alignbit(f1, f0, sf);
    const u32 fw_hi = alignbit(f2, f1, sf) & mhi;
    const u32 rc_lo = alignbit(g1, g0, sr);
    const u32 rc_hi = alignbit(g2, g1, sr) & mhi;
    const u64 fw = ((u64)fw_hi << 32) | fw_lo;
    const u64 rc = ((u64)rc_hi << 32) | rc_lo;
    const u64 canon = fw < rc ? fw : rc;  // canonical_kmer.rs:113-119
    acc.sum_canon += canon;
    if (FULL) {
        // LexHasher(k)(canon) = MASK[k] & ~max(fw,rc): the 2-bit-group reversal of x is the
        // complement of revcomp(x) inside 2k bits (hash.rs:60-71 vs kmer.rs:124-136).
        acc.xor_hash ^= maskk ^ fw ^ rc ^ canon;
        acc.sum_fw += fw;
    }
}

// One window, single-dword k-mer (k in 2..16)
template <bool FULL>
__device__ __forceinline__ void window1(u32 f0, u32 f1, u32 g0, u32 g1, u32 sf, u32 sr, u32 mlo, Acc& acc) {
    const u32 fw = alignbit(f1, f0, sf) & mlo;
    const u32 rc = alignbit(g1, g0, sr) & mlo;
    const u32 canon = fw < rc ? fw : rc;
    acc.sum_canon += canon;
    if (FULL) {
        acc.xor_hash ^= (u64)(mlo ^ fw ^ rc ^ canon);
        acc.sum_fw += fw;
    }
}

// NW  = packed dwords per read = ceil(L/16) rounded up to an instantiated size (L <= 16*NW)
// V   = 1: k in [2,17]   2: k in [18,32]   (fixes the static register index of the rc window)
// DW  = dwords per k-mer (1: k<=16, 2: k>=17)
// FULL= also fold LexHasher(k) xor and the forward-word sum
template <int NW, int V, int DW, bool FULL>
__global__ void __launch_bounds__(256)
scan_uniform_kernel(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 k, u32 want_hash, u32 want_sumfw,
                    kmx_summary* __restrict__ out, unsigned long long* __restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 lane = threadIdx.x & 63u;
    const u32 wib = threadIdx.x >> 6;
    const u32 chunks = 4u * L;                      // 16-byte chunks per 64-read tile
    const u32 ldsw = (chunks + 1u + 6u + 3u) & ~3u; // front pad 1, tail pad >= 6
    u32* P = lds + wib * ldsw;

    const u64 n_full = n_reads >> 6;
    const u64 n_waves = (u64)gridDim.x * 4u;
    const u64 wave_id = (u64)blockIdx.x * 4u + wib;

    // per-lane alignment of this lane's read inside the packed tile (LDS index 1+c holds bases [16c,16c+16))
    const u32 posF = lane * L + 16u;
    const u32 qF = posF >> 4, aF = 2u * (posF & 15u);
    const u32 delta = (1u - k) & 15u;  // rc stream pre-offset so that rc sub-shift == 30-2s
    const u32 posR = posF - delta;
    const u32 qR = posR >> 4, aR = 2u * (posR & 15u);

    const u32 omax = L - k;  // last window start
    const u32 imax = omax >> 4, smax = omax & 15u;
    const u64 maskk = mask2k(k);
    const u32 mlo = (u32)maskk;
    const u32 mhi = (u32)(maskk >> 32);
    const u32 nwin = omax + 1u;

    Acc acc;

    // dynamic tile queue (see kmx_bitslice.hip): NQ interleaved heads, one tile per ticket, ticket fetched one
    // tile ahead; removes the under-occupied tail that static striding leaves behind
    constexpr u32 NQ = 32;
    u32 qid = (blockIdx.x & 255u) >> 3;
    u32 heads_left = NQ;
    auto dequeue = [&]() -> u64 {
        while (heads_left != 0u) {
            unsigned long long v = 0;
            if (lane == 0) v = atomicAdd(queue + qid * 16u, 1ull);
            const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
            const u64 t = (((u64)hi << 32) | lo) * NQ + qid;
            if (t < n_full) return t;
            qid = (qid + 1u) & (NQ - 1u);
            heads_left -= 1u;
        }
        return ~0ull;
    };
    u64 next_tile = dequeue();
    for (u64 tile = next_tile; tile < n_full; tile = next_tile) {
        next_tile = dequeue();
        const uint4* __restrict__ tb = reinterpret_cast<const uint4*>(bases + tile * 64u * (u64)L);
        // ---- 1. stream the tile: all loads in flight before the first use
        uint4 w[NW];
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const u32 c = it * 64u + lane;
            if (c < chunks) w[it] = tb[c];
        }
        // ---- 2. pack + validate, stage packed words in LDS
        u32 bad = 0;
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const u32 c = it * 64u + lane;
            if (c < chunks) P[1u + c] = encode16(w[it], bad);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (__any(chunk_has_invalid(bad))) {
            // ---- rare: a non-ACGTacgt byte somewhere in this tile -> exact iterator semantics
            const uint8_t* s = bases + (tile * 64u + lane) * (u64)L;
            roll_read(s, L, k, [&](u32, u64 fw, u64 rc) {
                const u64 canon = fw < rc ? fw : rc;
                acc.n_valid += 1;
                acc.sum_canon += canon;
                if (FULL) {
                    acc.xor_hash ^= lex_hash(canon, k);
                    acc.sum_fw += fw;
                }
            });
            continue;
        }

        // ---- 3. this lane's read: forward words F, reverse-complement words G
        u32 F[NW + 2], G[NW + 2];
        {
            u32 R[NW + 1];
#pragma unroll
            for (int j = 0; j <= NW; ++j) R[j] = P[qF + j];
#pragma unroll
            for (int i = 0; i < NW; ++i) F[i] = alignbit(R[i + 1], R[i], aF);
            F[NW] = 0;
            F[NW + 1] = 0;
            u32 Rr[NW + 2];
#pragma unroll
            for (int j = 0; j <= NW + 1; ++j) Rr[j] = P[qR + j];
#pragma unroll
            for (int m = 0; m <= NW; ++m) G[m] = revgroups32(~alignbit(Rr[NW - m + 1], Rr[NW - m], aR));
            G[NW + 1] = 0;
        }

        // ---- 4. windows: o = 16*i + s;  fw from F[i..i+2] >> 2s;  rc from G[M..M+2] >> (30-2s), M = NW-V-i
#pragma unroll
        for (int i = 0; i <= NW - V; ++i) {
            constexpr int dummy = 0;
            (void)dummy;
            const int M = NW - V - i;
            if ((u32)i < imax) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (DW == 2)
                        window2<FULL>(F[i], F[i + 1], F[i + 2], G[M], G[M + 1], G[M + 2], 2 * s, 30 - 2 * s, mhi, maskk, acc);
                    else
                        window1<FULL>(F[i], F[i + 1], G[M], G[M + 1], 2 * s, 30 - 2 * s, mlo, acc);
                }
            } else if ((u32)i == imax) {
                for (u32 s = 0; s <= smax; ++s) {
                    if (DW == 2)
                        window2<FULL>(F[i], F[i + 1], F[i + 2], G[M], G[M + 1], G[M + 2], 2u * s, 30u - 2u * s, mhi, maskk, acc);
                    else
                        window1<FULL>(F[i], F[i + 1], G[M], G[M + 1], 2u * s, 30u - 2u * s, mlo, acc);
                }
            }
        }
        acc.n_valid += nwin;
    }

    // ---- final partial tile (n_reads % 64 reads): per-lane rolling
    const u32 rem = (u32)(n_reads & 63u);
    if (rem != 0u && wave_id == 0 && lane < rem) {
        const uint8_t* s = bases + (n_full * 64u + lane) * (u64)L;
        roll_read(s, L, k, [&](u32, u64 fw, u64 rc) {
            const u64 canon = fw < rc ? fw : rc;
            acc.n_valid += 1;
            acc.sum_canon += canon;
            if (FULL) {
                acc.xor_hash ^= lex_hash(canon, k);
                acc.sum_fw += fw;
            }
        });
    }

    flush_acc(acc, out, FULL && want_hash, FULL && want_sumfw);
}

// ------------------------------------------------------------------ launcher

struct ScanCfg {
    const void* fn;
    int blocks_per_cu;  // cached occupancy
};

template <int NW, int V, int DW, bool FULL>
static hipError_t launch_one(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 want_hash, u32 want_sumfw,
                             kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream) {
    auto kern = scan_uniform_kernel<NW, V, DW, FULL>;
    const u32 chunks = 4u * L;
    const u32 ldsw = (chunks + 1u + 6u + 3u) & ~3u;
    const size_t lds_bytes = (size_t)ldsw * 4u * 4u;
    static int bpc = 0;
    static size_t bpc_lds = 0;
    if (bpc == 0 || bpc_lds != lds_bytes) {
        int b = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kern, 256, lds_bytes);
        if (e != hipSuccess) return e;
        bpc = b > 0 ? b : 1;
        bpc_lds = lds_bytes;
    }
    const u64 n_tiles = (n_reads + 63u) >> 6;
    u64 grid = (u64)n_cu * (u64)bpc;
    const u64 need = (n_tiles + 3u) / 4u;
    if (grid > need) grid = need;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds_bytes, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue);
    return hipGetLastError();
}

// Returns hipSuccess and sets *handled=false when (L,k) is outside the fast kernel's domain.
hipError_t launch_scan_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw,
                               kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled) {
    *handled = false;
    if (k < 2 || k > 31 || k == 17 || L < k || L > 256 || (reinterpret_cast<uintptr_t>(bases) & 15u)) return hipSuccess;
    if (n_reads * (u64)L >= (1ull << 62)) return hipSuccess;
    *handled = true;
    const bool big = L > 160;
    const bool full = want_hash || want_sumfw;
#define KMX_GO(NW, V, DW)                                                                              \
    return full ? launch_one<NW, V, DW, true>(bases, n_reads, L, k, want_hash, want_sumfw, out, queue, n_cu, stream) \
                : launch_one<NW, V, DW, false>(bases, n_reads, L, k, 0, 0, out, queue, n_cu, stream)
    if (k <= 16) {
        if (big) { KMX_GO(16, 1, 1); } else { KMX_GO(10, 1, 1); }
    } else {
        if (big) { KMX_GO(16, 2, 2); } else { KMX_GO(10, 2, 2); }
    }
#undef KMX_GO
}

}  // namespace kmx
