// kmx_generic.hip -- reference-shaped kernels for everything outside the fast uniform scan:
// ragged reads (offsets), k in {1,17}, mis-aligned buffers, per-window materialisation,
// the [u64;2] (k in 33..64) extension and the bucket histogram.  One lane walks one read
// exactly like CanonicalKmerIterator::find_next (src/naive_impl/canonical_kmer_iterator.rs:42-70);
// correctness-first, these are not the roofline kernels.
#include "kmx_device.h"

// (the line-aligned stores of the tiled [u64;2] write-back carry the nt hint (k = 33 / 64: 5.5 -> 4.7 / 4.7 -> 4.0 ms per 2e7 reads, k = 47 / 63 +2 %: profiles/r03_nt_stores.txt)
namespace kmx {

struct ReadsView {
    const uint8_t* bases;
    u64 n_reads;
    u32 read_len;
    const u64* offsets;
    unsigned long long* too_long;   // the context's sticky "a read of 2^31 bases or more was skipped" flag
    __device__ __forceinline__ void span(u64 r, const uint8_t*& s, u32& len) const {
        if (offsets) {
            const u64 a = offsets[r], b = offsets[r + 1];
            s = bases + a;
            len = read_too_long(b - a, too_long) ? 0u : (u32)(b - a);
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
reduce2_generic_kernel(ReadsView rv, u32 k, u32 with_hash, kmx_summary2* __restrict__ out, const u32* __restrict__ gate) {
    // (gate: kmx_canonical_reduce2 deciding on the device whether the reads behind an offsets array are uniform; 1 = they are,
    // and the tiled uniform kernel launched beside this one scans them)
    if (gate != nullptr && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u) return;
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

// ------------------------------------------------------------------ [u64;2] materialise, tiled (round 3)
// kmx_canonical_windows2 on uniform reads of up to 16*NW bases: a wave takes 64 reads at a time.  The tile is packed to
// 2-bit codes in the wave's LDS slice (encode16: coalesced 16-byte loads, exact validation), lane r realigns ITS read from
// there, and the 128-bit window words come out of two rolling register windows -- the forward dwords F[i..i+4] and the
// dwords H[..] of the read's reverse complement, pre-shifted so that window 16i+s is a compile-time funnel shift of both
// (Kmer::<u64,K,2>::new, /root/reference/src/kmer.rs:21-28,67-69; the [u64;2] order / rolling are BUILD-DEFINED, kmx.h).
// Every lane writes its read's slots in order, 16 bytes per array and window: a read's W slots are contiguous, so 8
// consecutive windows of a lane fill one 128-byte line.  The lane-per-read kernel above walks base by base (8 bases per
// load, ~35 instructions per base) and wrote 1.3-1.5 TB/s; a tile with an invalid byte and the final partial tile still go
// through its body (one_read below): exact iterator semantics.
struct Win2Out {
    u64* fw;
    u64* rc;
    u64* canon;
    uint8_t* flags;
};

__device__ __forceinline__ void windows2_one_read(const uint8_t* s, u32 len, u32 k, u64 base, const Win2Out& o) {
    const u32 nwin = len >= k ? len - k + 1u : 0u;
    u32 next = 0;
    auto zero_to = [&](u32 end) {
        for (; next < end; ++next) {
            const u64 j = 2u * (base + next);
            if (o.fw) o.fw[j] = o.fw[j + 1] = 0;
            if (o.rc) o.rc[j] = o.rc[j + 1] = 0;
            if (o.canon) o.canon[j] = o.canon[j + 1] = 0;
            if (o.flags) o.flags[base + next] = 0;
        }
    };
    roll_read2(s, len, k, [&](u32 pos, U128 fw, U128 rc) {
        zero_to(pos);
        const bool lt = lt128(fw, rc);
        const U128 c = lt ? fw : rc;
        const u64 j = 2u * (base + pos);
        if (o.fw) { o.fw[j] = fw.lo; o.fw[j + 1] = fw.hi; }
        if (o.rc) { o.rc[j] = rc.lo; o.rc[j + 1] = rc.hi; }
        if (o.canon) { o.canon[j] = c.lo; o.canon[j + 1] = c.hi; }
        if (o.flags) o.flags[base + pos] = (uint8_t)(KMX_WIN_VALID | (lt ? KMX_WIN_FW_CANONICAL : 0u));
        next = pos + 1u;
    });
    zero_to(nwin);
}

// STAGE (one word array asked for -- the usual call): the 16 bytes of a window go to a lane-major staging block in LDS, and
// after every 8 windows the wave writes them back transposed -- 8 lanes cover one read's 128 contiguous bytes, a store
// instruction covers 8 such runs -- instead of 64 separate 16-byte pieces 16*W bytes apart (2.1 -> see DESIGN 4.4 TB/s).
// RAGGED (round 4): reads behind an offsets array, output slots from win_offsets.  A tile is still 64 consecutive reads = one
// contiguous byte span, streamed from its 16-byte aligned start; a lane carries its own start, window count and first slot, the
// write-back takes every read's line boundaries from its own first slot.  A tile whose span or longest read leaves the frame
// (L = the caller's length bound, at most 16 NW), or that would load past the end of the buffer, takes the per-read path.
template <int NW, bool STAGE, bool RAGGED = false>
__global__ void __launch_bounds__(256)
windows2_tiled_kernel(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 k, Win2Out out, u32 lead,
                      const u64* __restrict__ offsets, const u64* __restrict__ win_offsets, unsigned long long* __restrict__ too_long,
                      const u64* __restrict__ ends_arg, unsigned long long* __restrict__ queue) {
    // `queue` (round 6; nullptr: none): the context's queue block.  A tile with an invalid byte then stays on the tiled path -- all its reads
    // marked, one store, as the single-word window sinks do (kmx_scan_kernel.h, SinkMarksCoarse) -- and the ZERO sweep behind the passes
    // (kmx_sweep.hip) writes the spoiled windows' slots as the iterator leaves them; without it such a tile takes the per-read path.
    // `lead`: `bases` is the 16-byte aligned address at or below the first read, which starts `lead` bytes in
    // RAGGED: read r = bases[offsets[r], ends[r]); ends_arg == nullptr: back to back (ends = offsets + 1); a separate array serves
    // reads that overlap in memory -- the segments a long uniform read is planned as (kmx_segments.hip)
    [[maybe_unused]] const u64* __restrict__ const ends = RAGGED ? (ends_arg ? ends_arg : offsets + 1) : nullptr;
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    const u32 chunks_cap = 4u * L + ((RAGGED || lead != 0u) ? 1u : 0u);      // 16-byte chunks of a 64-read tile (ragged: the most a tile may span)
    u32 chunks = chunks_cap;
    const u32 ldsw = (chunks_cap + 1u + (u32)NW + 8u + 3u) & ~3u;   // front pad 1, tail pad: the last read's frame (and 6 dwords more) may lie past the tile
    constexpr u32 SPITCH = 68u;                               // dwords per lane of the staging ring: 16 windows x 4 + 4 (bank spread)
    constexpr u32 STG = STAGE ? 64u * SPITCH : 0u;
    u32* const P = lds + wib * (ldsw + STG);
    u32* const S = P + ldsw;                                  // [64 lanes][SPITCH]
    const u64 n_full = n_reads >> 6;
    const u32 W = L - k + 1u;
    u32 omax = L - k;              // the last window of the tile's longest read (ragged: per tile)
    u32 nwin = W;                  // windows of this lane's read
    const u64 total_bytes = RAGGED ? ends[n_reads - 1u] : 0;
    // the one array of a STAGE launch, and which words it takes
    u64* const one = out.fw ? out.fw : out.rc ? out.rc : out.canon;
    const u32 which = out.fw ? 0u : out.rc ? 1u : 2u;
    const u64 waves = (u64)gridDim.x * 4u, wave0 = (u64)blockIdx.x * 4u + wib;
    // masks of the two upper dwords of a 2k-bit value (k in 33..64: 66..128 bits)
    const u32 kb = 2u * k;
    const u32 m2 = kb >= 96u ? ~0u : ((1u << (kb - 64u)) - 1u);
    const u32 m3 = kb >= 128u ? ~0u : (kb > 96u ? ((1u << (kb - 96u)) - 1u) : 0u);
    // this lane's read inside the packed tile: LDS dword 1 + c holds bases [16c, 16c + 16) of the tile
    u32 posF = lane * L + lead + 16u;
    u32 qF = posF >> 4, aF = 2u * (posF & 15u);
    // reverse complement of the read's 16*NW-base frame: Gfull[j] = revgroups(~F[NW-1-j]); the rc word of window o starts at base
    // NF - k - o of it.  H = Gfull moved down by e = (NF - k) & 15 bases: window 16 i + s then starts at bit 2 (16 - s) of
    // H[Q - i - 1] (s > 0) or is H[Q - i ..] itself (s = 0), Q = (NF - k) >> 4.
    constexpr u32 NF = 16u * NW;
    const u32 e2 = 2u * ((NF - k) & 15u);
    const int Q = (int)((NF - k) >> 4);
    auto Fat = [&](int x) -> u32 {      // aligned forward dword x of this lane's read (x may run a few dwords past the frame: padded)
        return alignbit(P[qF + (u32)x + 1u], P[qF + (u32)x], aF);
    };
    auto Gat = [&](int m) -> u32 {      // Gfull[m], m in [-1, NW]: outside the frame nothing of it is ever used -- any value
        const int x = NW - 1 - m;
        return (x < 0 || x > NW + 4) ? 0u : revgroups32(~Fat(x));
    };
    auto Hat = [&](int m) -> u32 { return alignbit(Gat(m + 1), Gat(m), e2); };

    for (u64 t = wave0; t < n_full; t += waves) {
        // ---- A. the tile's chunks: 16 bytes per lane and row, packed + validated into the LDS slice
        const uint8_t* __restrict__ tb = bases + t * 64u * (u64)L;
        u64 my_off = 0;
        u32 my_len = L;
        u64 slot0 = (t * 64u + lane) * (u64)W;
        bool fits = true;
        if constexpr (RAGGED) {
            const u64 o0 = offsets[t * 64u + lane], o1 = ends[t * 64u + lane];
            my_off = o0;
            my_len = read_too_long(o1 - o0, too_long) ? 0u : (u32)(o1 - o0);    // (not materialised; kmx_ctx_synchronize reports it)
            slot0 = win_offsets[t * 64u + lane];
            const u32 t0l = __builtin_amdgcn_readfirstlane((u32)o0), t0h = __builtin_amdgcn_readfirstlane((u32)(o0 >> 32));
            const u32 t1l = __builtin_amdgcn_readlane((u32)o1, 63), t1h = __builtin_amdgcn_readlane((u32)(o1 >> 32), 63);
            const u64 t0 = ((u64)t0h << 32) | t0l, t1 = ((u64)t1h << 32) | t1l;
            const u64 base_al = t0 & ~15ull, n_ch = (t1 - base_al + 15u) >> 4;
            const u32 max_len = wave_max_u32(my_len);
            fits = n_ch <= (u64)chunks_cap && max_len <= 16u * (u32)NW && max_len <= L && base_al + 16u * n_ch <= total_bytes && !__any(o1 - o0 > 0x7FFFFFFFull);
            chunks = fits ? (u32)n_ch : 0u;
            tb = bases + base_al;
            posF = (u32)(o0 - base_al) + 16u;
            qF = posF >> 4;
            aF = 2u * (posF & 15u);
            nwin = my_len >= k ? my_len - k + 1u : 0u;
            const u32 nw_max = wave_max_u32(nwin);
            if (nw_max == 0u) continue;     // no read of the tile holds a window
            omax = nw_max - 1u;
        }
        u32 bad = 0;
#pragma unroll
        for (int it = 0; it < NW + 1; ++it) {
            const u32 c = (u32)it * 64u + lane;
            if (c < chunks) {
                typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(tb + 16u * (u64)c));
                P[1u + c] = encode16(make_uint4(v.x, v.y, v.z, v.w), bad);
            }
        }
        if (lane < (u32)NW + 8u) P[1u + chunks + lane] = 0u;   // the pad behind the tile
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const u64 read = t * 64u + lane;
        bool per_read = !fits;
        if (!per_read && __any(chunk_has_invalid(bad))) {
            const unsigned long long dq = queue ? queue[515] : 0ull;
            u64* const dmk = reinterpret_cast<u64*>(((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(dq >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((u32)dq));
            if (dmk != nullptr) {
                if (lane == 0) {
                    dmk[t] = ~0ull;
                    queue[512] = 1ull << 40;      // ("many": a plain store every marking wave agrees on)
                }
            } else {
                per_read = true;
            }
        }
        if (per_read) {   // (wave-uniform) exact iterator semantics, one lane per read
            windows2_one_read(RAGGED ? bases + my_off : bases + lead + read * (u64)L, my_len, k, slot0, out);
        } else {
            // rolling windows: f[j] = F[i + j], h[j] = H[Q - i - 1 + j], j = 0..4
            u32 f[5], h[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) f[j] = Fat(j);
#pragma unroll
            for (int j = 0; j < 5; ++j) h[j] = Hat(Q - 1 + j);
            u32 gprev = Gat(Q - 1);   // Gfull[Q - i - 1]: the upper source of the next H
            u32 fl4 = 0;              // the flags of up to four consecutive windows
            [[maybe_unused]] const u32 a_own = (u32)slot0 & 15u;   // STAGE: this read's first slot mod 16
            // (both loops stay loops: unrolled -- 16 NW window bodies, each with its write-back -- hipcc hoists the address
            // arithmetic of all of them and the kernel takes 256 registers; as loops it takes ~145 and the shifts are VGPR operands)
#pragma unroll 1
            for (int i = 0; i < NW; ++i) {
                if (16u * (u32)i > omax) break;
#pragma unroll 1
                for (int s = 0; s < 16; ++s) {
                    const u32 o = 16u * (u32)i + (u32)s;
                    if (o > omax) break;
                    u32 a0, a1, a2, a3, r0, r1, r2, r3;
                    if (s == 0) {
                        a0 = f[0]; a1 = f[1]; a2 = f[2]; a3 = f[3];
                        r0 = h[1]; r1 = h[2]; r2 = h[3]; r3 = h[4];
                    } else {
                        a0 = alignbit(f[1], f[0], 2u * (u32)s);
                        a1 = alignbit(f[2], f[1], 2u * (u32)s);
                        a2 = alignbit(f[3], f[2], 2u * (u32)s);
                        a3 = alignbit(f[4], f[3], 2u * (u32)s);
                        r0 = alignbit(h[1], h[0], 2u * (16u - (u32)s));
                        r1 = alignbit(h[2], h[1], 2u * (16u - (u32)s));
                        r2 = alignbit(h[3], h[2], 2u * (16u - (u32)s));
                        r3 = alignbit(h[4], h[3], 2u * (16u - (u32)s));
                    }
                    a2 &= m2; a3 &= m3; r2 &= m2; r3 &= m3;
                    const u64 fhi = ((u64)a3 << 32) | a2, flo = ((u64)a1 << 32) | a0;
                    const u64 rhi = ((u64)r3 << 32) | r2, rlo = ((u64)r1 << 32) | r0;
                    const bool lt = fhi < rhi || (fhi == rhi && flo < rlo);
                    const u64 j = 2u * (slot0 + o);
                    const bool act = !RAGGED || o < nwin;      // (ragged: a shorter read's lane idles through the longest read's windows)
                    if constexpr (STAGE) {
                        // ring slot = the window's ABSOLUTE output slot mod 16: a 128-byte line of the output array is one
                        // aligned half of the ring, whatever W is
                        const bool takef = which == 0u || (which == 2u && lt);
                        if (act)
                            *reinterpret_cast<uint4*>(S + lane * SPITCH + 4u * ((a_own + (u32)s) & 15u)) =
                                takef ? make_uint4(a0, a1, a2, a3) : make_uint4(r0, r1, r2, r3);
                    } else if (act) {
                        if (out.fw) *reinterpret_cast<uint4*>(out.fw + j) = make_uint4(a0, a1, a2, a3);
                        if (out.rc) *reinterpret_cast<uint4*>(out.rc + j) = make_uint4(r0, r1, r2, r3);
                        if (out.canon) *reinterpret_cast<uint4*>(out.canon + j) = lt ? make_uint4(a0, a1, a2, a3) : make_uint4(r0, r1, r2, r3);
                    }
                    if (out.flags && act) fl4 |= (KMX_WIN_VALID | (lt ? KMX_WIN_FW_CANONICAL : 0u)) << (8u * ((u32)s & 3u));
                    // four flags per store (one byte each: the slots of a read are contiguous; the address need not be aligned)
                    if (act && (((u32)s & 3u) == 3u || o + 1u == nwin)) {
                        if (out.flags) {
                            uint8_t* const fp = out.flags + slot0 + (o & ~3u);
                            const u32 nfl = (o & 3u) + 1u;
                            if (nfl == 4u) {
                                typedef u32 u32_unaligned __attribute__((aligned(1)));
                                *reinterpret_cast<u32_unaligned*>(fp) = fl4;
                            } else {
                                for (u32 q = 0; q < nfl; ++q) fp[q] = (uint8_t)(fl4 >> (8u * q));
                            }
                        }
                        fl4 = 0;
                    }
                    if constexpr (STAGE) {
                        // After every 8 windows (and after the read's last one) the wave writes back what has become a
                        // COMPLETE 128-byte line of the output array -- slots [wdone, wnew) of each read, wnew the last line
                        // boundary at or below the windows staged so far; 8 lanes serve one read, a store covers 8 reads.
                        // Only a read's first and last line are written in parts (they are shared with its neighbours).
                        // With the runs cut at multiples of 8 windows of the READ instead, every line of a read whose W is
                        // not a multiple of 8 was written as two partial lines by two stores: 2.2-2.5 against 4.2 TB/s.
                        if (((u32)s & 7u) == 7u || o == omax) {
                            const int o_prev = ((u32)s & 7u) == 7u ? (int)o - 8 : (int)(o & ~7u) - 1;   // (the flush before this one)
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            const u32 piece = lane & 7u;
#pragma unroll
                            for (u32 rr = 0; rr < 8u; ++rr) {
                                const u32 rl = rr * 8u + (lane >> 3);           // read (lane) of the tile this lane serves now
                                u64 s0 = (t * 64u + rl) * (u64)W;               // its first slot
                                u32 nw_rl = W;                                  // ... and its windows
                                if constexpr (RAGGED) {
                                    s0 = ((u64)(u32)__shfl((int)(u32)(slot0 >> 32), (int)rl, WAVE) << 32) | (u32)__shfl((int)(u32)slot0, (int)rl, WAVE);
                                    nw_rl = (u32)__shfl((int)nwin, (int)rl, WAVE);
                                }
                                const u32 a = (u32)s0 & 7u;
                                auto upto = [&](int oo) -> u32 {               // windows of read rl written once the flush at window oo is done
                                    if (oo < 0) return 0u;
                                    if ((u32)oo + 1u >= nw_rl) return nw_rl;
                                    const int w = (int)(((a + (u32)oo + 1u) & ~7u)) - (int)a;
                                    return w > 0 ? (u32)w : 0u;
                                };
                                const u32 wdone = upto(o_prev), wnew = upto((int)o);
#pragma unroll
                                for (u32 base = 0; base < 16u; base += 8u) {
                                    const u32 w = wdone + base + piece;
                                    if (w < wnew) {
                                        const uint4 v = *reinterpret_cast<const uint4*>(S + rl * SPITCH + 4u * (((u32)s0 + w) & 15u));
                                        { typedef u32 v4u __attribute__((ext_vector_type(4))); const v4u vv = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(vv, reinterpret_cast<v4u*>(one + 2u * (s0 + w))); }
                                    }
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        }
                    }
                }
            // next block: forward window up by one dword, reverse window down by one
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = f[j + 1];
                f[4] = Fat(i + 5);
#pragma unroll
                for (int j = 4; j > 0; --j) h[j] = h[j - 1];
                const u32 gnew = Gat(Q - i - 2);
                h[0] = alignbit(gprev, gnew, e2);
                gprev = gnew;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the slice is rewritten by the next tile
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- the final partial tile
    const u32 rem = (u32)(n_reads & 63u);
    if (rem != 0u && wave0 == 0 && lane < rem) {
        const u64 read = n_full * 64u + lane;
        if constexpr (RAGGED) {
            const u64 o0 = offsets[read], o1 = ends[read];
            if (!read_too_long(o1 - o0, too_long)) windows2_one_read(bases + o0, (u32)(o1 - o0), k, win_offsets[read], out);
        } else {
            windows2_one_read(bases + lead + read * (u64)L, L, k, read * (u64)W, out);
        }
    }
}

hipError_t launch_sweep_windows(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u64* fw, u64* rc, u64* canon, uint8_t* flags,
                                const u64* win_offsets, unsigned long long* queue, int n_cu, hipStream_t stream, const u64* offsets,
                                const u64* ends, bool two_words);
template <int NW, bool RAGGED = false>
static hipError_t launch_windows2_tiled_nw(const uint8_t* bases, u64 n_reads, u32 L, u32 k, const Win2Out& out, int n_cu, hipStream_t st,
                                           const u64* offsets = nullptr, const u64* win_offsets = nullptr, unsigned long long* too_long = nullptr,
                                           const u64* ends = nullptr, unsigned long long* queue = nullptr) {
    const u32 lead = RAGGED ? 0u : (u32)(reinterpret_cast<uintptr_t>(bases) & 15u);
    const u32 chunks = 4u * L + ((RAGGED || lead != 0u) ? 1u : 0u);
    const u32 ldsw = (chunks + 1u + (u32)NW + 8u + 3u) & ~3u;
    const size_t lds_bytes = (size_t)(ldsw + 64u * 68u) * 4u * 4u;
    u64 grid = (u64)n_cu * 4u;
    const u64 need = ((n_reads >> 6) + 3u) / 4u;
    if (grid > need) grid = need;
    if (grid == 0) grid = 1;
    // One word array per launch, through the staged write-back (whole 128-byte lines).  A launch writing all three wrote each
    // window's 16-byte pieces from its own lane, 16 W bytes apart: 1.2 TB/s against 3-4 per array this way -- the tile is read
    // and packed once more per array, which costs a tenth of what the writes do.  The flags ride with the first launch.
    u64* const arr[3] = {out.fw, out.rc, out.canon};
    bool first = true;
    for (int a = 0; a < 3; ++a) {
        if (!arr[a]) continue;
        Win2Out o1{nullptr, nullptr, nullptr, first ? out.flags : nullptr};
        (a == 0 ? o1.fw : a == 1 ? o1.rc : o1.canon) = arr[a];
        hipLaunchKernelGGL((windows2_tiled_kernel<NW, true, RAGGED>), dim3((unsigned)grid), dim3(256), lds_bytes, st, bases - lead, n_reads, L, k, o1, lead, offsets, win_offsets, too_long, ends, queue);
        first = false;
    }
    if (first) {   // flags only
        const size_t lds0 = (size_t)ldsw * 4u * 4u;
        hipLaunchKernelGGL((windows2_tiled_kernel<NW, false, RAGGED>), dim3((unsigned)grid), dim3(256), lds0, st, bases - lead, n_reads, L, k, out, lead, offsets, win_offsets, too_long, ends, queue);
    }
    if (hipError_t e = hipGetLastError()) return e;
    // the reads the passes marked (every pass marks the same ones): the spoiled windows' slots, zeroed
    if (queue) return launch_sweep_windows(bases, n_reads, RAGGED ? (L > 160u ? 256u : 160u) : L, k, out.fw, out.rc, out.canon, out.flags, win_offsets, queue, n_cu, st, offsets, ends, true);
    return hipSuccess;
}

// uniform reads of k..256 bases, k in 33..64; *handled = false: the caller takes the lane-per-read kernel
hipError_t launch_windows2_tiled(const kmx_reads* r, u32 k, u64* fw, u64* rc, u64* canon, uint8_t* flags, int n_cu, hipStream_t st,
                                 bool* handled, unsigned long long* queue) {
    *handled = false;
    const u32 L = r->read_len;
    if (r->d_offsets || k < 33 || k > 64 || L < k || L > 256 || r->n_reads < 64u) return hipSuccess;
    const bool mis = (reinterpret_cast<uintptr_t>(r->d_bases) & 15u) != 0u;
    if (mis && (L == 160 || L == 256)) return hipSuccess;   // (the extra chunk of an unaligned start must fit the frame)
    // 16-byte stores of the word arrays
    if ((reinterpret_cast<uintptr_t>(fw) | reinterpret_cast<uintptr_t>(rc) | reinterpret_cast<uintptr_t>(canon)) & 15u) return hipSuccess;
    if (r->n_reads * (u64)L >= (1ull << 62)) return hipSuccess;
    *handled = true;
    const Win2Out out{fw, rc, canon, flags};
    if (L <= 160) return launch_windows2_tiled_nw<10>(r->d_bases, r->n_reads, L, k, out, n_cu, st, nullptr, nullptr, nullptr, nullptr, queue);
    return launch_windows2_tiled_nw<16>(r->d_bases, r->n_reads, L, k, out, n_cu, st, nullptr, nullptr, nullptr, nullptr, queue);
}

// ragged reads (offsets + win_offsets), k in 33..64, 16-byte aligned bases; read_len = optional length bound (0: unknown -> the
// 256-base frame); tiles with a longer read take the per-read path inside the kernel
hipError_t launch_windows2_tiled_ragged(const kmx_reads* r, const u64* win_offsets, u32 k, u64* fw, u64* rc, u64* canon, uint8_t* flags, int n_cu,
                                        hipStream_t st, bool* handled, unsigned long long* too_long, const u64* ends, unsigned long long* queue) {
    *handled = false;
    if (!r->d_offsets || !win_offsets || k < 33 || k > 64 || r->n_reads < 64u || r->read_len > 256) return hipSuccess;
    if (reinterpret_cast<uintptr_t>(r->d_bases) & 15u) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(fw) | reinterpret_cast<uintptr_t>(rc) | reinterpret_cast<uintptr_t>(canon)) & 15u) return hipSuccess;
    u32 L = r->read_len ? r->read_len : 256u;
    if (L < k + 15u) L = k + 15u;
    *handled = true;
    const Win2Out out{fw, rc, canon, flags};
    if (L <= 160) return launch_windows2_tiled_nw<10, true>(r->d_bases, r->n_reads, L, k, out, n_cu, st, r->d_offsets, win_offsets, too_long, ends, queue);
    return launch_windows2_tiled_nw<16, true>(r->d_bases, r->n_reads, L, k, out, n_cu, st, r->d_offsets, win_offsets, too_long, ends, queue);
}

__global__ void __launch_bounds__(256)
windows2_generic_kernel(ReadsView rv, const u64* __restrict__ win_offsets, u32 k, u64* __restrict__ o_fw,
                        u64* __restrict__ o_rc, u64* __restrict__ o_canon, uint8_t* __restrict__ o_flags) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const Win2Out out{o_fw, o_rc, o_canon, o_flags};
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < rv.n_reads; r += stride) {
        const uint8_t* s;
        u32 len;
        rv.span(r, s, len);
        const u64 base = win_offsets ? win_offsets[r] : r * (u64)(rv.read_len >= k ? rv.read_len - k + 1u : 0u);
        windows2_one_read(s, len, k, base, out);
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
                                 int n_cu, hipStream_t st, unsigned long long* too_long) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets, too_long};
    hipLaunchKernelGGL(reduce_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, k, hasher, hk,
                       want_sumfw, out);
    return hipGetLastError();
}

hipError_t launch_windows_generic(const kmx_reads* r, const u64* win_off, u32 k, u64* fw, u64* rc, u64* canon,
                                  uint8_t* flags, int n_cu, hipStream_t st, unsigned long long* too_long) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets, too_long};
    hipLaunchKernelGGL(windows_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, win_off, k, fw,
                       rc, canon, flags);
    return hipGetLastError();
}

hipError_t launch_histogram_generic(const kmx_reads* r, u32 k, u32 hasher, u32 hk, u32 log2_buckets, u64* counts,
                                    int n_cu, hipStream_t st, unsigned long long* too_long) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets, too_long};
    hipLaunchKernelGGL(histogram_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, k, hasher, hk,
                       log2_buckets, counts);
    return hipGetLastError();
}

hipError_t launch_reduce2_generic(const kmx_reads* r, u32 k, u32 with_hash, kmx_summary2* out, int n_cu, hipStream_t st, unsigned long long* too_long,
                                  const u32* gate) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets, too_long};
    hipLaunchKernelGGL(reduce2_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, k, with_hash,
                       out, gate);
    return hipGetLastError();
}

hipError_t launch_windows2_generic(const kmx_reads* r, const u64* win_off, u32 k, u64* fw, u64* rc, u64* canon,
                                   uint8_t* flags, int n_cu, hipStream_t st, unsigned long long* too_long) {
    ReadsView rv{r->d_bases, r->n_reads, r->read_len, r->d_offsets, too_long};
    hipLaunchKernelGGL(windows2_generic_kernel, dim3(grid_for(r->n_reads, n_cu)), dim3(256), 0, st, rv, win_off, k, fw,
                       rc, canon, flags);
    return hipGetLastError();
}

}  // namespace kmx
