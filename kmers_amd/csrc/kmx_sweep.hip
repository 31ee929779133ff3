// kmx_sweep.hip -- the windows that hold an invalid byte, taken back out of the bit-sliced scan's sums (round 6).
//
// scan_bitsliced_kernel (kmx_bitslice_kernel.h, "reads with an invalid byte") scans a tile that holds a non-ACGTacgt byte as it is --
// such a byte counts as the base its bits (b >> 1) & 3 spell -- and leaves the 64-bit mask of the reads that touch a bad chunk
// behind queue[515].  What the reference's iterator does NOT yield is exactly the windows that hold an invalid byte (the
// last_invalid rule, canonical_kmer_iterator.rs:50-66): this kernel evaluates those windows, from the same codes, and subtracts them.
// Until round 5 the scan blanked such reads and they were ROLLED, one lane walking one read base by base: 0.5 TB/s of dirty
// reads, 2 % of the reads cost 42 % of the time, 10 % cost 2.2x (profiles/r05_dirty_bench.txt).
//
// A wave gathers 64 marked reads; every lane loads ITS read with 16-byte loads straight into registers, packs it (encode16) and
// keeps one bit per base, "invalid", which five shift-ORs of the multi-word mask smear over the k positions before it: one bit
// per WINDOW.  One N spoils k windows of a read's ~120, so a lane does not walk its read: it takes the first spoiled window a,
// pulls the packed words (and the complemented, group-reversed ones the rc side reads) back from its LDS row re-aligned to a,
// and evaluates the 32 windows from there with funnel shifts at compile-time positions, masked by the window bits -- again
// while any lane has windows left (a read of nothing but N: four rounds).  Every k from 13 to 64: a window is V1 + 1 dwords
// (V1 = (k - 1) / 16), compared from the top dword down.
#include "kmx_bitslice_kernel.h"

namespace kmx {

// NW: packed dwords per read (10: reads of up to 160 bases, 16: up to 256).  V1 = (k - 1) / 16: a k-mer is V1 + 1 dwords.
// RAGGED / SEG and the arguments: as scan_bitsliced_kernel's -- `n_reads` counts what that kernel calls a read (a segment, for
// SEG and for the long ragged reads), `L` is its frame.
template <int NW, int V1> constexpr int sweep_pitch() {     // dwords of a lane's LDS row: F, G, the window bits (odd: lane-strided access without bank conflicts)
    return ((NW + V1 + 4) + (NW + 4) + (NW / 2 + 1)) | 1;
}
// HIST: the windows are taken out of a bucket histogram instead of a summary (the word-domain scan's histogram sinks mark their dirty
// reads the same way: kmx_scan_kernel.h, SinkMarksDirty) -- `out` = the counters, want_hash = the hasher, want_sumfw = hasher_k | log2_buckets << 8.
// MODE 2 (ZERO, round 6): behind a MATERIALISE pass (kmx_canonical_windows: the word-domain scan's window sinks mark their dirty reads too and
// take the fast path on a dirty tile instead of rolling it per lane -- 2 % dirty reads cost the materialise 37 %): the slots of the windows
// that hold an invalid byte are written as the reference's iterator leaves them -- words 0, flags 0 (kmx.h) -- in every array asked for.
struct SweepZero {
    u64 *fw, *rc, *canon;
    uint8_t* flags;
    const u64* win_offsets;      // ragged: slot of window 0 of read r; nullptr: r (L - k + 1)
    u32 two;                     // [u64;2] k-mers (kmx_canonical_windows2): two words per slot
};
template <int NW, int V1, bool RAGGED, bool SEG, int MODE>
__device__ __forceinline__ void sweep_body(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 k, u32 want_hash,
                                           u32 want_sumfw, void* __restrict__ out, unsigned long long* __restrict__ queue,
                                           const u64* __restrict__ offsets, u32 lead, const u64* __restrict__ ends,
                                           const BsSeg& seg, const SweepZero& zo) {
    constexpr bool HIST = MODE == 1, ZERO = MODE == 2;
    u64* const masks = reinterpret_cast<u64*>(queue[515]);
    // how many reads the scan marked: the bit-sliced scan's last block leaves its count in [516] (and [512] at zero for the next launch,
    // kmx_device.h); the word-domain scan's histogram sinks count in [512], which their caller clears with the heads
    const u64 n_marked = queue[(HIST || ZERO) ? KMX_Q_MARKED : KMX_Q_MARKED_OUT];
    if (masks == nullptr || n_marked == 0) return;
    // A sweep costs the same for 5 reads as for 64: as many waves as fill their sweeps (~48 reads each), not as many as were launched
    // -- 0.1 % of 2e7 reads dirty: 33 us with every wave of the grid sweeping 5 reads, 16 us with a quarter of them
    // (profiles/r06_sweep_variants.txt).  The waves that stay stride over the mask groups by their own number.
    const u64 n_waves_all = (u64)gridDim.x * 4u;
    u64 n_waves = (n_marked + 47u) / 48u;
    n_waves = n_waves < 64u ? 64u : n_waves;
    n_waves = n_waves > n_waves_all ? n_waves_all : n_waves;
    // (waves 4b .. 4b + 3 are block b: the active ones are the first blocks, which the dispatcher spreads over the CUs)
    const u64 wave_id = (u64)blockIdx.x * 4u + (threadIdx.x >> 6);
    const bool idle = wave_id >= n_waves;
    if constexpr (!RAGGED && !SEG) {   // (the length the gate found, as in scan_bitsliced_kernel: the reads lie L0 bytes apart)
        const u32 gate = __builtin_amdgcn_readfirstlane(reinterpret_cast<const u32*>(queue)[2 * 513]);
        const u32 gate_len = __builtin_amdgcn_readfirstlane(reinterpret_cast<const u32*>(queue)[2 * 513 + 1]);
        if (gate == 1u && gate_len != 0u) L = gate_len;
    }
    constexpr int DWN = V1 + 1;          // dwords of a k-mer
    constexpr int NB = NW / 2;           // 32-bit words of a lane's per-base / per-window bits
    constexpr int NWW = V1 + 3;          // dwords that hold 32 windows: 31 + k bases
    static_assert(NW % 2 == 0 && V1 >= 0 && V1 <= 3, "frames of whole bit words; k <= 64");
    // a lane's LDS row: FR[0 .. NW) packed words, zeros up to NW + NWW | GR: two zero words, G[0 .. NW], a zero | WR: window bits, a zero
    constexpr int FR = 0, FRN = NW + NWW + 1, GR = FRN, GRN = NW + 4, WR = GR + GRN, PITCH = sweep_pitch<NW, V1>();
    static_assert(WR + NB + 1 <= PITCH, "row layout");
    __shared__ u64 aside_all[4][64];
    __shared__ u64 part[4][6];
    __shared__ u32 rows_all[4][64 * PITCH];
    const u32 lane = threadIdx.x & 63u;
    u64* const aside = aside_all[threadIdx.x >> 6];
    u32* const row = rows_all[threadIdx.x >> 6] + lane * PITCH;
#pragma unroll
    for (int j = 0; j < PITCH; ++j) row[j] = 0u;      // (the pads stay zero for good)
    const u64 n_full = n_reads >> 6;
    // the last byte any read of the batch owns: a lane's 16-byte loads run up to 15 bytes past ITS read, never past this
    const uint8_t* buf_end;
    if constexpr (RAGGED) buf_end = bases + ends[n_reads - 1u];
    else if constexpr (SEG) buf_end = bases + (n_reads / seg.J) * (u64)seg.L;
    else buf_end = bases + lead + n_reads * (u64)L;
    const u32 top_bits = 2u * k - 32u * (u32)V1;                       // bits of the k-mer's top dword (2 .. 32)
    const u32 mtop = top_bits >= 32u ? ~0u : (1u << top_bits) - 1u;
    const u32 cg = (k - 1u) & 15u;                                      // the rc stream is delayed by cg groups: the rc of window o then starts at group 16 (NW - V1) - 1 - o
    u64 a_n = 0, a_s0 = 0, a_s1 = 0, a_x0 = 0, a_x1 = 0, a_fw = 0;
    u32 n_aside = 0;
    auto sweep = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- this lane's read
        const uint8_t* sp = bases;
        u32 len = 0;
        [[maybe_unused]] u64 slot0 = 0;      // ZERO: the lane's read's first output slot
        if (lane < n_aside) {
            const u64 read = aside[lane];
            if constexpr (ZERO) slot0 = zo.win_offsets ? zo.win_offsets[read] : read * (u64)(L - k + 1u);
            sp = bases + lead + read * (u64)L;
            len = L;
            if constexpr (SEG) {     // segment `read` of a long uniform read (scan_bitsliced_kernel<.., SEG>)
                const u64 i = read / seg.J;
                const u32 j = (u32)(read - i * seg.J), w = L - k + 1u;
                sp = bases + i * (u64)seg.L + (j * w - (j > seg.J1 ? j - seg.J1 : 0u));
                len = L - (j >= seg.J1 ? 1u : 0u);
            }
            if constexpr (RAGGED) {
                const u64 o0 = offsets[read], o1 = ends[read];
                sp = bases + o0;
                len = (u32)(o1 - o0);
                if (len > 16u * NW) __builtin_trap();    // (the scan marks reads of tiles INSIDE the frame only)
            }
        }
        // ---- its bytes: all loads in flight before the first is looked at.  From the DWORD-aligned address below the read -- a
        // 16-byte load from an address that is not a multiple of 4 is taken apart by the memory pipeline, and a sweep then cost what
        // its 640 loads did, not what it computes (profiles/r06_sweep_parts.txt) -- and shifted into place afterwards (v_alignbyte_b32).
        // Unconditional: a chunk past the read's end, or one that would run past the batch's last byte, reads the batch's first 16
        // bytes instead -- what it returns is masked below.
        const u32 rsh = (u32)(reinterpret_cast<uintptr_t>(sp) & 3u);
        const uint8_t* const a4 = sp - rsh;
        uint4 v[NW];
        u32 vx = 0u;                                     // the dword behind the last chunk
#pragma unroll
        for (int g = 0; g < NW; ++g) {
            const uint8_t* p = a4 + 16u * g;
            const bool direct = 16u * g < len + rsh && p + 16 <= buf_end;
            __builtin_memcpy(&v[g], __builtin_assume_aligned(direct ? p : bases, 4), 16);
        }
        {
            const uint8_t* p = a4 + 16u * NW;
            const bool direct = 16u * NW < len + rsh && p + 4 <= buf_end;
            __builtin_memcpy(&vx, __builtin_assume_aligned(direct ? p : bases, 4), 4);
        }
        if (__any(len != 0u && a4 + 16u * ((len + rsh + 15u) >> 4) > buf_end)) {     // the batch's last bytes, one by one (at most one lane of one wave)
#pragma unroll
            for (int g = 0; g <= NW; ++g) {
                const uint8_t* p = a4 + 16u * g;
                if (16u * g < len + rsh && p + (g < NW ? 16 : 4) > buf_end) {
                    u32 t[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
                    for (u32 b = 0; b < 16u && p + b < buf_end; ++b) {
                        const u32 x = (u32)p[b] << (8u * (b & 3u));
                        t[0] |= (b >> 2) == 0u ? x : 0u; t[1] |= (b >> 2) == 1u ? x : 0u; t[2] |= (b >> 2) == 2u ? x : 0u; t[3] |= (b >> 2) == 3u ? x : 0u;
                    }
                    if (g < NW) v[g < NW ? g : 0] = make_uint4(t[0], t[1], t[2], t[3]);
                    else vx = t[0];
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NW; ++g) {
            const u32 nx = g + 1 < NW ? v[g + 1 < NW ? g + 1 : 0].x : vx;
            v[g].x = __builtin_amdgcn_alignbyte(v[g].y, v[g].x, rsh);
            v[g].y = __builtin_amdgcn_alignbyte(v[g].z, v[g].y, rsh);
            v[g].z = __builtin_amdgcn_alignbyte(v[g].w, v[g].z, rsh);
            v[g].w = __builtin_amdgcn_alignbyte(nx, v[g].w, rsh);
        }
        // ---- packed words F (an invalid byte: the code its bits spell, as the scan took it), and one bit per base: invalid, inside the read
        u32 F[NW], wb[NB + 1];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            u32 i0, i1;
            F[2 * j] = encode16_inv(v[2 * j], i0);
            F[2 * j + 1] = encode16_inv(v[2 * j + 1], i1);
            const u32 lo = 32u * j;
            const u32 inside = len <= lo ? 0u : (len >= lo + 32u ? ~0u : ~(~0u << (len - lo)));
            wb[j] = (i0 | (i1 << 16)) & inside;
        }
        wb[NB] = 0u;
        // ---- one bit per WINDOW: window o holds an invalid base <=> OR of the bits o .. o + k - 1; only the windows of the read
        for (u32 cover = 1u; cover < k;) {
            u32 s = cover < k - cover ? cover : k - cover;           // 1, 2, 4, 8, 16, 16, ... and the rest: below 32, one funnel shift per word
            s = s < 16u ? s : 16u;
#pragma unroll
            for (int j = 0; j < NB; ++j) wb[j] |= alignbit(wb[j + 1], wb[j], s);
            cover += s;
        }
        u32 ng = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const u32 lo = 32u * j, lastw = len - k;     // (len < k: no window)
            const u32 keep = len < k ? 0u : (lastw >= lo + 31u ? ~0u : (lastw < lo ? 0u : (2u << (lastw - lo)) - 1u));
            wb[j] &= keep;
            ng += (u32)__builtin_popcount(wb[j]);
            row[WR + j] = wb[j];
        }
        a_n += ng;
        // ---- the row: the packed words, and the complemented, group-reversed ones delayed by cg groups
        if constexpr (!ZERO) {
            u32 R[NW + 1];
#pragma unroll
            for (int m = 0; m < NW; ++m) {
                row[FR + m] = F[m];
                R[m] = revgroups32(~F[NW - 1 - m]);
            }
            R[NW] = 0u;
            const u32 sh = (32u - 2u * cg) & 31u;
#pragma unroll
            for (int m = 0; m <= NW; ++m) row[GR + 2 + m] = cg ? alignbit(R[m], m ? R[m - 1] : 0u, sh) : R[m];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- rounds of 32 windows from the lane's first spoiled one
        u64 r_x0 = 0, r_x1 = 0;
        for (;;) {
            u32 a = 0;
            bool active = false;
#pragma unroll
            for (int j = NB - 1; j >= 0; --j) {
                if (wb[j] != 0u) {
                    a = 32u * j + (u32)__builtin_ctz(wb[j]);
                    active = true;
                }
            }
            if (__ballot(active) == 0ull) break;
            // the bits of the windows a .. a + 31 (everything below a is clear); then everything below a + 32 is done
            const u32 q = a >> 5;
            u32 wv = alignbit(row[WR + q + 1u], row[WR + q], a & 31u);
            wv = active ? wv : 0u;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const u32 lo = 32u * j, e = a + 32u;
                wb[j] &= lo >= e ? ~0u : (lo + 32u <= e ? 0u : ~0u << (e - lo));
            }
            if constexpr (ZERO) {
                // the spoiled windows' slots, one store per array and window (k of them per invalid byte: 2 % of the reads with an N are
                // 0.6 windows per read of the batch)
                u32 bits = wv;
                while (bits != 0u) {
                    const u64 slot = slot0 + a + (u32)__builtin_ctz(bits);
                    bits &= bits - 1u;
                    if (zo.two) {
                        if (zo.fw) zo.fw[2u * slot] = zo.fw[2u * slot + 1u] = 0ull;
                        if (zo.rc) zo.rc[2u * slot] = zo.rc[2u * slot + 1u] = 0ull;
                        if (zo.canon) zo.canon[2u * slot] = zo.canon[2u * slot + 1u] = 0ull;
                    } else {
                        if (zo.fw) zo.fw[slot] = 0ull;
                        if (zo.rc) zo.rc[slot] = 0ull;
                        if (zo.canon) zo.canon[slot] = 0ull;
                    }
                    if (zo.flags) zo.flags[slot] = 0;
                }
                continue;
            }
            // the packed stream from base a; the rc stream from the group where window a + 31 starts (window a + j: 31 - j groups on)
            u32 F2[NWW], G2[NWW];
            {
                const u32 qa = a >> 4, sf = 2u * (a & 15u);
                const u32 p = 16u * (NW - V1) - a, qg = p >> 4, sg = 2u * (p & 15u);   // (group p of the ROW: its two zero words ahead of G[0] are 32 groups)
                u32 t[NWW + 1], u[NWW + 1];
#pragma unroll
                for (int j = 0; j <= NWW; ++j) {
                    t[j] = row[FR + qa + j];
                    u[j] = row[GR + qg + j];
                }
#pragma unroll
                for (int j = 0; j < NWW; ++j) {
                    F2[j] = alignbit(t[j + 1], t[j], sf);
                    G2[j] = alignbit(u[j + 1], u[j], sg);
                }
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int i = j >> 4, s = j & 15, iu = (31 - j) >> 4, su = (31 - j) & 15;
                u32 vi = (u32)__builtin_amdgcn_sbfe((int)wv, j, 1);   // all ones: a spoiled window of this read
                asm volatile("" : "+v"(vi));   // (opaque: hipcc otherwise turns the masks into a branch around every window)
                u32 fw[4] = {0u, 0u, 0u, 0u}, rc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int w = 0; w < DWN; ++w) {
                    const u32 x = s ? alignbit(F2[i + w + 1], F2[i + w], 2u * s) : F2[i + w];
                    const u32 y = su ? alignbit(G2[iu + w + 1], G2[iu + w], 2u * su) : G2[iu + w];
                    if (w == DWN - 1) {
                        fw[w] = __builtin_amdgcn_bitop3_b32(x, mtop, vi, 0x80 /* a & b & c */);
                        rc[w] = __builtin_amdgcn_bitop3_b32(y, mtop, vi, 0x80);
                    } else {
                        fw[w] = x & vi;
                        rc[w] = y & vi;
                    }
                }
                // fw < rc (canonical_kmer.rs:113-119; [u64;2]: the build-defined order of kmx.h -- the high word first)
                const u64 fw_lo = ((u64)fw[1] << 32) | fw[0], rc_lo = ((u64)rc[1] << 32) | rc[0];
                const u64 fw_hi = ((u64)fw[3] << 32) | fw[2], rc_hi = ((u64)rc[3] << 32) | rc[2];
                bool lt;
                if constexpr (DWN == 1) lt = fw[0] < rc[0];
                else if constexpr (DWN == 2) lt = fw_lo < rc_lo;
                else lt = fw_hi < rc_hi || (fw_hi == rc_hi && fw_lo < rc_lo);
                // LexHasher(k)(canon) = MASK[k] ^ max(fw, rc) (kmx_scan.hip, SinkReduce); the MASKs: once per read, by the parity of its count
                u32 cn[4], mx[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    cn[w] = w < DWN ? (lt ? fw[w] : rc[w]) : 0u;
                    mx[w] = w < DWN ? (lt ? rc[w] : fw[w]) : 0u;
                }
                if constexpr (HIST) {
                    // SinkHist::emit (kmx_hist.hip): the bucket of hash(canonical word), one device atomic per spoiled window
                    if (vi != 0u) {
                        const u64 canon = ((u64)cn[1] << 32) | cn[0], mxw = ((u64)mx[1] << 32) | mx[0];
                        const u32 hk = want_sumfw & 0xFFu, lb = want_sumfw >> 8;
                        const u64 h = want_hash == KMX_HASH_LEX ? (hk == k ? mask2k(k) ^ mxw : lex_hash(canon, hk)) : canon;
                        atomicAdd(static_cast<unsigned long long*>(out) + bucket_of(h, lb), ~0ull);
                    }
                    continue;
                }
                a_s0 += ((u64)cn[1] << 32) | cn[0];
                r_x0 ^= ((u64)mx[1] << 32) | mx[0];
                if constexpr (DWN > 2) {
                    a_s1 += ((u64)cn[3] << 32) | cn[2];
                    r_x1 ^= ((u64)mx[3] << 32) | mx[2];
                } else {
                    a_fw += fw_lo;
                }
                // (pinned order: left alone, the adds of a round become a balanced tree with every window's words live)
                asm volatile("" : "+v"(a_s0), "+v"(r_x0));
                if constexpr (DWN > 2) asm volatile("" : "+v"(a_s1), "+v"(r_x1));
                else asm volatile("" : "+v"(a_fw));
            }
        }
        if (ng & 1u) {      // an odd number of MASK[k]s
            const u32 kb = 2u * k;
            r_x0 ^= kb >= 64u ? ~0ull : (1ull << kb) - 1ull;
            if (kb > 64u) r_x1 ^= kb >= 128u ? ~0ull : (1ull << (kb - 64u)) - 1ull;
        }
        a_x0 ^= r_x0;
        a_x1 ^= r_x1;
        n_aside = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // A lane takes the mask of one tile, the wave gathers the reads 64 at a time (one ballot + one v_mbcnt per round: no list in
    // memory, no atomics) and sweeps whenever the next round would not fit.  Every mask goes back to zero: the caller never clears
    // the array.  (ONE call site of the sweep.)
    const u64 n_groups = (n_full + 63u) >> 6;
    u64 g = idle ? n_groups : wave_id;
    // (the masks of FOUR groups are requested together: one after the other, a wave that strides over a dozen groups to find its
    // reads spent more time waiting for masks than sweeping)
    u64 m0 = 0, m1 = 0, m2 = 0, m3 = 0, t = 0;   // m0: the masks being taken apart, of the tiles t (this lane's) -- m1..m3: of the groups n_waves, 2 n_waves, 3 n_waves on
    u32 left = 0;
    // ZERO: the window sinks mark COARSELY -- every read of a tile that holds an invalid byte (kmx_scan_kernel.h, SinkMarksCoarse) -- and a sweep
    // of 64 reads costs ~12 us: with an N in 2 % of the reads 73 % of the tiles are dirty, one sweep each, 0.7 ms per 1e7 reads.  So the wave first
    // looks at such a tile as the scan would have (its chunks in rows of 64, coalesced, one ballot per row) and keeps the reads that touch a chunk
    // with an invalid byte: ~1.5 us per tile, and one sweep per 64 DIRTY reads.  A tile whose chunks cannot be taken whole (the batch's last bytes,
    // a span outside the frame, an unaligned base) stays marked as it is: the per-read path is safe anywhere.
    [[maybe_unused]] auto lane64 = [&](u64 v, u32 l) -> u64 {
        return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), (int)l) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, (int)l);
    };
    [[maybe_unused]] auto refine = [&](u64 mi, u64 ti) -> u64 {
        u64 todo = __ballot(mi != 0ull);
        while (todo != 0ull) {
            const u32 src = (u32)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const u64 T = lane64(ti, src);
            u64 o0, o1;
            if constexpr (RAGGED) {
                o0 = offsets[T * 64u + lane];
                o1 = ends[T * 64u + lane];
            } else {
                o0 = (u64)lead + (T * 64u + lane) * (u64)L;
                o1 = o0 + L;
            }
            const u64 first = lane64(o0, 0u), last = lane64(o1, 63u);
            const u64 base_al = first & ~15ull, n_ch = (last - base_al + 15u) >> 4;
            const bool whole = n_ch <= 64u * (u64)NW && bases + base_al + 16u * n_ch <= buf_end && (reinterpret_cast<uintptr_t>(bases) & 15u) == 0u &&
                               !__any(o1 - o0 > 16u * (u64)NW);
            if (!whole) continue;
            const u32 rd_off = (u32)(o0 - base_al), rd_len = (u32)(o1 - o0);
            const u32 c0 = rd_off >> 4, c1 = rd_len ? (rd_off + rd_len - 1u) >> 4 : c0, q0 = c0 >> 6, b0 = c0 & 63u;
            const uint4* __restrict__ tb = reinterpret_cast<const uint4*>(bases + base_al);
            uint4 wv[NW];
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const u32 c = (u32)it * 64u + lane;
                wv[it] = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
                if (c < (u32)n_ch) wv[it] = tb[c];
            }
            u64 lo = 0, hi = 0;
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                u32 rb = 0;
                (void)encode16(wv[it], rb);
                const u64 rowb = __ballot(chunk_has_invalid(rb));
                lo = q0 == (u32)it ? rowb : lo;
                hi = q0 + 1u == (u32)it ? rowb : hi;
            }
            const u64 bits = b0 ? ((lo >> b0) | (hi << (64u - b0))) : lo;
            const u64 dm = __ballot(rd_len != 0u && (bits & ((1ull << (c1 - c0 + 1u)) - 1ull)) != 0ull);
            if (lane == src) mi = dm;
        }
        return mi;
    };
    auto fetch = [&](u64 gi) -> u64 {
        const u64 ti = gi * 64u + lane;
        u64 mi = (gi < n_groups && ti < n_full) ? masks[ti] : 0ull;
        if (mi != 0ull) masks[ti] = 0ull;
        if constexpr (ZERO) mi = refine(mi, ti);
        return mi;
    };
    for (bool more = true; more;) {
        for (;;) {
            const u64 b = __ballot(m0 != 0ull);
            if (b == 0ull) {
                if (left > 1u) {                 // the next group of the four
                    m0 = m1; m1 = m2; m2 = m3; m3 = 0ull;
                    t += 64u * n_waves;
                    left -= 1u;
                    continue;
                }
                if (g >= n_groups) {
                    more = false;
                    break;
                }
                m0 = fetch(g); m1 = fetch(g + n_waves); m2 = fetch(g + 2u * n_waves); m3 = fetch(g + 3u * n_waves);
                t = g * 64u + lane;
                left = 4u;
                g += 4u * n_waves;
                continue;
            }
            const u32 nd = (u32)__builtin_popcountll(b);
            if (n_aside + nd > 64u) break;   // sweep first
            const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(b >> 32), __builtin_amdgcn_mbcnt_lo((u32)b, 0u));
            if (m0 != 0ull) {
                aside[n_aside + rank] = t * 64u + (u32)__builtin_ctzll(m0);
                m0 &= m0 - 1ull;
            }
            n_aside += nd;
        }
        if (n_aside != 0u) sweep();
    }
    // what the scan counted and the reference does not yield: taken back out (wrapping sums, a self-inverse fold).  One set of
    // atomics per BLOCK (the waves all finish within microseconds of each other: profiles/r03_dirty_breakdown.txt)
    {
        const u64 wn = wave_sum(a_n), ws0 = wave_sum(a_s0), ws1 = wave_sum(a_s1), wx0 = wave_xor(a_x0), wx1 = wave_xor(a_x1), wf = wave_sum(a_fw);
        if (lane == 0) {
            u64* pw = part[threadIdx.x >> 6];
            pw[0] = wn; pw[1] = ws0; pw[2] = ws1; pw[3] = wx0; pw[4] = wx1; pw[5] = wf;
        }
    }
    __syncthreads();
    if (HIST || ZERO || threadIdx.x != 0) return;
    const u64 n = part[0][0] + part[1][0] + part[2][0] + part[3][0];
    if (n == 0) return;   // nothing to take out: no atomics
    const u64 s0 = part[0][1] + part[1][1] + part[2][1] + part[3][1], s1 = part[0][2] + part[1][2] + part[2][2] + part[3][2];
    const u64 x0 = part[0][3] ^ part[1][3] ^ part[2][3] ^ part[3][3], x1 = part[0][4] ^ part[1][4] ^ part[2][4] ^ part[3][4];
    const u64 f = part[0][5] + part[1][5] + part[2][5] + part[3][5];
    if constexpr (V1 <= 1) {
        kmx_summary* o = static_cast<kmx_summary*>(out);
        atomicAdd((unsigned long long*)&o->n_valid, (unsigned long long)(0ull - n));
        atomicAdd((unsigned long long*)&o->sum_canon, (unsigned long long)(0ull - s0));
        if (want_hash) atomicXor((unsigned long long*)&o->xor_hash, (unsigned long long)x0);
        if (want_sumfw != 0u) atomicAdd((unsigned long long*)&o->sum_fw, (unsigned long long)(0ull - f));
    } else {
        kmx_summary2* o = static_cast<kmx_summary2*>(out);
        atomicAdd((unsigned long long*)&o->n_valid, (unsigned long long)(0ull - n));
        atomicAdd((unsigned long long*)&o->sum_lo, (unsigned long long)(0ull - s0));
        atomicAdd((unsigned long long*)&o->sum_hi, (unsigned long long)(0ull - s1));
        if (want_hash) {
            atomicXor((unsigned long long*)&o->xor_lo, (unsigned long long)x0);
            atomicXor((unsigned long long*)&o->xor_hi, (unsigned long long)x1);
        }
    }
}

template <int NW, int V1, bool RAGGED, bool SEG, bool HIST = false>
__global__ void __launch_bounds__(256) sweep_flagged_kernel(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 k, u32 want_hash,
                                                            u32 want_sumfw, void* __restrict__ out, unsigned long long* __restrict__ queue,
                                                            const u64* __restrict__ offsets, u32 lead, const u64* __restrict__ ends,
                                                            const BsSeg seg) {
    sweep_body<NW, V1, RAGGED, SEG, HIST ? 1 : 0>(bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg, SweepZero{});
}
// (V1 = 0: a window's words are not formed)
template <int NW, bool RAGGED>
__global__ void __launch_bounds__(256) sweep_zero_kernel(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 k,
                                                         unsigned long long* __restrict__ queue, const u64* __restrict__ offsets, u32 lead,
                                                         const u64* __restrict__ ends, const SweepZero zo) {
    sweep_body<NW, 0, RAGGED, false, 2>(bases, n_reads, L, k, 0u, 0u, nullptr, queue, offsets, lead, ends, BsSeg{0, 0, 0, 0, 0}, zo);
}

template <int NW, bool RAGGED, bool SEG>
static hipError_t launch_sweep_v(u32 v1, dim3 grid, hipStream_t stream, const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 want_hash, u32 want_sumfw,
                                 void* out, unsigned long long* queue, const u64* offsets, u32 lead, const u64* ends, const BsSeg& seg) {
    switch (v1) {
    case 0: hipLaunchKernelGGL((sweep_flagged_kernel<NW, 0, RAGGED, SEG>), grid, dim3(256), 0, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg); break;
    case 1: hipLaunchKernelGGL((sweep_flagged_kernel<NW, 1, RAGGED, SEG>), grid, dim3(256), 0, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg); break;
    case 2: hipLaunchKernelGGL((sweep_flagged_kernel<NW, 2, RAGGED, SEG>), grid, dim3(256), 0, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg); break;
    case 3: hipLaunchKernelGGL((sweep_flagged_kernel<NW, 3, RAGGED, SEG>), grid, dim3(256), 0, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// The launch behind every scan_bitsliced_kernel on ASCII input (launch_bs): arguments as that kernel's.
hipError_t launch_sweep_flagged(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 want_hash, u32 want_sumfw, void* out,
                                unsigned long long* queue, int n_cu, hipStream_t stream, const u64* offsets, u32 lead, const u64* ends,
                                const BsSeg& seg, bool ragged, bool is_seg) {
    if (k < 2u || k > 64u || k == 32u || L < k || L > 256u) return hipErrorInvalidValue;
    u64 grid1 = (u64)n_cu * 2u;     // (eight waves per CU: 10 % of 2e7 reads dirty 184 us, sixteen 223, four 212 -- profiles/r06_sweep_variants.txt)
    const u64 need1 = ((n_reads >> 6) + 255u) / 256u;   // a wave takes 64 masks at a time
    if (grid1 > need1) grid1 = need1;
    const dim3 grid((unsigned)(grid1 ? grid1 : 1));
    const u32 v1 = (k - 1u) / 16u;
    const bool big = L > 160u;
    if (ragged) {
        if (big) return launch_sweep_v<16, true, false>(v1, grid, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg);
        return launch_sweep_v<10, true, false>(v1, grid, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg);
    }
    if (is_seg) {
        if (big) return launch_sweep_v<16, false, true>(v1, grid, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg);
        return launch_sweep_v<10, false, true>(v1, grid, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg);
    }
    if (big) return launch_sweep_v<16, false, false>(v1, grid, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg);
    return launch_sweep_v<10, false, false>(v1, grid, stream, bases, n_reads, L, k, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg);
}

// The sweep behind a scan of uniform reads (L <= 256, single-word k) that was launched WITHOUT one (KMX_BS_NO_SWEEP:
// kmx_canonical_reduce_host enqueues it only when the scan's count of marked reads, which it has seen, is not zero).
hipError_t launch_sweep_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw, kmx_summary* out,
                                unsigned long long* queue, int n_cu, hipStream_t stream) {
    const u32 lead = (u32)(reinterpret_cast<uintptr_t>(bases) & 15u);   // (as launch_bs: streamed from the aligned address below)
    const BsSeg seg{0, 0, 0, 0, 0};
    return launch_sweep_flagged(bases - lead, n_reads, L, k, want_hash ? 1u : 0u, want_sumfw ? 1u : 0u, out, queue, n_cu, stream, nullptr, lead,
                                nullptr, seg, false, false);
}

// Behind the materialise passes (kmx_scan.hip, launch_windows_*): uniform reads (L <= 256; offsets == nullptr) or reads behind offsets
// (L = their bound, 0: none; `ends`: nullptr = offsets + 1), k <= 31 -- or, two_words, the [u64;2] k-mers of kmx_canonical_windows2
// (k = 33..64, two words per slot).  Any of the four arrays may be nullptr.
hipError_t launch_sweep_windows(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u64* fw, u64* rc, u64* canon, uint8_t* flags,
                                const u64* win_offsets, unsigned long long* queue, int n_cu, hipStream_t stream, const u64* offsets,
                                const u64* ends, bool two_words) {
    if (two_words ? (k < 33u || k > 64u) : (k < 2u || k > 31u)) return hipErrorInvalidValue;
    u32 lead = 0;
    if (!offsets) {
        lead = (u32)(reinterpret_cast<uintptr_t>(bases) & 15u);
        bases -= lead;
    } else if (!ends) {
        ends = offsets + 1;
    }
    u64 grid1 = (u64)n_cu * 2u;
    const u64 need1 = ((n_reads >> 6) + 255u) / 256u;
    if (grid1 > need1) grid1 = need1;
    const dim3 grid((unsigned)(grid1 ? grid1 : 1));
    const bool big = L > 160u || (offsets && L == 0u);
    const u32 Lf = offsets ? (big ? 256u : 160u) : L;
    if (Lf < k || Lf > 256u) return hipErrorInvalidValue;
    const SweepZero zo{fw, rc, canon, flags, win_offsets, two_words ? 1u : 0u};
    if (offsets) {
        if (big) hipLaunchKernelGGL((sweep_zero_kernel<16, true>), grid, dim3(256), 0, stream, bases, n_reads, Lf, k, queue, offsets, lead, ends, zo);
        else hipLaunchKernelGGL((sweep_zero_kernel<10, true>), grid, dim3(256), 0, stream, bases, n_reads, Lf, k, queue, offsets, lead, ends, zo);
    } else {
        if (big) hipLaunchKernelGGL((sweep_zero_kernel<16, false>), grid, dim3(256), 0, stream, bases, n_reads, Lf, k, queue, offsets, lead, ends, zo);
        else hipLaunchKernelGGL((sweep_zero_kernel<10, false>), grid, dim3(256), 0, stream, bases, n_reads, Lf, k, queue, offsets, lead, ends, zo);
    }
    return hipGetLastError();
}

// Behind a word-domain scan whose sink marks dirty reads (the bucket histograms): what the windows with an invalid byte added to the
// counters is subtracted.  Uniform reads (L) or reads behind offsets (L = their bound, 0: none -- the 16-word frame); k <= 31.
hipError_t launch_sweep_hist(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 hasher, u32 hk, u32 log2_buckets, u64* counts,
                             unsigned long long* queue, int n_cu, hipStream_t stream, const u64* offsets) {
    if (k < 2u || k > 31u) return hipErrorInvalidValue;
    u32 lead = 0;
    if (!offsets) {       // (as launch_one: streamed from the aligned address below an unaligned base)
        lead = (u32)(reinterpret_cast<uintptr_t>(bases) & 15u);
        bases -= lead;
    }
    u64 grid1 = (u64)n_cu * 2u;
    const u64 need1 = ((n_reads >> 6) + 255u) / 256u;
    if (grid1 > need1) grid1 = need1;
    const dim3 grid((unsigned)(grid1 ? grid1 : 1));
    const bool big = L > 160u || (offsets && L == 0u);
    const u32 Lf = offsets ? (big ? 256u : 160u) : L;
    if (Lf < k) return hipErrorInvalidValue;
    const BsSeg seg{0, 0, 0, 0, 0};
    const u32 packed = hk | (log2_buckets << 8);
    const u64* ends = offsets ? offsets + 1 : nullptr;
#define KMX_SWEEP_HIST(NWv, V1v, RG) hipLaunchKernelGGL((sweep_flagged_kernel<NWv, V1v, RG, false, true>), grid, dim3(256), 0, stream, bases, n_reads, Lf, k, hasher, packed, static_cast<void*>(counts), queue, offsets, lead, ends, seg)
    const bool v1 = k >= 17u;
    if (offsets) {
        if (big) { if (v1) KMX_SWEEP_HIST(16, 1, true); else KMX_SWEEP_HIST(16, 0, true); }
        else     { if (v1) KMX_SWEEP_HIST(10, 1, true); else KMX_SWEEP_HIST(10, 0, true); }
    } else {
        if (big) { if (v1) KMX_SWEEP_HIST(16, 1, false); else KMX_SWEEP_HIST(16, 0, false); }
        else     { if (v1) KMX_SWEEP_HIST(10, 1, false); else KMX_SWEEP_HIST(10, 0, false); }
    }
#undef KMX_SWEEP_HIST
    return hipGetLastError();
}

}  // namespace kmx
