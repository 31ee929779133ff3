// kmx_bitslice_kernel.h -- K1b: bit-sliced canonical k-mer scan (the headline kernel: every k from 9 to 64).
//
// Why: on gfx950 only the simplest VALU ops (v_and/or/xor/not, v_lshrrev, v_add_u32, v_bitop3_b32)
// issue at 32 lanes/clk; v_alignbit, v_perm, v_cmp*, v_cndmask, carry adds, 64-bit ops and v_bcnt
// run at half that (measured, tools/ubench2.hip, profiles/r01_valu_rates.txt).  A word-per-k-mer
// window costs >= 16 full-rate issue slots, which caps the scan near 45 % of the HBM roofline.
// Bit-slicing ACROSS 32 READS, with the counting on the matrix pipe, brings it to ~5 lane-operations per k-mer:
//
//   * a register holds ONE bit (base i, bit b) of 32 different reads ("plane");
//   * fw < rc is a ripple of v_bitop3_b32 over the top ceil(k/2) base pairs only, because rc is the
//     complemented mirror of fw: fw base k-1-j meets rc base = ~(fw base j)   (kmer.rs:124-136);
//   * the wrapping sum of canonical words needs no select and no 64-bit add: with m = (fw < rc),
//       sum(canon) = sum_{t,b} 2^(2t+b) * (C[t][b] + C[k-1-t][b]) - popcount(m)*MASK[k] + sum(all rc)
//     where C[t][b] = sum over windows o and reads r of m[o][r] * plane(o+t,b)[r], and sum(all rc), sum(all fw) and the
//     LexHasher xor-fold follow from per-plane popcount totals;
//   * C is a 0/1 correlation: the diagonal sums of G[o][beta] = sum_r m[o][r] * plane(beta)[r], a matrix product over the
//     tile's 64 reads that v_mfma_scale_f32_32x32x64_f8f6f4 computes exactly from FP4 operands made with one v_and per dword
//     (round 4; rounds 1-3 spent a v_and + v_bcnt per (window, plane) pair on it, half of the kernel's instructions).
//
// Data flow per 64-read tile (one wave, no block barrier):
//   A. coalesced 16 B/lane buffer loads -> encode16 (v_dot4_u32_u8 pack, v_perm_b32 validate) -> LDS
//   B. lane r pulls read r's packed words back (ds_read_b32) and realigns them (v_alignbit_b32)
//   C. 32x32 bit transposes across lanes, 5 x (exchange with lane^d, rotate, bit-select) per 16 bases:
//      lane p of each half-wave ends with plane p -> LDS plane array of the half's 32 reads
//   D. pass 1: lane = (set, WPL adjacent windows): the ripples over 2 ceil(k/2) plane pairs (ds_read_b64) -> mask words m;
//      pass 2: mask words and planes become matrix operands, <= 4 WPL matrix instructions accumulate G for every tile.
// The final partial tile takes roll_read (exact iterator semantics, canonical_kmer_iterator.rs:42-70).  A tile with a
// non-ACGTacgt byte is scanned as it is; the windows that hold such a byte -- the ones the iterator does not yield -- are taken
// back out by sweep_flagged_kernel (kmx_sweep.hip) -- see "reads with an invalid byte" in the kernel.
#pragma once
#include "kmx_device.h"

#include <type_traits>

namespace kmx {

// per-plane weights of sum over all windows of fw / rc (closed form; evaluated once per wave)
__device__ __forceinline__ void plane_weights(u32 i, u32 L, u32 k, u64& wf, u64& wr) {
    wf = 0;
    wr = 0;
    if (i >= L) return;
    const int o_lo = (int)i - (int)k + 1 > 0 ? (int)i - (int)k + 1 : 0;
    const int o_hi = (int)i < (int)(L - k) ? (int)i : (int)(L - k);
    if (o_lo > o_hi) return;
    // wf = sum_{o} 4^(i-o) ; wr = sum_{o} 4^(k-1-(i-o))
    const u32 e_lo = i - (u32)o_hi, e_hi = i - (u32)o_lo;  // exponents of 4, e_hi <= k-1
    wf = ((1ull << (2u * (e_hi + 1u))) - (1ull << (2u * e_lo))) / 3ull;
    const u32 f_lo = k - 1u - e_hi, f_hi = k - 1u - e_lo;
    wr = ((1ull << (2u * (f_hi + 1u))) - (1ull << (2u * f_lo))) / 3ull;
}

// a wave-uniform 64-bit value, pinned to scalar registers
__device__ __forceinline__ u64 uniform_u64(u64 v) {
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// the lane id, rematerialised where it is wanted (two VALU instructions that nothing can hoist): at a cold use site of a long kernel
// the alternative is a register held -- or spilled and reloaded behind an s_waitcnt vmcnt(0), with the next tile's rows in flight -- across the tile loop
__device__ __forceinline__ u32 lane_now() {
    u32 r;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(r));
    return r;
}

// the word of a wave-uniform 64-bit mask that belongs to this lane's half-wave (lanes 0..31: the low word).  hipcc turns
// `half ? (u32)(x >> 32) : (u32)x` into a 64-bit shift by a VGPR holding 32 * half -- one more value kept across the tile loop, and in the
// variants that run at their register limit the one that gets spilled: its reload sat in the middle of every tile behind an
// s_waitcnt vmcnt(0), i.e. behind the next tile's rows (round 5; the 13-word segment variant had three of them per tile)
__device__ __forceinline__ u32 half_word(u64 x) { return (u32)(x >> (lane_now() & 32u)); }

// one step of the fw<rc ripple: lt' = (~a & ~q) | ((a ^ q) & lt) as a single v_bitop3_b32
// (truth table with S0=lt=0xF0, S1=a=0xCC, S2=q=0xAA: 0x11 | (0x66 & 0xF0) = 0x71)
__device__ __forceinline__ u32 ripple(u32 lt, u32 a, u32 q) { return __builtin_amdgcn_bitop3_b32(lt, a, q, 0x71); }

// keep ? x : y per bit as ONE full-rate v_bitop3_b32 (hipcc otherwise emits v_and + half-rate v_and_or)
// (S0=x=0xF0, S1=y=0xCC, S2=keep=0xAA: (0xF0 & 0xAA) | (0xCC & 0x55) = 0xE4)
__device__ __forceinline__ u32 bitsel(u32 x, u32 y, u32 keep) { return __builtin_amdgcn_bitop3_b32(x, y, keep, 0xE4); }

// d += popcount(x) as ONE v_bcnt_u32_b32 (hipcc otherwise splits it into v_bcnt(x,0) + v_add3_u32)
__device__ __forceinline__ void pc_acc(u32& d, u32 x) { asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x)); }

// ---- FP4 (E2M1) operands of v_mfma_scale_f32_32x32x64_f8f6f4 out of bit planes, without spreading bits (tools/mfma/fp4_probe.hip).
// A nibble with ONE bit set is 0.5 (bit 0), 1.0 (bit 1) or 2.0 (bit 2); bit 3 is the sign.  A lane's operand is 4 dwords = 32
// nibbles = its row's (column's) 32 values along K, and K-slot (dword d, nibble n) may stand for any read as long as both
// operands agree.  So dword d of the B operand (a plane word P, bit rho = read rho) is just P masked to one bit per nibble --
// reads 4n+2 at weight 2, 4n+1 at 1, 4n at 0.5, and 4n+3 (one shift: bit 3 would be the sign) at 2 -- and the A operand (a
// mask word M) carries the same reads in the same slots at the reciprocal weights (rotations of M): every product is exactly
// 0 or 1, the fp32 sums are exact integers below 2^24.
typedef int bs_v8i __attribute__((ext_vector_type(8)));
typedef float bs_v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bs_v8i fp4_operand_b(u32 x) {
    bs_v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
    r[0] = (int)(x & 0x44444444u);
    r[1] = (int)(x & 0x22222222u);
    r[2] = (int)(x & 0x11111111u);
    r[3] = (int)((x >> 1) & 0x44444444u);
    return r;
}
__device__ __forceinline__ bs_v8i fp4_operand_a(u32 m) {
    bs_v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
    r[0] = (int)(alignbit(m, m, 2) & 0x11111111u);
    r[1] = (int)(m & 0x22222222u);
    r[2] = (int)(alignbit(m, m, 30) & 0x44444444u);
    r[3] = (int)(alignbit(m, m, 3) & 0x11111111u);
    return r;
}

// PACKED: `bases` is a SeqVector (kmx_seqvec.hip), read r = its bases [r*L, (r+1)*L): a tile is 16*L bytes of ready-made
// 2-bit codes that go from HBM straight into the packed LDS buffer -- no phase A, nothing to validate.
// RAGGED: reads of different lengths, read r = bases[offsets[r], offsets[r+1]); L is then the frame: the longest read a
// tile may hold (<= 16*NW).  A tile is still 64 consecutive reads = one contiguous byte span, streamed from its
// 16-byte-aligned start.  Lane r realigns its own read from its own offset and zeroes the bases past its end; two
// kinds of extra 32x32 transposes ride along with the NW data groups in phase C:
//   * NV validity words per read (bit o = "window o lies inside this read") -> plane V_o = the reads that own window
//     o, ANDed into the fw<rc mask of phase D (so a window past a read's end counts for nothing);
//   * the last K-1 bases of every read (NE dwords): the rc-side sum of the epilogue runs over all planes >= K-1-t of
//     a read instead of stopping t bases before ITS end, and these per-base totals take the excess back out.
// A tile whose span or longest read leaves the frame, or that would load past the end of the buffer, and tiles with
// an invalid byte, take the per-lane rolling path as before.
// dwords of one set's plane area (the kernel and launch_bs size the LDS from it)
constexpr int bs_plane_dwords(int NW) { return 32 * NW + 16; }
// rows of the next tile requested LATE (between pass 1 and pass 2 of phase D): see the kernel, "rows of the prefetch requested late"
template <int K, int NW, int WPL, bool PACKED, bool RAGGED, bool SEG> constexpr int bs_late() {
    return PACKED ? 0 : NW == 10 ? ((K > 32 || RAGGED || SEG) ? 5 : WPL > 4 ? (K <= 17 ? 0 : 1) : 0) : NW < 10 ? (K > 32 ? 0 : 3) : NW == 13 ? ((K <= 32 && !SEG) ? 3 : 7) : NW == 11 ? (SEG ? 6 : 3) : 8;
}
// PARK (round 6): phase A leaves every chunk's validation word in LDS beside its packed word, and a tile with an invalid byte
// looks its reads up there (30 instructions) instead of validating the tile a second time from w[] (150).  The second dword per
// chunk is LDS time, though, and the variants that are short of it pay on CLEAN input: interleaved with round 5's build on one box the
// headline <31,10,4> and k = 13 lost nothing, k = 21 (five windows per lane, a late row) 0.6 %, the two-word k = 63 2 %, the 7- / 13- /
// 16-word frames 2-3 % (profiles/r06_ab_r5_r6.txt).  So: the uniform 10-word frame without late rows -- k = 9..17 and 22..31 on
// 150-base reads, the metric's shape -- parks; every other variant finds its dirty reads the round-5 way.
template <int K, int NW, int WPL, bool PACKED, bool RAGGED, bool SEG> constexpr bool bs_park() {
    return !PACKED && !RAGGED && !SEG && NW == 10 && K <= 32 && bs_late<K, NW, WPL, PACKED, RAGGED, SEG>() == 0;
}
// dwords of a wave's packed region (kernel and launch_bs)
template <int NW, bool PACKED, bool PARK> __host__ __device__ constexpr unsigned bs_packed_dwords(unsigned chunks, unsigned wpl) {
    if (PARK) return 64u * (NW + 1);
    unsigned ldsw = (chunks + (PACKED ? 4u : 1u) + 6u + 3u) & ~3u;
    return ldsw < 64u * wpl ? 64u * wpl : ldsw;
}
constexpr int BS_LOAD_NT = 2;   // cache policy of the tile loads (aux bit 1 = nt: streamed once; +1.6 % over none, sc0 / sc1 nothing -- profiles/r03_load_policy_variants.txt)
// fp32 accumulator blocks (16 registers each) of pass 2: window block i (32 windows) meets the planes of bases 32i .. 32i+K+30,
// i.e. the blocks of 16 bases (32 planes) 2i .. 2i + bs_acc_blocks(K) - 1
constexpr int bs_acc_blocks(int K) { return (K + 30) / 16 + 1; }
// Waves per SIMD a variant is compiled for.  With 16 (bs_acc_blocks(K) - 1) accumulators and the prefetch rows of a tile in
// registers the kernels need 100..168 registers: three waves (measured round 4: a 32-accumulator form at four waves was 3 %
// slower than the 64-accumulator one at three -- profiles/r04_mfma_variants.txt; the short frames that fit 128 registers run
// at four).  The two-word k on the 13- and 16-word frames run at two.  Round 4 lost a build of the ragged 10-word frame at three waves
// (68 bytes of spills) that returned a wrong sum_canon on 1.2e6 segments of 1000-base reads while a build that differed by the order of
// two conjuncts passed; round 5 could not get it back (three reconstructions pass: DESIGN section 7) -- the full-input oracle compare
// and the at-size tests of every ragged frame (tests/test_gpu_fullsize.py, test_gpu_round5.py) are what guards these variants now.
// (Two-word k, round 4: the 13- and 16-word frames and every segment variant at two waves -- 32..270 bytes of spills at three.  Since
// round 5's register diet the 13-word frame and the 10-word segment variant fit three waves up to k = 49 with nothing in scratch
// (uniform 208-base reads at k = 47: profiles/r05_two_word_3waves.txt); the single-word segment variant in the 13-word frame, which
// kept 16 bytes at three waves in round 4, keeps none.)
// (Round 5: the single-word ragged variants on the 7- and the 10-word frame run at THREE waves.  Round 4 had measured that at +4..10 %
// with 8..68 bytes of spills; what those spills really cost was where their reloads sat: three per tile, each behind an
// s_waitcnt vmcnt(0) -- i.e. behind the next tile's rows, the software pipeline drained three times a tile.  With the lane id and
// the half-wave selects rematerialised at their cold use sites (lane_now, half_word) and the wave's LDS bases scalar, what is left
// in scratch (0..16 bytes) is parked across the tile loop and read in the epilogue or on a rare path only -- no scratch access on the
// loop's main path (tools/asm_loop_scratch.py) -- and three waves are +15 % on 2 %-trimmed 150-base reads (3.41 -> 2.97 ms per 1e8,
// 0.63 of the roofline; profiles/r05_ragged_3waves_final.txt).  The two-word ragged variants (200..350 bytes at three waves, most
// of it inside the loop) and the 16-word frame (235..250 registers) stay at two.)
template <int K, int NW, int WPL, bool PACKED, bool RAGGED, bool SEG = false> constexpr int bs_waves() {
    // two-word k: the 16-word frame, and from k = 50 (six accumulator blocks) the 13-word frame and the segment variants, at two
    return ((RAGGED && (NW > 10 || K > 32)) || (K > 32 && (NW > 13 || ((NW == 13 || SEG) && K > 49)))) ? 2 : 3;
}
// tiles between two folds of the fp32 accumulators into the 64-bit class sums: a power of two, far below the 2^24 / (8 window
// blocks x 64 reads) the sums stay exact integers for, and small enough that the full-size runs (~500 tiles per wave) exercise
// the fold (~1.5 instructions per tile)
constexpr unsigned BS_FOLD_TILES = 256;
// SEG (round 4): uniform reads too long for a frame (L > 256) as overlapping SEGMENTS on the UNIFORM kernel.  A read of
// seg.L bases = Wr windows is cut into seg.J segments; the first seg.J1 hold T windows, the others T - 1 (J1 T + (J - J1)(T - 1)
// = Wr: every window belongs to exactly one segment), segment j starts at base pos(j) = j T - max(0, j - J1) of its read.  A
// "read" of the kernel is a segment: L = T + K - 1 bases are loaded for each; a SHORT segment's last window (the next segment's
// first) is masked out of m, and the one thing that still counts it -- the closed form over the plane totals -- is corrected
// by the totals of the short segments' last K planes (TOTS).  No offsets, no validity planes, no extra transposes: the
// uniform kernel's instruction count, three waves per SIMD, every k from 13 to 64.
struct BsSeg {
    u32 L, J, J1, magic;   // magic = floor(2^32 / J) + 1: n / J = umulhi(n, magic) for the n < J + 64 < 128 it is used for (J < 64)
    u64 magic64;           // floor(2^64 / J) + 1: g / J = umul64hi(g, magic64), exact while g J < 2^64 (a 64-bit division per tile cost ~150 instructions)
};
template <int K, int NW, int WPL, bool PACKED = false, bool RAGGED = false, bool SEG = false>
__global__ void __launch_bounds__(256, (bs_waves<K, NW, WPL, PACKED, RAGGED, SEG>()))
scan_bitsliced_kernel(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 want_hash, u32 want_sumfw,
                      void* __restrict__ out /* kmx_summary (K<=32) or kmx_summary2 (K>32) */,
                      unsigned long long* __restrict__ queue, const u64* __restrict__ offsets, u32 lead,
                      const u64* __restrict__ ends, const BsSeg seg) {
    static_assert(!SEG || (!PACKED && !RAGGED && (NW == 10 || NW == 11 || NW == 13)), "segments of long uniform reads: ASCII, the 10-, the 11- and the 13-word frame");
    // `ends` (RAGGED with offsets): read r = bases[offsets[r], ends[r]) -- offsets + 1 for reads stored back to back, an array of
    // its own for the overlapping SEGMENTS a batch of long ragged reads was cut into (kmx_segments.hip, round 4).
    // `lead` (uniform ASCII input whose first byte is not 16-byte aligned): `bases` is the aligned address below it and
    // read r starts at byte lead + r*L.  A tile then spans one more chunk (its first holds the tail of the tile before
    // it), exactly like a ragged tile streamed from its aligned start; 0 for every other input.
    static_assert(!RAGGED || !PACKED, "ragged input: ASCII");
    static_assert(WPL <= 8, "the zero words behind the validity planes cover a lane's windows");
    constexpr int NE = RAGGED ? (K - 1 + 15) / 16 : 0;            // dwords holding the last K-1 bases of a read
    constexpr int NV = RAGGED ? (16 * NW - K + 1 + 31) / 32 : 0;  // validity words per read (one bit per window of the frame)
    constexpr u32 VS = RAGGED ? 32u * NV + 8u : 0u;             // validity planes of one set + 8 always-zero words (a lane's windows past the frame, idle lanes)
    constexpr int NXT = NW + NE;                                 // 32x32 transposes per half-wave and tile
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    // Gate (queue[513], zero unless kmx_canonical_reduce / kmx_canonical_reduce2 is deciding on the device whether reads behind
    // an offsets array are in fact uniform -- offsets_uniform_gate_kernel): 1 = only the uniform kernels run, 2 = only the ragged
    // ones (two-word k: the lane-per-read kernel).
    if constexpr (!PACKED) {
        const u32 gate = __builtin_amdgcn_readfirstlane(reinterpret_cast<const u32*>(queue)[2 * 513]);
        if (gate == (RAGGED ? 1u : 2u)) return;
        if constexpr (!RAGGED && !SEG) {
            // (round 5) the gate found the reads uniform at a length BELOW the bound this launch was laid out for (frame, windows per
            // lane, LDS): that length is what is scanned -- every quantity below derives from L
            const u32 gate_len = __builtin_amdgcn_readfirstlane(reinterpret_cast<const u32*>(queue)[2 * 513 + 1]);
            if (gate == 1u && gate_len != 0u) L = gate_len;
        }
    }
    // Plane storage of one 32-read set: base beta (2 planes = one u64) lives at u64 index
    // (beta & 3) * S2 + (beta >> 2).  In phase D lane g reads bases 4g+i: consecutive lanes then touch
    // consecutive u64s (conflict-free ds_read_b64) instead of a 32-byte stride (4-way bank conflicts,
    // measured SQ_LDS_BANK_CONFLICT = 65 % of LDS cycles with the linear layout).
    // (For an odd number of windows per lane the lane stride is an odd number of u64s and the plain linear
    // layout is already conflict-free.)
    // (Rotated layout, generally: base beta at u64 index (beta % WPL) * S2 + beta / WPL, for the WPL that divide 16.)
    constexpr bool ROT = WPL == 2 || WPL == 4 || WPL == 8;
    constexpr int RW = ROT ? WPL : 4;                // rows of the rotated layout
    constexpr int S2 = (16 * NW) / RW + 1;           // u64 row pitch (4*NW + 1 at WPL = 4)
    constexpr int PLANES = bs_plane_dwords(NW);      // dwords per set (>= 2 * RW * S2, >= 32*NW)
    static_assert(PLANES >= 2 * RW * S2 && PLANES >= 32 * NW, "plane area");
    const u32 lane = threadIdx.x & 63u;
    const u32 half = lane >> 5, p = lane & 31u;
    const u32 wib = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform, and only readfirstlane tells hipcc so: the LDS bases derived from it are scalars, not registers held -- or spilled -- across the tile loop)
    const u32 chunks = 4u * L + ((RAGGED || SEG || lead != 0u) ? 1u : 0u);   // 16-byte chunks a tile may span (+1 for an unaligned start)
    constexpr u32 PAD = PACKED ? 4u : 1u;                    // front pad of the packed region (4: keeps ds_write_b128 aligned)
    // packed region (as in kmx_scan.hip; it holds the mask words of pass 2 afterwards).  The variants that PARK (round 6): a multiple of 64
    // dwords fixed by the frame -- phase A leaves the VALIDATION word of chunk c (expected letters ^ bytes) exactly XOFF dwords behind its packed
    // word, in the plane area (free until phase C), with the same LDS instruction (ds_write2st64_b32: two dwords a compile-time multiple
    // of 256 bytes apart); see "reads with an invalid byte"
    constexpr bool PARK = bs_park<K, NW, WPL, PACKED, RAGGED, SEG>();
    u32 ldsw = bs_packed_dwords<NW, PACKED, PARK>(chunks, WPL);
    constexpr u32 XOFF = 64u * (NW + 1);
    static_assert(!PARK || (XOFF >= 64u * WPL && 1u + 64u * NW <= 2u * (u32)bs_plane_dwords(NW)), "the validation words fit the plane area");
    constexpr u32 CSA_DW = 4u * ((K + 1) / 2);         // 2 * NT 64-bit sums of the counter classes
    // SEG: per-plane totals of the SHORT segments (as TOT, [group][lane]) -- of the groups that hold the last window's K bases
    // only, the first of them (W - 1) >> 4: the whole frame's worth cost the 13-word frame its third block per CU
    constexpr u32 TOTS_G = (K - 1) / 16 + 2;
    constexpr u32 TOTS_DW = SEG ? 64u * TOTS_G : 0u;
    const u32 wave_dw = ldsw + 4u * PLANES + 2u * VS + (RAGGED ? 64u * (NE + 2) : 0u) + CSA_DW + TOTS_DW;   // a wave's area; behind the four: 64 dwords of the block's (the end of the kernel)
    u32* P = lds + wib * wave_dw;
    u32* PL = P + ldsw;                                      // [2][PLANES] plane array, 16-byte aligned

    const u64 n_full = n_reads >> 6;
    const u64 wave_id = (u64)blockIdx.x * 4u + wib;

    u32 posF = lane * L + lead + 16u * PAD;    // (ragged: set per tile from the lane's own offset)
    u32 qF = posF >> 4, aF = 2u * (posF & 15u);
    const u32 W = L - (u32)K + 1u;      // windows per read
    const u32 NG = (W + WPL - 1u) / WPL; // groups of WPL adjacent windows per read (<= 32: launch_bs checks it)
    const u32 rounds = (2u * NG + 63u) >> 6;   // == 1; a runtime value on purpose (the trip count of phase D's loop, see there)
    constexpr u32 SP = (u32)PLANES;      // set stride (dwords): each half-wave reads its own set, 32 lanes over the 64 banks

    // transpose stage constants: rotate amount and keep-mask per butterfly distance
    u32 tr_sh[5], tr_keep[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const u32 d = 16u >> s;
        u32 md = 0xFFFFFFFFu / ((1u << d) + 1u);             // bits whose index has bit d clear: 0x0000FFFF,0x00FF00FF,...
        tr_sh[s] = (p & d) ? d : 32u - d;
        tr_keep[s] = (p & d) ? ~md : md;
    }

    // counter class min(t, K-1-t): C[t][b] = sum over windows and reads of (m & plane(t,b)) and C[K-1-t][b] always enter the result
    // as a sum (the weights are symmetric under t <-> K-1-t)
    constexpr int NT = (K + 1) / 2;     // distinct t classes (the middle one of odd K pairs with itself)
    // v_perm_b32 selectors of the byte-granular stages ({S0=y: bytes 4-7, S1=x: bytes 0-3})
    const u32 tr_sel16 = (p & 16u) ? 0x03020706u : 0x05040100u;   // keep x.hi, take y.hi>>16  |  keep x.lo, take y.lo<<16
    const u32 tr_sel8 = (p & 8u) ? 0x03070105u : 0x06020400u;     // odd bytes kept, even from y.odd | even kept, odd from y.even
    u32 mcnt = 0;                       // sum popcount(m)
    // per-plane popcount totals of the planes this lane produces live in LDS (TOT[g][lane]); only this
    // lane ever touches its own slots, so plain read-modify-write is enough
    u32* TOT = PL + 2u * PLANES;
#pragma unroll
    for (int g = 0; g < NW; ++g) TOT[64u * g + lane] = 0;   // (indexed [group][lane]: one address register, a compile-time offset per group)
    u32* VAL = TOT + 2u * PLANES;       // ragged: [2][32*NV] validity planes of the current tile
    // ragged, per lane, kept in LDS (registers are what caps this variant's occupancy): QT[e][lane] = running popcount of the
    // lane's plane of the read-end words, NVR[lane] = windows of the lane's reads in bit-sliced tiles
    u32* QT = VAL + 2u * VS;
    u64* NVR = reinterpret_cast<u64*>(QT + 64u * NE);
    if constexpr (RAGGED) {
#pragma unroll
        for (int e = 0; e < NE; ++e) QT[e * 64 + lane] = 0;
        NVR[lane] = 0;
        if (lane < 16u) VAL[(lane >> 3) * VS + 32u * NV + (lane & 7u)] = 0;   // the zero words (never written again)
    }
    const bool sumfw_on = (want_sumfw & KMX_BS_SUMFW) != 0u;   // (the other bits: how the launch ends, kmx_device.h)
    const u64 total_bytes = RAGGED ? ends[n_reads - 1u] : 0;
    // ragged: per-tile geometry of the current and of the next tile (rel/len per lane, the rest wave-uniform)
    struct TileMeta { u32 rel = 0, len = 0, n_ch = 0; u64 base = 0; bool fits = true; };
    TileMeta cur_m, nx_m;
    // the two offsets of a lane are requested one iteration before anything looks at them (meta_issue / meta_finish)
    u64 raw_o0 = 0, raw_o1 = 0;
    auto meta_issue = [&](u64 t) {
        const u32 ln = lane_now();
        raw_o0 = offsets[t * 64u + ln];
        raw_o1 = ends[t * 64u + ln];
    };
    auto meta_finish = [&](TileMeta& m) {
        const u64 o0 = raw_o0, o1 = raw_o1;
        // (the builtins return int: through u32 first, or offsets >= 2^31 get sign-extended into the high word)
        const u32 t0l = __builtin_amdgcn_readfirstlane((u32)o0), t0h = __builtin_amdgcn_readfirstlane((u32)(o0 >> 32));
        const u32 t1l = __builtin_amdgcn_readlane((u32)o1, 63), t1h = __builtin_amdgcn_readlane((u32)(o1 >> 32), 63);
        const u64 t0 = ((u64)t0h << 32) | t0l, t1 = ((u64)t1h << 32) | t1l;
        m.base = t0 & ~15ull;
        const u64 nch = (t1 - m.base + 15u) >> 4;
        const u64 len64 = o1 - o0;
        m.len = len64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)len64;
        m.rel = (u32)(o0 - m.base);
        m.n_ch = (u32)(nch > 0xFFFFFFFFull ? 0xFFFFFFFFull : nch);
        m.fits = nch <= (u64)chunks && nch <= 64u * NW && !__any(m.len > L) && m.base + 16u * nch <= total_bytes;   // (one compare + a scalar test: a wave-wide max costs ten instructions)
    };
    u32 n_bs_tiles = 0;
    // pass 2 on the matrix pipe (see phase D): block pair q = (window block i, plane block 2i + q) of EVERY i and every tile goes
    // to the same accumulator block.  The first and the last pair share ONE: of pair 0 (bases 0..15 past the window block's
    // first) only the windows 0..15 can lie on a diagonal 0 <= t < K, of pair NAB-1 only the windows 16..31 (16 NAB - 31 >= K).
    // The rows that do not count are switched off through the A operand's per-lane E8M0 scale (0 = 2^-127: what they add
    // vanishes next to an integer and truncates to 0 where there is none -- tools/mfma/fp4_share_probe.hip).
    constexpr int NAB = bs_acc_blocks(K);
    constexpr int NAR = NAB - 1;       // accumulator blocks in registers
    static_assert(NAB >= 3 && 16 * NAB - 31 >= K, "the first and the last block pair share an accumulator block");
    bs_v16f acc[NAR];
#pragma unroll
    for (int q = 0; q < NAR; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[q][j] = 0.f;
    u64* const CSA = reinterpret_cast<u64*>(P + ldsw + 4u * PLANES + 2u * VS + (RAGGED ? 64u * (NE + 2) : 0u));
    for (u32 i = lane; i < 2u * NT; i += 64u) CSA[i] = 0ull;
    // ---- SEG: where the 64 segments of a tile lie.  Wave-uniform: the tile's first byte (aligned down to 16), its bytes from
    // there, the alignment lead, the index j0 of its first segment within that segment's read; per lane: the segment's first
    // byte relative to the tile's, and whether it is a short one.
    u32* const TOTS = reinterpret_cast<u32*>(CSA) + CSA_DW;
    struct SegTile { u64 base = 0; u32 nbytes = 0, lead = 0, j0 = 0; };
    SegTile cur_g, nx_g;
    u32 seg_rel = 0;            // this lane's segment of the CURRENT tile: first byte - the tile's first byte
    u64 seg_short = 0;          // the current tile's short segments (bit = lane = segment)
    u32 n_short = 0;            // short segments in this wave's scanned tiles: one window less each
    auto seg_pos = [&](u32 j) -> u32 { return j * W - (j > seg.J1 ? j - seg.J1 : 0u); };
    auto seg_div = [&](u32 n, u32& q, u32& r) {      // n < J + 64
        q = seg.J >= 64u ? (n >= seg.J ? 1u : 0u) : __umulhi(n, seg.magic);
        r = n - q * seg.J;
    };
    auto seg_geom = [&](u64 t, SegTile& g) {
        const u64 g0 = t * 64u, i0 = __umul64hi(g0, seg.magic64);
        const u32 j0 = (u32)(g0 - i0 * seg.J);
        const u64 first = i0 * (u64)seg.L + seg_pos(j0);
        u32 di, j1;
        seg_div(j0 + 63u, di, j1);
        const u64 last_end = (i0 + di) * (u64)seg.L + seg_pos(j1) + (W - (j1 >= seg.J1 ? 1u : 0u)) + (u32)(K - 1);
        g.base = first & ~15ull;
        g.lead = (u32)(first & 15u);
        // whole 16-byte chunks: the bytes behind the tile's last segment are the next tile's (real bases, not the zeros of an
        // out-of-range load, which phase A would take for invalid bytes and blank the tile's last segment for) -- except at
        // the very end of the batch, where the range stops with the last read
        const u64 all_end = (n_reads / seg.J) * (u64)seg.L;
        u64 span_end = (last_end + 15u) & ~15ull;
        span_end = span_end < all_end ? span_end : all_end;
        g.nbytes = (u32)(span_end - g.base);
        g.j0 = j0;
    };
    auto seg_lane = [&](const SegTile& g) {          // seg_rel / seg_short of the tile that becomes current
        u32 di, j;
        seg_div(g.j0 + lane_now(), di, j);
        seg_rel = di * seg.L + seg_pos(j) - seg_pos(g.j0);    // (mod 2^32: the true difference is below 64 L)
        seg_short = __ballot(j >= seg.J1);
    };
    if constexpr (SEG) {
#pragma unroll
        for (u32 g = 0; g < TOTS_G; ++g) TOTS[64u * g + lane] = 0;
    }
    // acc[q][j] of lane (half, p) is G[o][beta] for o = (j & 3) + 8 (j >> 2) + 4 half (mod 32), plane 32 q + p relative to the
    // window block: base o_blk + 16 q + p / 2, bit p & 1.  The diagonal t = beta - o in [0, K) is base t of the window; t and
    // K-1-t share a counter class.
    auto fold_acc = [&]() {
        u32 ln_ = lane;
        asm volatile("" : "+v"(ln_));   // (opaque: hipcc otherwise hoists the 16 NAB addresses and conditions of this rare path out of the tile loop -- and spills them)
        const int pb = (int)((ln_ & 31u) >> 1) - 4 * (int)(ln_ >> 5);
        u64* const cs_b = CSA + (ln_ & 1u);
#pragma unroll
        for (int q = 0; q < NAR; ++q) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // (registers 8..15 of the shared block are the rows 16..31: block pair NAB - 1)
                // (no branch around the add -- a diagonal outside [0, K) adds zero to one of eight classes)
                const int t = 16 * ((q == 0 && j >= 8) ? NAB - 1 : q) + pb - ((j & 3) + 8 * (j >> 2));
                const bool ok = t >= 0 && t <= K - 1;
                const int tt = ok ? t : (t & 7);
                const int tc = tt < K - 1 - tt ? tt : K - 1 - tt;
                atomicAdd(reinterpret_cast<unsigned long long*>(cs_b + 2 * tc), ok ? (unsigned long long)(u32)acc[q][j] : 0ull);
                acc[q][j] = 0.f;
            }
        }
    };
    // ---- reads with an invalid byte (ASCII input, uniform or ragged).  A tile that holds one used to go to the per-lane path
    // as a whole (64 reads rolled at 6.5x the cost of a bit-sliced tile); rounds 3-5 scanned such a tile with the offending reads
    // BLANKED and had them rolled, later swept, elsewhere -- but finding those reads re-validated the whole tile from w[], 150
    // instructions per dirty tile (73 % of the tiles when 2 % of the reads hold an N: +16 % on the scan), and the code's mere presence --
    // w[] alive past phase A, the prefetch behind it -- cost 2.7 % on CLEAN input (profiles/r06_dirty_variants.txt).  Round 6:
    //   * phase A parks every chunk's validation word in LDS at no instruction's cost (XOFF above); a dirty tile looks its reads'
    //     chunks up there -- LDS only, the next tile's rows already requested -- and leaves the 64-bit mask of the reads that touch a
    //     bad chunk in the array behind queue[515] (8 bytes per tile, all zero between calls; queue[512] = how many reads were marked);
    //   * nothing is blanked: the tile is scanned as it is, an invalid byte counting as the base its bits (b >> 1) & 3 spell, and
    //     sweep_flagged_kernel (kmx_sweep.hip) takes the windows that hold such a byte -- exactly those the reference's iterator does
    //     not yield (canonical_kmer_iterator.rs:50-66) -- back OUT of the sums, from the same codes.
    // queue[515] == 0: no array; a ragged tile then rolls as a whole here (uniform input: the host side always provides the array).
    constexpr bool INLINE = !PACKED;              // (packed input has no invalid codes)
    u64* const masks = INLINE ? reinterpret_cast<u64*>(uniform_u64(queue[515])) : nullptr;   // (read once: two scalar registers)
    u32 n_marked = 0;                             // reads this wave marked (wave-uniform); their total, in queue[512], tells the sweep how many waves to field
    // word-domain accumulators of the fallback path (tiles with invalid bytes, the final partial tile)
    struct FbAcc { u64 n = 0, s0 = 0, s1 = 0, x0 = 0, x1 = 0, fw = 0; };
    // (into one of sixteen partial summaries, each on a line of its own in the queue block -- the launch's last block adds them
    // up, at the end of the kernel: the blocks of a small batch all get here within microseconds, and 3-5 atomics each on the ONE
    // line of the summary were served one after the other)
    auto emit_sums = [&](u64 n, u64 r0, u64 r1, u64 h0, u64 h1, u64 f) {   // wave-uniform values, one set of atomics
        if (lane == 0) {
            unsigned long long* const S = queue + KMX_Q_SLOTS + (blockIdx.x & 15u) * 16u;
            atomicAdd(S + 0, (unsigned long long)n);
            atomicAdd(S + 1, (unsigned long long)r0);
            if constexpr (K > 32) atomicAdd(S + 2, (unsigned long long)r1);
            if (want_hash) {
                atomicXor(S + 3, (unsigned long long)h0);
                if constexpr (K > 32) atomicXor(S + 4, (unsigned long long)h1);
            }
            if constexpr (K <= 32) {
                if (sumfw_on) atomicAdd(S + 5, (unsigned long long)f);
            }
        }
    };
    constexpr bool FB_FLUSH = K <= 32;   // (two-word k-mers: the per-tile sums then live in scratch, 0.46 -> 0.37 of the roofline at k=63)
    FbAcc fb_all;                                             // !FB_FLUSH: summed over the whole run of the wave
    // (`pieces` > 1: the read's windows in `pieces` equal runs, this lane rolls run `piece` -- the final partial tile, below)
    // (always_inline: called from two places, the body is past hipcc's threshold in the widest variants -- as a CALL its closure and the
    // accumulators live in scratch: tests/test_kernel_resources.py)
    auto fallback_read_acc = [&](u64 read, FbAcc& fb, u32 piece = 0, u32 pieces = 1) __attribute__((always_inline)) {
        // first byte (packed input: first base) and length of what this lane rolls
        u64 at = (u64)lead + read * (u64)L;
        u32 len = L;
        if constexpr (RAGGED) {
            const u64 o0 = offsets[read], o1 = ends[read];
            if (read_too_long(o1 - o0, queue + KMX_TOOLONG_FROM_QUEUE)) return;   // (not scanned; kmx_ctx_synchronize reports it)
            at = o0;
            len = (u32)(o1 - o0);
        } else if constexpr (SEG) {
            const u64 i = read / seg.J;
            const u32 j = (u32)(read - i * seg.J);
            at = i * (u64)seg.L + seg_pos(j);
            len = L - (j >= seg.J1 ? 1u : 0u);
        }
        if (pieces > 1u) {
            const u32 wins = len >= (u32)K ? len - (u32)K + 1u : 0u, run = (wins + pieces - 1u) / pieces, first = piece * run;
            if (first >= wins) return;
            at += first;
            len = (wins - first < run ? wins - first : run) + (u32)K - 1u;
        }
        if constexpr (PACKED) {
            static_assert(!PACKED || K <= 32, "packed input: single-word k-mers");
            roll_read_packed(reinterpret_cast<const u64*>(bases), at, len, (u32)K, [&](u32, u64 fw, u64 rc) {
                const u64 canon = fw < rc ? fw : rc;
                fb.n += 1;
                fb.s0 += canon;
                fb.x0 ^= lex_hash(canon, (u32)K);
                fb.fw += fw;
            });
        } else if constexpr (K <= 32) {
            roll_read<false>(bases + at, len, (u32)K, [&](u32, u64 fw, u64 rc) {
                const u64 canon = fw < rc ? fw : rc;
                fb.n += 1;
                fb.s0 += canon;
                fb.x0 ^= lex_hash(canon, (u32)K);
                fb.fw += fw;
            });
        } else {
            roll_read2<false>(bases + at, len, (u32)K, [&](u32, U128 fw, U128 rc) {
                const U128 c = lt128(fw, rc) ? fw : rc;
                const U128 h = lex_hash128(c, (u32)K);
                fb.n += 1;
                fb.s0 += c.lo;
                fb.s1 += c.hi;
                fb.x0 ^= h.lo;
                fb.x1 ^= h.hi;
            });
        }
    };
    // one tile (or a part of the final partial one) on the per-lane path; `mine`: this lane has something to roll
    auto fallback_read = [&](u64 read, bool mine, u32 piece = 0, u32 pieces = 1) __attribute__((always_inline)) {
        if constexpr (FB_FLUSH) {
            FbAcc fb;
            if (mine) fallback_read_acc(read, fb, piece, pieces);
            emit_sums(wave_sum(fb.n), wave_sum(fb.s0), wave_sum(fb.s1), wave_xor(fb.x0), wave_xor(fb.x1), wave_sum(fb.fw));
        } else {
            if (mine) fallback_read_acc(read, fb_all, piece, pieces);
        }
    };

    // software pipeline: the loads of tile t+1 are issued right after tile t has been packed, so they
    // are in flight during the realign / transpose / item phases of tile t (HBM latency ~4 us under load)
    constexpr int NLD = PACKED ? (NW + 3) / 4 : NW;   // 16-byte loads per lane and tile (packed: 16*L bytes per tile)
    uint4 w[NLD];
    // address = wave-uniform tile base (SGPR pair) + 32-bit per-lane byte offset: no per-chunk 64-bit
    // pointers stay live across the loop
    const u32 lane16 = lane * 16u;
    const u32 last_off = PACKED ? (L - 1u) * 16u : (chunks - 1u) * 16u;
    const bool short_rows = PACKED || chunks < 64u * (NW - 1);   // whole rows of the load grid may lie past the tile
    auto issue_loads_ragged = [&](const TileMeta& m, int row0 = 0, int row1 = 64) {
        // unconditional (a tile outside the frame reads 16 bytes of the queue block instead, and rolls per lane): loads under a
        // branch make hipcc wait with vmcnt(0) where phase A would count them down
        // (the stand-in source: a quiet line of the queue block, NOT the ticket heads -- reads of a line that thousands of atomics
        // hit at the same time took ~2.6 us apiece, 26 us per tile that does not fit)
        const uint8_t* __restrict__ tb = m.fits ? bases + m.base : reinterpret_cast<const uint8_t*>(queue + 544);
        const u32 lo = m.fits ? (m.n_ch - 1u) * 16u : 0u;
        u32 l16 = lane16;                       // (opaque copy: see issue_loads)
        asm volatile("" : "+v"(l16));
        // per-tile descriptor over the tile's chunks: lanes past its end read zeros (see issue_loads)
        uint8_t* const tbu = reinterpret_cast<uint8_t*>(uniform_u64(reinterpret_cast<u64>(tb)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tbu, 0, (int)__builtin_amdgcn_readfirstlane(lo + 16u), 0x00020000);
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            if (it < row0 || it >= row1) continue;
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, l16 + (u32)it * 1024u, 0, BS_LOAD_NT);
            w[it] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    // rows of the prefetch requested late (between the passes of phase D; their registers are free until then): half of a tile's
    // rows -- 5 of 10, 7 of 13, 8 of 16 --, 3 of the 7 (5) rows of the short frames
    // Rows of the next tile requested LATE (between pass 1 and pass 2 of phase D) instead of right behind phase A.  The 10-word
    // frame at up to four windows per lane (k = 22..31 on 150-base reads): none -- since the ticket is no longer waited for at once
    // (see ticket_issue) the whole tile requested early is +2 % (160-168 registers, no spills); at five windows per lane the
    // same cost 8-16 bytes of spills and 7 % in round 4; since round 5's register diet it fits without a spill up to k = 17 (three waves
    // instead of four with five late rows: k = 13 / 17 0.74 -> 0.78) and with ONE late row from k = 18 (k = 21 +1 %; none spills 40 bytes
    // inside the loop, -8 %) -- profiles/r05_late_rows.txt.  (The ragged 10-word frame keeps five late rows: at two waves
    // none measured +3 % at up to four windows per lane and -12 % at five -- profiles/r05_ragged_variants.txt -- and at three waves,
    // where it runs since round 5, the registers are not there.)
    constexpr int LATE = bs_late<K, NW, WPL, PACKED, RAGGED, SEG>();
    u64 tile = ~0ull, next_tile = ~0ull;
    bool seg_ld_next = false;         // SEG: issue_loads is asked for the next tile (nx_g), not for the current one (cur_g)
    auto issue_loads = [&](u64 tile, int row0 = 0, int row1 = 64) {
        const uint8_t* __restrict__ tb = SEG ? bases + (seg_ld_next ? nx_g.base : cur_g.base) : bases + tile * (PACKED ? 16u : 64u) * (u64)L;
        // The per-row offsets are derived afresh from an opaque copy of the lane offset: left to itself hipcc hoists all
        // NLD of them out of the tile loop as zero-extended 64-bit values (24 registers at NLD = 10, and a 64-bit add per
        // row per tile); recomputed they are one 32-bit op each and the loads take the SGPR-base + VGPR-offset form
        // (168 -> 152 registers at k = 31, 240 -> 220 at k = 63; same speed).
        u32 l16 = lane16;
        asm volatile("" : "+v"(l16));
        if constexpr (!PACKED) {
            // (the tile index is wave-uniform, but only readfirstlane tells hipcc so: a descriptor it takes for lane-dependent is
            // fed to every load through a waterfall loop)
            uint8_t* const tbu = reinterpret_cast<uint8_t*>(uniform_u64(reinterpret_cast<u64>(tb)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tbu, 0, (int)__builtin_amdgcn_readfirstlane(SEG ? (seg_ld_next ? nx_g.nbytes : cur_g.nbytes) : chunks * 16u), 0x00020000);
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                if (it < row0 || it >= row1) continue;
                typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, l16 + (u32)it * 1024u, 0, BS_LOAD_NT);
                w[it] = make_uint4(v.x, v.y, v.z, v.w);
            }
            return;
        }
        // (packed input: plain loads)
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            if (it < row0 || it >= row1) continue;
            // lanes past the tile end re-read its last chunk: no branch, so all loads of a tile sit in
            // one basic block and stay in flight together (a guarded load would be fenced by vmcnt(0))
            u32 off = l16 + (u32)it * 1024u;
            // (the bound of a row that cannot leave the tile is ~0: one v_min per row, the choice is made on the scalar side)
            const u32 row_bound = (it == NLD - 1 || short_rows) ? last_off : 0xFFFFFFFFu;
            off = off < row_bound ? off : row_bound;
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(tb + off));  // streamed once
            w[it] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    // Dynamic tile queue.  Static striding ends with a long under-occupied tail because VALU arbitration
    // favours the oldest waves on a SIMD (measured: the 3 blocks of a CU finished at 3.2 / 3.9 / 4.7 ms with
    // equal work).  Tiles are handed out ONE at a time so that the tiles in flight across the chip stay
    // adjacent in memory (handing out runs of 8 tiles cost 30 % of read bandwidth: 5.0 vs 7.1 TB/s in a
    // compute-free build); to stay far below the ~88 atomics/us a single word sustains, there are NQ queue
    // heads, each on its own cache line, head q owning the tiles == q (mod NQ).  A head is shared by the
    // three co-resident blocks of 8 CUs (old and young waves alike), so heads drain at equal rates.
    // (Handing each head a contiguous region instead changed nothing: 2.50 vs 2.55 ms compute-free, equal in the full kernel.)
    constexpr u32 NQ = 32;
    const u64 n_static = (u64)gridDim.x * 4u;           // the tiles the waves own without a ticket: wave w starts with tile w (below); head q owns the tiles == q (mod NQ) past them
    // (a grid of fewer than 256 blocks -- a small batch -- spreads over all 32 heads too: crowded on gridDim / 8 of them, most waves found their
    // head drained at once and walked the others in step, one round trip per head: 1e4 reads took longer than 1e5)
    u32 qid = ((blockIdx.x & 255u) * NQ) / (gridDim.x < 256u ? gridDim.x : 256u);
    u32 heads_left = NQ;                                // non-zero: some head may still hold a ticket (dequeue() clears it)
    // Every ticket from the NEXT head (round 4): a wave that stayed with "its" head tied the head's pace to the 24 blocks that
    // share it, the heads drifted apart, and with them the addresses in flight -- the HBM stream is measurably better when the
    // tiles being read lie close together (+1..2 %; tools/stream_patterns.hip).  Until the first head is seen exhausted: from
    // then on a wave stays with the nearest head that still holds a ticket (dequeue).
    bool rot = !RAGGED;   // (the ragged variants measured 10 % SLOWER with it, same box, same day: profiles/r04_rotation_ragged.txt)
    // A head seen drained (round 5): ALL heads at a glance -- lane i reads head i's counter, coherently -- and the next ticket from the
    // nearest head that still holds one.  Until then a wave swept the 32 heads one synchronous device atomic after the other before it
    // believed the queue empty: ~40 us at the end of EVERY launch, all waves at once (1e6 reads: 38 of a wave's 63 us; at 1e8 reads, where the
    // waves do not all get there together, 15 us = 0.6 % of the launch: profiles/r05_small_batches.txt, r05_box_spread.txt).  Now the end
    // costs one failed atomic and one look.  (A head's counter only grows, and a
    // failed atomic leaves it drained: the loop ends.)
    auto dequeue = [&]() -> u64 {
        for (;;) {
            // (the low word: a head hands out fewer than 2^32 tickets; lanes 32..63 look at the heads again -- same answer, no branch)
            const u32 ln = lane_now() & (NQ - 1u);
            const u32 c = __hip_atomic_load(reinterpret_cast<const u32*>(queue) + ln * 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 live = (u32)__ballot((u64)c * NQ + ln + n_static < n_full);
            if (live == 0u) break;
            // (the nearest live head counted from a place that differs from wave to wave: the waves that fail together do not all fall on one head)
            const u32 at = (qid + (u32)wave_id) & (NQ - 1u);
            const u32 from = (live >> at) | (at ? live << (NQ - at) : 0u);             // bit i: head at + i
            qid = (at + (u32)__builtin_ctz(from)) & (NQ - 1u);
            unsigned long long v = 0;
            if (lane_now() == 0u) v = atomicAdd(queue + qid * 16u, 1ull);               // heads are 128 bytes apart
            const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
            const u64 t = (((u64)hi << 32) | lo) * NQ + qid + n_static;
            if (t < n_full) return t;
        }
        heads_left = 0u;
        return ~0ull;
    };
    // The ticket for tile t+2 is requested in iteration t right behind phase A -- after the wave has taken its rows and asked for
    // the first rows of tile t+1 -- and looked at when the iteration ends: phases B-D later.  (Requested at the very end of the
    // iteration, as rounds 2-4 had it, the atomic was the youngest operation in flight when phase A waited for its last rows,
    // and loads and atomics complete in one order on gfx9: every tile waited for a fresh device atomic.  Reading the return
    // value right away, as dequeue() does, costs the same round trip.)
    // (BOTH words of the atomic's return stay live until ticket_take (round 6): only the low one is used -- a head hands out fewer than 2^32
    // tickets, launch_bs checks the tile count -- but the atomic writes a register PAIR when it returns, and with the high word dead hipcc
    // hands that register to the next instruction that needs one, behind an `s_waitcnt vmcnt(0)`: the wave then waits for the ticket
    // AND for the rows just requested at the top of phase B, every tile.  Which variants were hit was the allocator's lottery: round 5's
    // k = 63 was not, this round's was, 2.6 % (profiles/r06_ab_r5_r6.txt).)
    unsigned long long pend = 0;
    u32 pend_qid = 0;
    auto ticket_issue = [&]() {
        pend_qid = qid;
        if (heads_left != 0u && lane == 0) {
            unsigned long long one = 1ull;   // (made here: hoisted, the constant holds a register pair across the tile loop)
            asm volatile("" : "+v"(one));
            // The head's address goes through a VGPR the compiler cannot see through: with a wave-uniform address hipcc's atomic
            // optimizer rewrites the add (a scan over the active lanes, ONE atomic, the result handed back to the lanes) and needs
            // the returned value AT ONCE -- an `s_waitcnt vmcnt(0)` right behind the atomic: the device-atomic round trip and the
            // rows of the next tile, every tile (~17 % of a wave's cycles sat there since round 3: profiles/r04_phase_timing*.txt).
            u32 zero = 0;
            asm volatile("" : "+v"(zero));
            pend = atomicAdd(queue + qid * 16u + zero, one);
        }
        if (rot) qid = (qid + 1u) & (NQ - 1u);
    };
    auto ticket_take = [&]() -> u64 {
        if (heads_left == 0u) return ~0ull;
        {
            u32 pend_hi = (u32)(pend >> 32);
            asm volatile("" : : "v"(pend_hi));      // (the use that keeps the pair together: no instruction)
        }
        const u32 lo = __builtin_amdgcn_readfirstlane((u32)pend);
        const u64 t = (u64)lo * NQ + pend_qid + n_static;
        if (t < n_full) return t;
        rot = false;
        qid = (pend_qid + 1u) & (NQ - 1u);   // that head is drained: move on, synchronously (rare)
        return dequeue();
    };
    auto prefetch = [&](u64 t, u64 fallback_t, int row0 = 0, int row1 = 64) {   // clamped => unconditional, one basic block, pinned by sched barriers
        const u64 nxt = t < n_full ? t : fallback_t;
        if constexpr (SEG) seg_ld_next = t < n_full;
        __builtin_amdgcn_sched_barrier(0);
        issue_loads(nxt, row0, row1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- per-tile phases
    // encode16, hand-scheduled for the VALU co-issue rule of gfx950 (see phase D, pass 2): the half-rate instructions
    // (v_perm_b32 validation look-ups; v_dot4_u32_u8 packs + v_lshl_or_b32 merges) run as two raised-priority runs
    // (one asm statement each, so that nothing else is scheduled into them), the full-rate ones (v_and, v_xor,
    // v_bitop3, v_lshrrev) around them at base priority.
    const u32 k55 = 0x55555555u;   // (a literal: an SGPR or literal source does not keep a full-rate instruction from pairing -- profiles/r02_valu_coissue_ubench9.txt)
    auto encode_prio = [&](const uint4& wv, u32& bad) -> u32 {   // bad = expected letters ^ bytes, ORed over the chunk's four dwords (a valid chunk: 0x00 / 0x20 in every byte)
        constexpr u32 TBL_LO = 0x00430041u, TBL_HI = 0x00470054u, W4 = 0x40100401u;   // as in encode16
        u32 t0 = wv.x & 0x06060606u, t1 = wv.y & 0x06060606u, t2 = wv.z & 0x06060606u, t3 = wv.w & 0x06060606u;
        u32 e0, e1, e2, e3;
        asm volatile("s_setprio 3\n\t"
                     "v_perm_b32 %0, %8, %9, %4\n\t"
                     "v_perm_b32 %1, %8, %9, %5\n\t"
                     "v_perm_b32 %2, %8, %9, %6\n\t"
                     "v_perm_b32 %3, %8, %9, %7\n\t"
                     "s_setprio 0"
                     : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
                     : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "s"(TBL_HI), "v"(TBL_LO));
        // bad |= expected ^ actual, one v_bitop3_b32 per dword (S0 | (S1 ^ S2) = 0xF6) instead of 4 v_xor + 2 three-input ORs
        if constexpr (PARK) bad = e0 ^ wv.x;                                   // a word per chunk
        else bad = __builtin_amdgcn_bitop3_b32(bad, e0, wv.x, 0xF6);            // one running word per lane
        bad = __builtin_amdgcn_bitop3_b32(bad, e1, wv.y, 0xF6);
        bad = __builtin_amdgcn_bitop3_b32(bad, e2, wv.z, 0xF6);
        bad = __builtin_amdgcn_bitop3_b32(bad, e3, wv.w, 0xF6);
        // 2 * (4 bases in 8 bits) per dword, merged to 16 bases in 32 bits: ((d0 | d1<<8 | d2<<16) >> 1) | d3<<23
        asm volatile("s_setprio 3\n\t"
                     "v_dot4_u32_u8 %0, %0, %4, 0\n\t"
                     "v_dot4_u32_u8 %1, %1, %4, 0\n\t"
                     "v_dot4_u32_u8 %2, %2, %4, 0\n\t"
                     "v_dot4_u32_u8 %3, %3, %4, 0\n\t"
                     "s_nop 0\n\t"   // gfx940+: a VALU read of a DOT result needs 3 wait states (hipcc does not look inside asm)
                     "v_lshl_or_b32 %0, %1, 8, %0\n\t"
                     "v_lshl_or_b32 %0, %2, 16, %0\n\t"
                     "v_lshrrev_b32 %0, 1, %0\n\t"
                     "v_lshl_or_b32 %0, %3, 23, %0\n\t"
                     "s_setprio 0"
                     : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)
                     : "s"(W4));
        return __builtin_amdgcn_bitop3_b32(t0 >> 1, t0, k55, 0x6c);   // internal (ACTG) -> naive_impl (ACGT) codes
    };
    auto phase_A = [&]() -> bool {   // pack + validate the tile sitting in w[] into the packed LDS buffer; true: some chunk holds an invalid byte
        // one chunk: its packed word and, XOFF dwords behind it, its validation word (hipcc merges the two stores into ONE
        // ds_write2st64_b32).  `any` collects the validation words two rows at a time (a three-input OR at full rate).
        u32 any = 0, held = 0;
        auto row = [&](const int it, const bool mine) {
            if constexpr (PARK) {
                u32 r = 0;
                if (mine) {
                    const u32 code = encode_prio(w[it], r);
                    P[1u + (u32)it * 64u + lane] = code;
                    P[1u + XOFF + (u32)it * 64u + lane] = r;
                }
                if (it & 1) any = __builtin_amdgcn_bitop3_b32(any, held, r, 0xFE);
                else held = r;
            } else {
                if (mine) P[1u + (u32)it * 64u + lane] = encode_prio(w[it], any);
            }
        };
        if constexpr (PACKED) {      // already 2-bit codes: 16 bytes = 4 packed dwords per lane and load
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const u32 c = it * 64u + lane;
                if (c < L) *reinterpret_cast<uint4*>(P + PAD + 4u * c) = w[it];
            }
            return false;
        } else {
            // the chunks of the tile: ragged -- cur_m.n_ch from its aligned start (neighbouring tiles' bytes at both ends); SEG -- the
            // tile's own count; uniform -- `chunks`
            const u32 n_ch = RAGGED ? cur_m.n_ch : SEG ? (cur_g.nbytes + 15u) >> 4 : chunks;
            if (!SEG && n_ch >= 64u * (NW - 1)) {   // wave-uniform: only the last row is partial (64 reads of 150: 600 or 601 chunks)
#pragma unroll
                for (int it = 0; it < NW - 1; ++it) row(it, true);
                row(NW - 1, (NW - 1) * 64u + lane < n_ch);
            } else {
                // (row bound on the scalar side, compared with the lane id: ten per-row chunk indices would sit in registers across the tile loop)
#pragma unroll
                for (int it = 0; it < NW; ++it) row(it, (int)lane < (int)n_ch - 64 * it);
            }
            if (PARK && (NW & 1)) any |= held;
            return __any(chunk_has_invalid(any));
        }
    };
    auto lds_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto phase_BC = [&]() {
        // ---- B. this lane's read, realigned: F[g] = bases [16g, 16g+16)
        u32 F[NXT];
        if constexpr (RAGGED) {
            posF = cur_m.rel + 16u * PAD;
            qF = posF >> 4;
            aF = 2u * (posF & 15u);
        }
        if constexpr (SEG) {
            posF = seg_rel + cur_g.lead + 16u * PAD;
            qF = posF >> 4;
            aF = 2u * (posF & 15u);
        }
        {
            u32 R[NW + 1];
#pragma unroll
            for (int j = 0; j <= NW; ++j) R[j] = P[qF + j];
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int g = 0; g < NW; ++g) F[g] = alignbit(R[g + 1], R[g], aF);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (RAGGED) {
            const u32 len = cur_m.len >= (u32)K ? cur_m.len : 0u;   // a read shorter than k owns no window: it is blanked out entirely
            // bases past the end of the read belong to the next read.  They stay: a window that holds one is masked out of m by
            // its validity plane, and the plane totals count a plane through the validity plane of its base (below)
            // the last K-1 bases of the read, base len-K+1+i at position i of the NE dwords
            const u32 posE = len ? posF + len - (u32)(K - 1) : posF;
            const u32 qE = posE >> 4, aE = 2u * (posE & 15u);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                u32 v = alignbit(P[qE + e + 1], P[qE + e], aE);
                constexpr int tail = (K - 1) - 16 * (NE - 1);          // bases in the last dword (1..16)
                if (e == NE - 1 && tail < 16) v &= (1u << (2 * tail)) - 1u;
                F[NW + e] = len ? v : 0u;
            }
            // validity: bit o of word o/32 = window o lies inside the read
            const u32 wr = len ? len - (u32)K + 1u : 0u;
            atomicAdd(reinterpret_cast<unsigned long long*>(&NVR[lane_now()]), (unsigned long long)wr);   // ds_add_u64, no return
            {
                // Validity planes without transposes: V_o (bit r = read r owns window o <=> wr_r > o) = OR over i > o of E_i,
                // E_i = the reads with wr == i.  The reads mark E (in the plane area, free until the planes are stored at the end
                // of this phase); lane p of a half-wave then takes the NV planes o = NV*(31-p) .. +NV-1 -- the lanes below it hold
                // the higher o -- ORs its own marks downwards and gets the marks of all higher o from a prefix-OR across the lanes
                // (4 row shifts + the row broadcast).  The reads that share lane 0's wr (all of them, in untrimmed FASTQ) are
                // marked by ONE lane per half with the whole ballot: 64 atomics on one LDS word would serialise.
                // (Round 6) A tile of trimmed FASTQ holds a handful of reads that differ from the rest: with at most RG_EXC of them the
                // planes follow from the common count and the exceptions themselves -- V_o = (ref > o ? the reads at ref : 0) | the
                // exceptions with wr > o, each broadcast from its lane -- no marks, no LDS atomics, one fence instead of three.
                constexpr u32 EMS = 32u * NV + 4u;      // dwords per half-wave: E_0 .. E_(32 NV)
                constexpr u32 RG_EXC = 4u;
                const u32 ref = (u32)__builtin_amdgcn_readfirstlane(wr);
                const bool is_ref = wr == ref;
                const u64 bref = __ballot(is_ref);
                const u32 ob = (u32)NV * (31u - p);
                u32* const vh = VAL + half * VS + ob;
                if ((u32)__builtin_popcountll(~bref) <= RG_EXC) {          // (wave-uniform)
                    const u32 refm = half_word(bref);
                    u32 e[NV];
#pragma unroll
                    for (int j = 0; j < NV; ++j) e[j] = ref > ob + (u32)j ? refm : 0u;
                    for (u64 left = ~bref; left != 0ull; left &= left - 1ull) {
                        const u32 sl = (u32)__builtin_ctzll(left);
                        const u32 wr_s = (u32)__builtin_amdgcn_readlane((int)wr, (int)sl);
                        const u32 add = (lane_now() >> 5) == (sl >> 5) ? 1u << (sl & 31u) : 0u;
#pragma unroll
                        for (int j = 0; j < NV; ++j) e[j] |= wr_s > ob + (u32)j ? add : 0u;
                    }
#pragma unroll
                    for (int j = 0; j < NV; ++j) vh[j] = e[j];
                } else {
                    u32* const EM = PL;
                    static_assert(2u * EMS <= 2u * PLANES, "end marks fit the plane area");
                    // (addresses from the lane id as it is now: hoisted out of the tile loop they are spilled, and reloaded behind the next tile's rows)
                    for (u32 i = 4u * lane_now(); i < 2u * EMS; i += 256u) *reinterpret_cast<uint4*>(EM + i) = make_uint4(0u, 0u, 0u, 0u);
                    lds_fence();
                    u32* const emh = EM + (lane_now() >> 5) * EMS;
                    if (!is_ref) atomicOr(emh + wr, 1u << (lane_now() & 31u));
                    if (p == 0u) atomicOr(emh + ref, half ? (u32)(bref >> 32) : (u32)bref);
                    lds_fence();
                    u32 e[NV];
#pragma unroll
                    for (int j = 0; j < NV; ++j) e[j] = emh[ob + 1u + j];
#pragma unroll
                    for (int j = NV - 2; j >= 0; --j) e[j] |= e[j + 1];
                    u32 x = e[0];
                    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
                    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
                    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
                    x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
                    u32 c = (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, true);        // the lanes below, within the row of 16
                    c |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);      // rows 1 and 3: all of the row before
#pragma unroll
                    for (int j = 0; j < NV; ++j) vh[j] = e[j] | c;
                }
                lds_fence();
            }
        }
        // ---- C. transpose each 32 reads x 32 bits block across the 32 lanes of the half-wave (no LDS round trips):
        //      butterfly stage d exchanges with lane^d and keeps/merges the bits whose index has bit d clear/set.
        //        d=16, 8: ds_swizzle xor-d (the LDS crossbar, no VALU) + byte merge (v_perm_b32)
        //        d=4    : ds_swizzle xor-4                             + rotate (v_alignbit) + bit select (v_bitop3)
        //        d=2, 1 : DPP quad_perm                                + rotate + bit select
        // Stage-major order: every butterfly stage runs over all NW groups, with its half-rate instructions
        // (v_perm_b32 / v_alignbit_b32 / DPP moves) as one raised-priority run and its full-rate bit selects after it.
        {
            u32 Y[NXT];
            const u32 c_sh2 = tr_sh[2], c_sh3 = tr_sh[3], c_sh4 = tr_sh[4], c_sel16 = tr_sel16;
            const u32 c_keep2 = tr_keep[2], c_keep3 = tr_keep[3], c_keep4 = tr_keep[4], c_sel8 = tr_sel8;
#define KMX_HRUN_BEGIN __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(3);
#define KMX_HRUN_END __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int g = 0; g < NXT; ++g) Y[g] = (u32)__builtin_amdgcn_ds_swizzle((int)F[g], (16 << 10) | 0x1f);
            KMX_HRUN_BEGIN
#pragma unroll
            for (int g = 0; g < NXT; ++g) F[g] = __builtin_amdgcn_perm(Y[g], F[g], c_sel16);
            KMX_HRUN_END
#pragma unroll
            for (int g = 0; g < NXT; ++g) Y[g] = (u32)__builtin_amdgcn_ds_swizzle((int)F[g], (8 << 10) | 0x1f);
            KMX_HRUN_BEGIN
#pragma unroll
            for (int g = 0; g < NXT; ++g) F[g] = __builtin_amdgcn_perm(Y[g], F[g], c_sel8);
            KMX_HRUN_END
#pragma unroll
            for (int g = 0; g < NXT; ++g) Y[g] = (u32)__builtin_amdgcn_ds_swizzle((int)F[g], (4 << 10) | 0x1f);
            KMX_HRUN_BEGIN
#pragma unroll
            for (int g = 0; g < NXT; ++g) Y[g] = alignbit(Y[g], Y[g], c_sh2);
            KMX_HRUN_END
#pragma unroll
            for (int g = 0; g < NXT; ++g) F[g] = bitsel(F[g], Y[g], c_keep2);
#pragma unroll
            for (int st = 3; st < 5; ++st) {
                KMX_HRUN_BEGIN
#pragma unroll
                for (int g = 0; g < NXT; ++g)
                    Y[g] = st == 3 ? (u32)__builtin_amdgcn_update_dpp(0, (int)F[g], 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true)
                                   : (u32)__builtin_amdgcn_update_dpp(0, (int)F[g], 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
#pragma unroll
                for (int g = 0; g < NXT; ++g) Y[g] = alignbit(Y[g], Y[g], st == 3 ? c_sh3 : c_sh4);
                KMX_HRUN_END
#pragma unroll
                for (int g = 0; g < NXT; ++g) F[g] = bitsel(F[g], Y[g], st == 3 ? c_keep3 : c_keep4);
            }
            {
                // plane q = 32g + p  <->  base beta = 16g + p/2, bit p & 1.  One lane-dependent base and a compile-time offset per
                // group (indexing PL[] with the whole expression costs an address register per group: the u32 sum could wrap, so
                // hipcc cannot split it into register + immediate offset)
                const u32 b0 = p >> 1;
                const u32 slot0 = ROT ? (b0 % (u32)RW) * S2 + (b0 / (u32)RW) : b0;
                u32* const pst = PL + (half * SP + 2u * slot0 + (p & 1u));
#pragma unroll
                for (int g = 0; g < NW; ++g) pst[ROT ? (32 / RW) * g : 32 * g] = F[g];
            }
            if constexpr (RAGGED) {
                // plane (g, p) holds base beta = 16g + p/2 of the set's reads: only the reads that HAVE a base beta count
                // (len > beta <=> wr > beta - (K-1): validity plane V_(beta-K+1); below K-1 every read that owns a window at all)
                // (from the lane id as it is now: held across the tile loop these two addresses are what the three-wave variants spill, and
                // reload right here behind the next tile's rows)
                const u32 ln_v = lane_now();
                const u32* const vh = VAL + (ln_v >> 5) * VS;
                const u32 b0 = (ln_v & 31u) >> 1;
#pragma unroll
                for (int g = 0; g < NW; ++g) {
                    u32 vm;
                    if (16 * g + 15 < K - 1) vm = vh[0];
                    else if (16 * g >= K - 1) vm = (vh + b0)[16 * g - (K - 1)];
                    else { const u32 beta = 16u * g + b0; vm = vh[beta >= (u32)(K - 1) ? beta - (u32)(K - 1) : 0u]; }
                    Y[g] = F[g] & vm;
                }
                KMX_HRUN_BEGIN
#pragma unroll
                for (int g = 0; g < NW; ++g) Y[g] = (u32)__builtin_popcount(Y[g]);
#pragma unroll
                for (int g = NW; g < NW + NE; ++g) Y[g] = (u32)__builtin_popcount(F[g]);
                KMX_HRUN_END
            } else {
            KMX_HRUN_BEGIN
#pragma unroll
            for (int g = 0; g < NW + NE; ++g) Y[g] = (u32)__builtin_popcount(F[g]);
            KMX_HRUN_END
            }
            {
                u32* const tot_l = TOT + lane;
#pragma unroll
                for (int g = 0; g < NW; ++g) atomicAdd(tot_l + 64 * g, Y[g]);
            }
            if constexpr (SEG) {
                // the planes of bases W-1 .. W+K-2 (the last window) restricted to the short segments: what the closed form takes back out
                const u32 sh_half = half_word(seg_short);
                if (seg_short != 0ull) {
                    u32* const tots_l = TOTS + lane;
#pragma unroll
                    for (int g = 0; g < NW; ++g) {
                        const u32 gs = (u32)g - ((W - 1u) >> 4);  // (wave-uniform: 16 g + 15 >= W - 1 from the first one on)
                        if (gs >= TOTS_G) continue;
                        atomicAdd(tots_l + 64u * gs, (u32)__builtin_popcount(F[g] & sh_half));
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) atomicAdd(&QT[e * 64 + lane], Y[NW + e]);               // ragged: read-end planes, totals only
#undef KMX_HRUN_BEGIN
#undef KMX_HRUN_END
        }
        lds_fence();

    };
    auto phase_D = [&](const bool run) {
        // ---- D. a lane handles the WPL windows o..o+WPL-1 of one set (o = WPL*group): they share the planes of
        //      bases o..o+K+WPL-2, streamed from LDS as u64 (2 planes per base):
        //      pass 1 = WPL interleaved fw<rc ripples -> the mask words m; pass 2 = the masked plane counts, on the matrix pipe.
        // `run` (wave-uniform) = false: a tile that is not scanned here (it rolled per lane); with LATE > 0 the call
        // is still made, for the ONE static site of the late prefetch rows between the passes (a second site in another
        // branch gets its own registers and is hoisted above the branch).
        // The launchers pick WPL = ceil(W / 32): the 32 lanes of a half-wave hold all windows of a set, ONE round (launch_bs
        // checks it).  LATE > 0 relies on it: with a compile-time single trip the late prefetch rows have one definition on
        // every path.  Otherwise the round stays a real loop (runtime bound, one trip): as straight-line code hipcc hoists
        // the plane requests of pass 1 across phase C and the kernels gain 20-60 registers.
        const u32 n_rounds = LATE > 0 ? 1u : (run ? rounds : 0u);
#pragma unroll 1
        for (u32 r = 0; LATE > 0 ? r < 1u : r < n_rounds; ++r) {
            // a half-wave per set, lane p of it the windows WPL p .. WPL p + WPL - 1 (the launchers pick WPL = ceil(W / 32)): the
            // lanes past the last group compute on whatever lies behind the planes and count nothing (nwin = 0), but they write
            // the zero words that complete the set's 32 WPL mask words -- the A operands of the matrix products.
            const u32 set = lane >> 5;
            const bool active = (lane & 31u) < NG;
            const u32 o = (u32)WPL * (lane & 31u);
            const u32 nwin = active ? (W - o < (u32)WPL ? W - o : (u32)WPL) : 0u;   // valid windows in this group
            // base o+i  ->  u64 index (i % WPL) * S2 + o / WPL + i / WPL in the rotated layout (o is a multiple of WPL), else o + i
            // (an address-space-3 pointer: a volatile access through a generic pointer stays a flat load)
            typedef const volatile u64 __attribute__((address_space(3))) * lds_cvu64p;
            const lds_cvu64p src = (lds_cvu64p)(reinterpret_cast<const u64*>(PL + set * SP) + (ROT ? (o / (u32)RW) : o));
#define KMX_PLANE(i) src[ROT ? (((i) % RW) * S2 + ((i) / RW)) : (i)]
            u32 lt[WPL];
#pragma unroll
            for (int w = 0; w < WPL; ++w) lt[w] = 0u;
            if (run) {
                u64 Pv[K + WPL - 1];
                // ripple from the least significant deciding pair (j = ceil(K/2)-1) to the most significant (j = 0);
                // window w compares fw base K-1-j (plane o+w+K-1-j) with rc base = ~(fw base j) (plane o+w+j).
                // Step J0 needs the planes J0 .. K-1-J0+WPL-1, every later step one more on either side (j and
                // K-1-j+WPL-1).  They are requested P1D steps ahead (one step = 2*WPL bit ops, far less than an LDS
                // round trip): left to hipcc, every second step ended in `ds_read ... s_waitcnt lgkmcnt(0)`.
                constexpr int J0 = (K + 1) / 2 - 1, JEND = 0, P1D = 2;
#pragma unroll
                for (int i = J0; i <= K - 1 - J0 + WPL - 1; ++i) Pv[i] = KMX_PLANE(i);
#pragma unroll
                for (int d = 1; d < P1D; ++d) {
                    if (J0 - d >= JEND) {
                        Pv[J0 - d] = KMX_PLANE(J0 - d);
                        Pv[K - 1 - (J0 - d) + WPL - 1] = KMX_PLANE(K - 1 - (J0 - d) + WPL - 1);
                    }
                }
#define KMX_PLANE_AT(p, i) (p)[ROT ? (((i) % RW) * S2 + ((i) / RW)) : (i)]
                auto fetch = [&](int j) {
                    // an (empty) asm ties the address to the ripple state of the previous step, so the two requests are
                    // issued here and not hoisted to the top of the unrolled loop (34 planes live = spills)
                    // (an index, not the pointer: an opaque pointer would lose its LDS address space and turn the reads into flat loads)
                    // (the LDS byte address itself goes through the asm and comes back as an address-space-3 pointer: a zero
                    // offset added to `src` costs a v_mov and a v_lshl_add per step, 30 VALU instructions a tile)
                    asm volatile("" : : "v"(lt[0]) : "memory");   // a compiler-level fence only: no instruction, no copy
                    Pv[j] = KMX_PLANE_AT(src, j);
                    Pv[K - 1 - j + WPL - 1] = KMX_PLANE_AT(src, K - 1 - j + WPL - 1);
                };
#pragma unroll
                for (int j = J0; j >= JEND; --j) {
                    if (j - P1D >= JEND) fetch(j - P1D);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < WPL; ++w) {
                        const int ia = K - 1 - j + w, iq = j + w;
                        const u32 a0 = (u32)Pv[ia], a1 = (u32)(Pv[ia] >> 32);
                        const u32 q0 = (u32)Pv[iq], q1 = (u32)(Pv[iq] >> 32);
                        lt[w] = ripple(lt[w], a0, q0);
                        lt[w] = ripple(lt[w], a1, q1);
                    }
                }
            }
            u32 m[WPL];
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                if constexpr (RAGGED) {   // only the reads that own window o+w (none of them past the frame)
                    // (an idle lane and a window past the frame read one of the 8 zero words behind the set's planes)
                    m[w] = lt[w] & (VAL + set * VS + (active ? o : 32u * NV))[w];
                } else {
                    m[w] = ((u32)w < nwin) ? lt[w] : 0u;
                }
            }
            if constexpr (SEG) {
                if (seg_short != 0ull) {      // (wave-uniform)
                    const u32 keep = ~half_word(seg_short);
                    const u32 ws = W - 1u - o;                 // which of the lane's windows is window W-1 (none: >= WPL)
#pragma unroll
                    for (int w = 0; w < WPL; ++w) m[w] = (u32)w == ws ? m[w] & keep : m[w];
                }
            }
#pragma unroll
            for (int w = 0; w < WPL; ++w) pc_acc(mcnt, m[w]);   // (a tile that is not scanned: lt == 0, nothing is added)
            if constexpr (LATE > 0) {   // the rest of the next tile's rows: the registers of pass 1's plane window are free now
                if constexpr (RAGGED) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (r == 0 && next_tile < n_full) issue_loads_ragged(nx_m, NLD - LATE, NLD);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    if (r == 0) prefetch(next_tile, tile, NLD - LATE, NLD);
                }
            }
            if (!run) break;
            asm volatile("" ::: "memory");
            {
                // ---- pass 2 on the matrix pipe.  C[t][b] = sum_o sum_r m[o][r] & plane[o+t][b][r] is the sum along the diagonal
                // beta - o = t of G[o][beta] = sum_r m[o][r] * plane[beta][r], a 0/1 matrix product over the tile's 64 reads:
                // v_mfma_scale_f32_32x32x64_f8f6f4, FP4 operands (fp4_operand_a / _b), unit scales.  Rows = the 32 windows of a
                // block, columns = the 32 planes of 16 bases, K = reads: lanes 0..31 carry set 0, lanes 32..63 set 1 -- which is
                // how phase C left the planes (lane (half, p) holds plane 32 g + p of set `half`) and how the mask words come back
                // from LDS below.  Per tile: WPL + NW ds_read_b32, 7 WPL + 5 NW VALU instructions for the operands and at most
                // NAB WPL matrix instructions (16 at k = 31, L = 150), instead of a v_and + v_bcnt pair per (window, plane) -- 504 of
                // the tile's 1007 VALU instructions in rounds 1-3 (597 now).  A weighted form (nibble = the 2-bit code of a base,
                // half the matrix instructions and accumulators, no hash fold) measured 3 % slower at three or four waves per SIMD
                // (profiles/r04_mfma_variants.txt).
                u32* const MW = P;          // the packed reads are dead since phase B
                u32* const mwr = MW + set * (32u * WPL) + (u32)WPL * (lane & 31u);
                if constexpr (WPL == 4) {
                    *reinterpret_cast<uint4*>(mwr) = make_uint4(m[0], m[1], m[2], m[3]);
                } else if constexpr (WPL == 2) {
                    *reinterpret_cast<uint2*>(mwr) = make_uint2(m[0], m[1]);
                } else if constexpr (WPL == 8) {
                    reinterpret_cast<uint4*>(mwr)[0] = make_uint4(m[0], m[1], m[2], m[3]);
                    reinterpret_cast<uint4*>(mwr)[1] = make_uint4(m[4], m[5], m[6], m[7]);
                } else {
#pragma unroll
                    for (int w = 0; w < WPL; ++w) mwr[w] = m[w];
                }
                lds_fence();
                const u32 pp = lane & 31u;
                u32 ab[WPL];
#pragma unroll
                for (int i = 0; i < WPL; ++i) ab[i] = MW[set * (32u * WPL) + 32u * i + pp];
                // the lane's own planes, where phase C put them
                const u32 b0 = pp >> 1;
                const u32 slot0 = ROT ? (b0 % (u32)RW) * S2 + (b0 / (u32)RW) : b0;
                const u32* const prd = PL + (set * SP + 2u * slot0 + (pp & 1u));
                const int unit = 0x7F7F7F7F;    // E8M0 scale 2^0 in every byte
                const int rows_lo = pp < 16u ? unit : 0, rows_hi = pp < 16u ? 0 : unit;   // the shared accumulator block: which rows of A count
                bs_v8i A[WPL];
#pragma unroll
                for (int g = 0; g < NW; ++g) {
                    const bs_v8i B = fp4_operand_b(prd[ROT ? (32 / RW) * g : 32 * g]);
                    // (window blocks in ascending order for even g, descending for odd g: two products into the shared block are then
                    // never back to back)
#pragma unroll
                    for (int ii = 0; ii < WPL; ++ii) {
                        const int i = (g & 1) ? WPL - 1 - ii : ii;
                        const int q = g - 2 * i;
                        if (q < 0 || q >= NAB) continue;
                        if (32u * (u32)i >= W) continue;             // (wave-uniform: a window block past the last window -- the two-word k on the long frames run with more windows per lane than their reads have)
                        if (q == 0) A[i] = fp4_operand_a(ab[i]);     // first use of window block i
                        if (q == 0)
                            acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[i], B, acc[0], 4, 4, 0, rows_lo, 0, unit);
                        else if (q == NAB - 1)
                            acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[i], B, acc[0], 4, 4, 0, rows_hi, 0, unit);
                        else
                            acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[i], B, acc[q], 4, 4, 0, unit, 0, unit);
                    }
                }
            }
        }
#undef KMX_PLANE
#undef KMX_PLANE_AT
        if (run) n_bs_tiles += 1;
        if constexpr (SEG) {
            if (run) n_short += (u32)__builtin_popcountll(seg_short);
        }
        if (run && (n_bs_tiles & (BS_FOLD_TILES - 1u)) == 0u) fold_acc();   // (wave-uniform, rare: keeps every fp32 accumulator an exact integer)
    };

    // Pipeline order 1: [A of tile t] [issue loads of tile t+1] [B,C,D of tile t]
    // A wave's FIRST tile is its own number -- no ticket -- and the ticket for its second is in flight while the first tile's rows
    // are (round 6: two synchronous device atomics used to open every launch, ~2 us before the first byte was asked for).  The
    // heads hand out the tiles from n_static on.
    tile = wave_id < n_full ? wave_id : ~0ull;
    ticket_issue();
    if constexpr (RAGGED) {
        if (tile < n_full) meta_issue(tile);
        next_tile = uniform_u64(ticket_take());
        if (tile < n_full) {
            meta_finish(cur_m);
            issue_loads_ragged(cur_m);
        }
        if (next_tile < n_full) meta_issue(next_tile);
    } else {
        if constexpr (SEG) {
            if (tile < n_full) seg_geom(tile, cur_g);
        }
        if (tile < n_full) issue_loads(tile);
        next_tile = uniform_u64(ticket_take());
        if constexpr (SEG) {
            if (next_tile < n_full) seg_geom(next_tile, nx_g);
        }
    }
    while (tile < n_full) {
        if constexpr (SEG) seg_lane(cur_g);
        // A tile with an invalid byte: which reads touch a bad chunk?  Every lane ORs the validation words of its read's chunks (phase A
        // left them in the plane area, free until phase C) -- a chunk shared by two reads marks both, the sweep looks at the bytes -- and
        // the reads' mask goes to the array behind queue[515].  LDS reads and ONE global store.
        auto mark_dirty_reads = [&]([[maybe_unused]] const u32 n_chunks) -> bool {
            if (masks == nullptr) {
                if constexpr (!RAGGED) __builtin_trap();   // (the host side always provides the array)
                return false;
            }
            u32 rd_off = lane_now() * L + lead, rd_len = L;    // the read's bytes, relative to the tile's aligned start
            if constexpr (RAGGED) { rd_off = cur_m.rel; rd_len = cur_m.len; }
            if constexpr (SEG) { rd_off = seg_rel + cur_g.lead; rd_len = L - (u32)((seg_short >> lane_now()) & 1ull); }
            const u32 c0 = rd_off >> 4, c1 = rd_len ? (rd_off + rd_len - 1u) >> 4 : c0;
            bool dirty_read;
            if constexpr (PARK) {
                const u32* const xs = P + (1u + XOFF) + c0;
                u32 x = 0;
#pragma unroll
                for (int j = 0; j <= NW; ++j) {      // (a read of 16 NW bases touches at most NW + 1 chunks)
                    const u32 v = xs[j];             // (past the read's last chunk: whatever lies there -- inside the wave's own area -- is not looked at)
                    x |= c0 + (u32)j <= c1 ? v : 0u;
                }
                dirty_read = chunk_has_invalid(x);
            } else {
                // the round-5 way: the tile is still in w[] (the next tile's rows are not yet asked for) -- its chunks' verdicts again, one
                // ballot per row, the bitmap parked in the plane area (free until phase C); every lane looks up the chunks of its read
                u64* BM = reinterpret_cast<u64*>(PL);
#pragma unroll
                for (int it = 0; it < NW; ++it) {
                    const u32 c = it * 64u + lane;
                    u32 rb = 0;
                    (void)encode16(w[it], rb);
                    const u64 row = __ballot(c < n_chunks && chunk_has_invalid(rb));
                    if (lane == 0) BM[it] = row;
                }
                if (lane == 0) {
                    u32 z = 0;                       // (made here: as a constant the 64-bit zero was kept -- spilled -- across the tile loop for this rare path)
                    asm volatile("" : "+v"(z));
                    reinterpret_cast<u32*>(BM + NW)[0] = z; reinterpret_cast<u32*>(BM + NW)[1] = z;
                    reinterpret_cast<u32*>(BM + NW)[2] = z; reinterpret_cast<u32*>(BM + NW)[3] = z;
                }
                lds_fence();
                const u32 q0 = c0 >> 6, b0 = c0 & 63u;
                const u64 lo = BM[q0], hi = BM[q0 + 1u];
                const u64 bits = b0 ? ((lo >> b0) | (hi << (64u - b0))) : lo;
                dirty_read = (bits & ((1ull << (c1 - c0 + 1u)) - 1ull)) != 0ull;
            }
            const u64 dm = uniform_u64(__ballot(rd_len != 0u && dirty_read));
            if (dm != 0ull) {
                // (buffer stores: the addresses stay on the scalar side.  A flat store's 64-bit address and the constants around it
                // cost the widest variants registers ACROSS the tile loop -- the two-word k on the 10-word frame spilled 24..160 bytes
                // with reloads on the loop's main path, the ticket among them)
                typedef u32 u32x2 __attribute__((ext_vector_type(2)));
                const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(uniform_u64(reinterpret_cast<u64>(masks + tile))), 0, 8, 0x00020000);
                if (lane_now() == 0u) {
                    u32x2 dv = {(u32)dm, (u32)(dm >> 32)};
                    __builtin_amdgcn_raw_buffer_store_b64(dv, rm, 0, 0, 0);
                }
                n_marked += (u32)__builtin_popcountll(dm);
            }
            return true;
        };
        bool bad_tile = false;      // a tile that is not scanned here: it rolls per lane, as a whole
        if constexpr (RAGGED) {
            const bool dirty = cur_m.fits && phase_A();
            bad_tile = !cur_m.fits;
            if constexpr (LATE > 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see the uniform branch
            if (dirty) {     // (before the next tile's rows are asked for: their registers are still free here)
                lds_fence();
                if (!mark_dirty_reads(cur_m.n_ch)) bad_tile = true;   // (no mask array: exact, the slow way)
            }
            __builtin_amdgcn_sched_barrier(0);
            if (next_tile < n_full) {
                meta_finish(nx_m);            // offsets requested a whole iteration ago
                issue_loads_ragged(nx_m, 0, NLD - LATE);
            }
            __builtin_amdgcn_sched_barrier(0);
            ticket_issue();
        } else {
            const bool dirty = phase_A();
            // LATE > 0: the waits for the rows' loads sit under branches of phase A (lanes past the tile skip their chunk), so
            // for hipcc a row may still be in flight afterwards, and the first write to one of its registers -- they are free
            // until the late rows go out -- would cost an s_waitcnt vmcnt(0) with the next tile's loads already out.  Here, on
            // every path, the wait is free: the wave has just used all of them.
            if constexpr (LATE > 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            if constexpr (INLINE) {
                if (dirty) {     // (before the next tile's rows are asked for: their registers are still free here)
                    lds_fence();
                    (void)mark_dirty_reads(SEG ? (cur_g.nbytes + 15u) >> 4 : chunks);
                }
            }
            prefetch(next_tile, tile, 0, NLD - LATE);
            ticket_issue();
        }
        lds_fence();
        if (bad_tile) {   // a ragged tile outside the frame (packed input has no such tiles, uniform ASCII reads never get here)
            if constexpr (RAGGED) fallback_read(tile * 64u + lane, true);
        }
        {
            const bool run = !bad_tile;
            if (run) phase_BC();
            if (run || LATE > 0) phase_D(run);
        }
        tile = next_tile;
        next_tile = uniform_u64(ticket_take());     // requested behind this iteration's phase A: phases B-D ago
        if constexpr (RAGGED) {
            cur_m = nx_m;
            if (next_tile < n_full) meta_issue(next_tile);
        }
        if constexpr (SEG) {
            cur_g = nx_g;
            if (next_tile < n_full) seg_geom(next_tile, nx_g);
        }
    }

    // ---- final partial tile: per-lane rolling.  (Round 6: every read in TAIL runs of windows, a lane per run, over as many waves
    // as that takes -- one wave rolling up to 63 whole reads was the last thing a small batch waited for: 25 of the 41 us of a
    // launch on 1e5 reads, profiles/r06_small_batches.txt.  A run of a 150-base read at k = 31: 45 bases.)
    const u32 rem = (u32)(n_reads & 63u);
    if (rem != 0u) {
        constexpr u32 TAIL = 8;
        const u32 items = rem * TAIL, n_lanes = gridDim.x * 256u;
        for (u32 it0 = (u32)wave_id * 64u; it0 < items; it0 += n_lanes) {     // (wave-uniform bounds)
            const u32 it = it0 + lane_now();
            fallback_read(n_full * 64u + it / TAIL, it < items, it % TAIL, TAIL);
        }
    }

    // ---- combine the bit-sliced counters into word-domain results (once per wave; wave-uniform branch)
    // Number of set bits of canonical bit (t,b) over all k-mers of the wave:
    //   cnt(t,b) = C[t][b] + C[K-1-t][b] + (nk - sum popcount(m)) - Tq[t][b],  Tq[t][b] = sum_o popcount(plane(o+K-1-t, b))
    // (canon bit = m ? fw bit (t,b) : rc bit (t,b) = ~fw bit (K-1-t,b)).  Then
    //   sum of word w of canon = sum_{t in word w, b} 2^(2(t&31)+b) * cnt(t,b)   (wrapping),
    //   xor-fold of the 2-bit-group-reversed hash: bit 2(K-1-t)+b = parity of cnt(t,b),
    //   sum of fw words (K<=32) = sum over planes of popcount total * closed-form per-base weight.
    u64 bs_n = 0, bs_s0 = 0, bs_s1 = 0, bs_x0 = 0, bs_x1 = 0, bs_fw = 0;
    if (n_bs_tiles != 0u) {
        // k-mers handled bit-sliced by this wave
        const u64 nk = RAGGED ? wave_sum(NVR[lane]) : (u64)n_bs_tiles * 64u * (u64)W - n_short;   // (SEG: a short segment holds one window less)
        bs_n = nk;
        u64 fwall = 0;
        u32 tot[NW];
#pragma unroll
        for (int g = 0; g < NW; ++g) {
            const u32 qidx = 32u * g + p;
            const u32 pcq = TOT[64u * g + lane];           // per-plane totals of this half's set
            if constexpr (K <= 32 && !RAGGED) {
                if (sumfw_on) {     // (two 64-bit divisions per plane: only for the caller who asked for the sum)
                    u64 wf, wr;
                    plane_weights(qidx >> 1, L, (u32)K, wf, wr);
                    fwall += (u64)pcq * (wf << (qidx & 1u));
                }
            }
            tot[g] = pcq + __shfl_xor(pcq, 32, WAVE);      // lanes p and p+32 hold the same plane of the two sets
            if constexpr (SEG) {
                // the short segments' last window (W-1) does not exist: base i of it sat at exponent i - (W-1) of the forward word
                const u32 gs = (u32)g - ((W - 1u) >> 4), i = qidx >> 1;
                const u32 pcs = gs < TOTS_G ? TOTS[64u * gs + lane] : 0u;
                if constexpr (K <= 32) {
                    if (i >= W - 1u && i <= L - 1u) fwall -= (u64)pcs * ((1ull << (2u * (i - (W - 1u)))) << (qidx & 1u));
                }
                const u32 both = pcs + __shfl_xor(pcs, 32, WAVE);
                if (half == 0) (PL + PLANES)[32u * g + p] = both;     // (the set-1 plane area is free by now)
            }
        }
        if constexpr (K <= 32) bs_fw = wave_sum(fwall);
        u32 mcnt_c;
        asm volatile("v_mov_b32 %0, %1" : "=&v"(mcnt_c) : "v"(mcnt));
        const u64 mc = wave_sum((u64)mcnt_c);
        fold_acc();
        const u64* const CS = CSA;
#pragma unroll
        for (int g = 0; g < NW; ++g)
            if (half == 0) PL[32u * g + p] = tot[g];        // PL[2*base + bit] = popcount total of that plane
        // ragged: QE[2*i + bit] = how many reads have that bit set in base i of their last K-1 bases (base len-K+1+i)
        u32* QE = VAL;   // (the validity planes are dead by now; the plane area right after the totals holds the counter sums)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const u32 qte = QT[e * 64 + lane];
            const u32 both = qte + __shfl_xor(qte, 32, WAVE);
            if (half == 0) QE[32u * e + p] = both;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        u64 s0 = 0, s1 = 0, x0 = 0, x1 = 0;
        // Tq[t][b] is a window of plane totals that slides one base down as t goes up: Tq[0] = the planes K-1 .. L-1, and
        //   Tq[t+1] - Tq[t] = PL[K-2-t] - PL[L-1-t]          (ragged: PL[K-2-t] - QE[K-2-t], the read ends instead of the frame's end)
        // -- a sum the lanes of a half-wave share and a running sum over them.  (Round 6.  Until then every
        // lane walked its window: up to L-K+1 dependent LDS reads, 8 us at the end of every launch whatever its size -- a quarter of
        // what a batch of 1e5 reads took: profiles/r06_small_batches.txt.)
        // (lane = (bit b, t mod 32): the scan runs within a half-wave, on DPP alone)
        const u32 bb = lane >> 5;
        u64 tq0 = 0;
        for (u32 i = (u32)(K - 1) + (lane & 31u); i + 1u <= L; i += 32u) tq0 += PL[2u * i + bb];
        tq0 = half_sum(tq0);
        u64 carry = 0;                                      // the scan's total over this half-wave, passes before this one
        for (u32 t0 = 0; t0 < (u32)K; t0 += 32u) {
            const u32 t = t0 + (lane & 31u), t2 = (u32)K - 1u - t;
            const bool on = t < (u32)K;
            u64 dlt = 0;
            if (on && t + 2u <= (u32)K) {
                const u32 i = (u32)K - 2u - t;
                dlt = PL[2u * i + bb];
                if constexpr (RAGGED) dlt -= QE[2u * i + bb];
                else dlt -= PL[2u * (L - 1u - t) + bb];
            }
            const u64 inc = half_scan_sum(dlt);
            u64 tq = tq0 + carry + (inc - dlt);
            if constexpr (K > 32) {
                const u32 ilo = (u32)inc, ihi = (u32)(inc >> 32);
                const u64 e0 = ((u64)(u32)__builtin_amdgcn_readlane((int)ihi, 31) << 32) | (u32)__builtin_amdgcn_readlane((int)ilo, 31);
                const u64 e1 = ((u64)(u32)__builtin_amdgcn_readlane((int)ihi, 63) << 32) | (u32)__builtin_amdgcn_readlane((int)ilo, 63);
                carry += bb ? e1 : e0;
            }
            if (on) {
                const u32 tc = t < t2 ? t : t2;
                u64 cc = CS[2u * tc + bb];                      // C[t][b] + C[K-1-t][b]; the middle class holds C[mid][b] once
                if (t == t2) cc += cc;
                if constexpr (SEG) tq -= (PL + PLANES)[2u * (W - 1u + t2) + bb];   // window W-1 of the short segments
                const u64 cnt = cc + (nk - mc) - tq;
                const u32 sh = 2u * (t & 31u) + bb;
                if (t < 32u) s0 += cnt << sh; else s1 += cnt << sh;
                if (want_hash && (cnt & 1ull)) {
                    const u32 hb = 2u * t2 + bb;
                    if (hb < 64u) x0 ^= 1ull << hb; else x1 ^= 1ull << (hb - 64u);
                }
            }
        }
        bs_s0 = wave_sum(s0);
        bs_s1 = wave_sum(s1);
        bs_x0 = wave_xor(x0);
        bs_x1 = wave_xor(x1);
        if constexpr (K <= 32 && RAGGED) {
            // Sum of the forward words of reads of unequal length (round 3).  Base i of a read of len bases sits at exponent e = i - o of
            // the windows o = max(0, i-K+1) .. min(i, len-K): its weight is S(min(i, K-1)) - S(i - (len-K+1)), S(n) = (4^(n+1) - 1) / 3,
            // S(<0) = 0.  The first term needs the plane totals only (a plane counts a base for the reads that have it); the second is
            // non-zero for the last K-1 bases of a read, whose totals by position from the end are QE[].
            if (sumfw_on) {
                u64 f = 0;
                for (u32 pid = lane; pid < 2u * L; pid += 64u) {
                    const u32 i = pid >> 1, e = i < (u32)(K - 1) ? i : (u32)(K - 1);
                    f += (u64)PL[pid] * ((((1ull << (2u * e + 2u)) - 1ull) / 3ull) << (pid & 1u));
                }
                for (u32 pid = lane; pid < 2u * (u32)(K - 1); pid += 64u)
                    f -= (u64)QE[pid] * ((((1ull << (2u * (pid >> 1) + 2u)) - 1ull) / 3ull) << (pid & 1u));
                bs_fw = wave_sum(f);
            }
        }
    }

    // ---- one set of atomics per BLOCK (round 5; until then per wave).  The waves of a batch that is not huge finish within
    // microseconds of each other, and their 3-5 atomics each on the ONE cache line of the summary are then served one after the
    // other at ~6 ns apiece: ~100 us per call for every batch from 1e6 to 2e7 reads (1e6 reads: 205 us per call, 24 us of them
    // the bytes at the scan's rate) -- profiles/r05_small_batches.txt.  The four waves meet once, here, at a block barrier (every
    // wave gets here: the gate above returns for the whole grid or for nobody).
    if constexpr (!FB_FLUSH) {
        bs_n += wave_sum(fb_all.n); bs_s0 += wave_sum(fb_all.s0); bs_s1 += wave_sum(fb_all.s1);
        bs_x0 ^= wave_xor(fb_all.x0); bs_x1 ^= wave_xor(fb_all.x1); bs_fw += wave_sum(fb_all.fw);
    }
    // (The block is FOUR waves -- __launch_bounds__(256), and launch_bs starts nothing else: BLK holds four slots of seven words -- and
    // every early return above this barrier is grid-uniform; a wave that scanned nothing leaves zeros in its slot.)
    u64* const BLK = reinterpret_cast<u64*>(lds + 4u * wave_dw);
    if (lane == 0) {
        u64* const mine = BLK + 7u * wib;
        mine[0] = bs_n; mine[1] = bs_s0; mine[2] = bs_s1; mine[3] = bs_x0; mine[4] = bs_x1; mine[5] = bs_fw; mine[6] = (u64)n_marked;
    }
    __syncthreads();
    if (wib == 0u) {
        u64 v[7];
#pragma unroll
        for (u32 i = 0; i < 7u; ++i) {
            const u64 a = BLK[i], b = BLK[7u + i], c = BLK[14u + i], d = BLK[21u + i];
            v[i] = (i == 3u || i == 4u) ? (a ^ b ^ c ^ d) : (a + b + c + d);
        }
        if (v[0] != 0ull) emit_sums(v[0], v[1], v[2], v[3], v[4], v[5]);   // (no k-mer: nothing to add -- every word is zero then)
        if (v[6] != 0ull && lane == 0) atomicAdd(queue + KMX_Q_MARKED, (unsigned long long)v[6]);
        // ---- the last block to hand in closes the launch (kmx_device.h, "the context's queue block")
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        u32 before = 0;
        if (lane == 0) before = (u32)atomicAdd(queue + KMX_Q_DONE, 1ull);
        before = __builtin_amdgcn_readfirstlane(before);
        if (before + 1u == gridDim.x) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const u32 ln = lane_now();
            u64 part[6];
#pragma unroll
            for (u32 i = 0; i < 6u; ++i) {      // lane s < 16 takes partial summary s -- and leaves zeros
                part[i] = 0;
                if (ln < 16u) part[i] = atomicExch(queue + KMX_Q_SLOTS + ln * 16u + i, 0ull);
            }
            u64 marked = 0;
            if (ln == 0u) marked = atomicExch(queue + KMX_Q_MARKED, 0ull);
            marked = wave_sum(marked);
            const u64 f_n = wave_sum(part[0]), f_s0 = wave_sum(part[1]), f_s1 = wave_sum(part[2]);
            const u64 f_x0 = wave_xor(part[3]), f_x1 = wave_xor(part[4]), f_fw = wave_sum(part[5]);
            if (ln < 32u) queue[ln * 16u] = 0ull;          // the ticket heads
            if (ln == 0u) {
                queue[KMX_Q_DONE] = 0ull;
                queue[KMX_Q_MARKED_OUT] = marked;
                u64 wds[5];
                if constexpr (K <= 32) { wds[0] = f_n; wds[1] = f_s0; wds[2] = f_x0; wds[3] = f_fw; wds[4] = 0; }
                else { wds[0] = f_n; wds[1] = f_s0; wds[2] = f_s1; wds[3] = f_x0; wds[4] = f_x1; }
                constexpr u32 NWD = K <= 32 ? 4u : 5u;     // kmx_summary / kmx_summary2
                constexpr u32 XOR0 = K <= 32 ? 2u : 3u, XOR1 = K <= 32 ? 2u : 4u;
                unsigned long long* const o = static_cast<unsigned long long*>(out);
                if ((want_sumfw & KMX_BS_STORE) != 0u) {
#pragma unroll
                    for (u32 i = 0; i < NWD; ++i) o[i] = wds[i];
                } else {
#pragma unroll
                    for (u32 i = 0; i < NWD; ++i) {
                        if (i == XOR0 || i == XOR1) atomicXor(o + i, (unsigned long long)wds[i]);
                        else atomicAdd(o + i, (unsigned long long)wds[i]);
                    }
                }
                if ((want_sumfw & KMX_BS_PUBLISH) != 0u) {
                    unsigned long long* const host = reinterpret_cast<unsigned long long*>(queue[KMX_Q_HOST]);
                    host[1] = marked;
#pragma unroll
                    for (u32 i = 0; i < 5u; ++i) host[2u + i] = wds[i];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: the words above before the token
                    __hip_atomic_store(host, (unsigned long long)(want_sumfw >> 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ the reads the main pass marked
// masks[t] (behind queue[515]) = the reads of tile t that touch a chunk with an invalid byte, left by the main pass, which
// scanned the tile as it is.  sweep_flagged_kernel (kmx_sweep.hip, round 6) gathers them 64 at a time, finds the windows that
// hold an invalid byte -- exactly those the reference's iterator does not yield (canonical_kmer_iterator.rs:50-66) -- and
// subtracts them.  Every mask goes back to zero: the caller never clears the array.
// Arguments as scan_bitsliced_kernel's; `ragged` / `is_seg` name the variant whose reads these are.
hipError_t launch_sweep_flagged(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 want_hash, u32 want_sumfw, void* out,
                                unsigned long long* queue, int n_cu, hipStream_t stream, const u64* offsets, u32 lead, const u64* ends,
                                const BsSeg& seg, bool ragged, bool is_seg);

// ------------------------------------------------------------------ launcher

template <int K, int NW, int WPL, bool PACKED = false, bool RAGGED = false, bool SEG = false>
static hipError_t launch_bs(const uint8_t* bases, u64 n_reads, u32 L, u32 want_hash, u32 want_sumfw, void* out,
                            unsigned long long* queue, int n_cu, hipStream_t stream, const u64* offsets = nullptr,
                            const u64* ends = nullptr /* RAGGED: the reads' ends (nullptr: offsets + 1) */,
                            BsSeg seg = BsSeg{0, 0, 0, 0, 0} /* SEG: n_reads counts segments, L = the segment frame */) {
    if (offsets != nullptr && ends == nullptr) ends = offsets + 1;
    auto kern = scan_bitsliced_kernel<K, NW, WPL, PACKED, RAGGED, SEG>;
    // uniform ASCII reads from a base that is not 16-byte aligned: the kernel streams from the aligned address below it
    u32 lead = 0;
    if constexpr (!PACKED && !RAGGED && !SEG) {
        lead = (u32)(reinterpret_cast<uintptr_t>(bases) & 15u);
        bases -= lead;
        if (lead != 0u && 4u * L + 1u > 64u * (u32)NW) return hipErrorInvalidValue;   // (callers check: the extra chunk must fit the frame)
    }
    if constexpr (SEG) {   // (a tile is addressed from the buffer's first byte: its own 16-byte alignment is the kernel's business)
        if ((reinterpret_cast<uintptr_t>(bases) & 15u) != 0u || 4u * L + 1u > 64u * (u32)NW) return hipErrorInvalidValue;
    }
    const u32 chunks = 4u * L + ((RAGGED || SEG || lead != 0u) ? 1u : 0u);
    const u32 ldsw = bs_packed_dwords<NW, PACKED, bs_park<K, NW, WPL, PACKED, RAGGED, SEG>()>(chunks, (u32)WPL);
    constexpr u32 NV = RAGGED ? (16 * NW - K + 1 + 31) / 32 : 0;
    constexpr u32 NE = RAGGED ? (K - 1 + 15) / 16 : 0;
    constexpr u32 CSA_DW = 4u * ((K + 1) / 2);
    constexpr u32 TOTS_DW = SEG ? 64u * ((K - 1) / 16 + 2) : 0u;
    // (four waves' areas + the 64 dwords in which they combine their sums at the end)
    const size_t lds_bytes = (size_t)(ldsw + 4u * (u32)bs_plane_dwords(NW) + (RAGGED ? 2u * (32u * NV + 8u) : 0u) + (RAGGED ? 64u * (NE + 2) : 0u) + CSA_DW + TOTS_DW) * 4u * 4u + 256u;
    // blocks per CU, cached per host thread and device (one thread per context / GPU is the ABI's model: a plain static
    // would be shared, and written, by all of them)
    static thread_local int bpc = 0, bpc_dev = -1;
    static thread_local size_t bpc_lds = 0;
    int dev_now = -1;
    (void)hipGetDevice(&dev_now);
    if (bpc == 0 || bpc_lds != lds_bytes || bpc_dev != dev_now) {
        bpc_dev = dev_now;
        int b = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kern, 256, lds_bytes);
        if (e != hipSuccess) return e;
        bpc = b > 0 ? b : 1;
        bpc_lds = lds_bytes;
    }
    if (L < (u32)K || L - (u32)K + 1u > 32u * (u32)WPL) return hipErrorInvalidValue;   // phase D runs ONE round: a half-wave per set, its 32 lanes hold all of a read's windows
    if ((n_reads >> 6) >= (1ull << 36)) return hipErrorInvalidValue;   // tickets are kept as 32-bit values (32 heads x 2^32 tiles; 2^42 reads is far past any HBM)
    const u64 n_tiles = (n_reads + 63u) >> 6;
    u64 grid = (u64)n_cu * (u64)bpc;
    const u64 need = (n_tiles + 3u) / 4u;
    if (grid > need) grid = need;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds_bytes, stream, bases, n_reads, L, want_hash, want_sumfw, out, queue, offsets, lead, ends, seg);
    if constexpr (!PACKED) {
        // the windows with an invalid byte, out again (none on clean input: the waves return at once)
        if ((want_sumfw & KMX_BS_NO_SWEEP) == 0u)
            return launch_sweep_flagged(bases, n_reads, L, (u32)K, want_hash, want_sumfw & KMX_BS_SUMFW, out, queue, n_cu, stream, offsets, lead, ends, seg, RAGGED, SEG);
    }
    return hipGetLastError();
}

// Uniform reads longer than a frame (L > 256), any k from 13 to 64: how they are cut (see BsSeg).  As few segments as the
// largest frame allows, all of T or T - 1 windows.  Single-word k: the 13-word frame (207 bases + the alignment lead, six
// windows per lane: T <= min(192, 208 - k)) -- k - 1 of every T + k - 1 bases are scanned twice, 30 of 158 in the 10-word frame
// and 30 of 207 here (1 000-base reads 0.59 -> 0.64 of the roofline) -- unless the segments that come out fit the 10-word frame
// (reads of 257..~450 bases).  Two-word k: the 10-word frame up to k = 49, the 13-word frame from k = 50 (below).
struct BsSegPlan { u32 J, J1, T, NW; };
static inline BsSegPlan bs_seg_plan(u32 L, u32 k) {
    const u32 wr = L - k + 1u;
    const u32 t10 = 160u - k < 128u ? 160u - k : 128u, t13 = 208u - k < 192u ? 208u - k : 192u;
    // Two-word k (round 5): up to k = 49 -- five accumulator blocks -- the 10-word frame's segment variant fits three waves with nothing in
    // scratch inside the tile loop, and beats the 13-word frame at two waves although it re-reads more (k = 33 on 10 000-base reads
    // 0.64 against 0.55, k = 36 / 37 on 1 000-base reads 0.58 against 0.50 / 0.52, k = 41 on 300-base reads 0.54 against 0.50; on
    // 1 000-base reads k = 41 is a tie, 0.52 / 0.53); from k = 50 the 13-word frame at two waves (k = 63 on 1 000-base reads 0.45 against
    // the 10-word frame's 0.43 at two waves; at three it spills inside the loop) -- profiles/r05_seg2_frames.txt, r05_seg2_3waves.txt.
    const u32 t_max = (k > 32u && k <= 49u) ? t10 : t13;
    const u32 J = (wr + t_max - 1u) / t_max, T = (wr + J - 1u) / J;
    // (round 6) single-word k: segments of up to 176 bases take the 11-word frame -- reads of 257..~320 bases (2 x 300 runs) are two segments of
    // ~165 bases, three quarters of the 13-word frame's 208 (0.64 of the roofline at 300 bases: profiles/r06_len_sweep.txt)
    const u32 nw = T <= t10 ? 10u : (k <= 31u && T + k - 1u <= 176u - 1u && T <= 160u) ? 11u : 13u;
    return BsSegPlan{J, J - (J * T - wr), T, nw};
}
template <int K>
static hipError_t launch_bs_seg(const uint8_t* bases, u64 n_reads, u32 L, u32 want_hash, u32 want_sumfw, void* out,
                                unsigned long long* queue, int n_cu, hipStream_t stream) {
    const BsSegPlan pl = bs_seg_plan(L, (u32)K);
    if (pl.T <= 64u || pl.T > 192u || n_reads > (1ull << 40) / pl.J) return hipErrorInvalidValue;
    const BsSeg seg{L, pl.J, pl.J1, pl.J < 64u ? (u32)(0x100000000ull / pl.J) + 1u : 0u, ~0ull / pl.J + 1ull};   // (J >= 2: floor((2^64 - 1) / J) = floor(2^64 / J) unless J is a power of two, where the + 1 lands on 2^64 / J + 1 as well)
    const u64 n_seg = n_reads * pl.J;
    const u32 Lf = pl.T + (u32)K - 1u;
    if constexpr (K <= 31) {
        if (pl.NW == 11u) return launch_bs<K, 11, 5, false, false, true>(bases, n_seg, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, nullptr, nullptr, seg);
        if (pl.NW == 13u) return launch_bs<K, 13, 6, false, false, true>(bases, n_seg, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, nullptr, nullptr, seg);
    } else if constexpr (K > 49) {
        // Two-word k from 50 up in the 13-word frame (round 5; bs_seg_plan says why the smaller ones stay in the 10-word frame): of a
        // 10-word segment's <= 159 bases k - 1 are shared with the next one -- 97 windows per 159 bases loaded at k = 63 -- and the
        // 13-word frame's 207 bases hold 145 (1.43 instead of 1.64 bytes loaded per byte of input).  T <= 208 - k <= 158: five windows per lane.
        if (pl.NW == 13u) return launch_bs<K, 13, 5, false, false, true>(bases, n_seg, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, nullptr, nullptr, seg);
    }
    if (pl.NW != 10u) return hipErrorInvalidValue;
    if (pl.T <= 96u) return launch_bs<K, 10, 3, false, false, true>(bases, n_seg, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, nullptr, nullptr, seg);
    return launch_bs<K, 10, 4, false, false, true>(bases, n_seg, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, nullptr, nullptr, seg);
}

// One entry point per k (the instantiations are spread over several translation units so that they compile in
// parallel): picks the frame (NW) and the windows per lane (WPL) so that the 2*ceil(W/WPL) (set, group) items of a
// tile fit the 64 lanes in ONE round.
template <int K, bool PACKED>
static hipError_t launch_bs_any(const uint8_t* bases, u64 n_reads, u32 L, u32 want_hash, u32 want_sumfw, void* out,
                                unsigned long long* queue, int n_cu, hipStream_t stream) {
    const u32 W = L - (u32)K + 1u;
    if constexpr (!PACKED) {
        if (L > 256u) return launch_bs_seg<K>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
        // up to 112 bp (the 100 / 101 / 75 / 76 / 50 / 36 bp of older runs): the 7-word frame -- 7 instead of 10 transposes per
        // half-wave, 28 instead of 40 prefetch registers, two thirds of the plane area (a read's extra chunk from an unaligned base must fit too)
        const u32 mis = (reinterpret_cast<uintptr_t>(bases) & 15u) ? 1u : 0u;
        // up to 80 bp (round 3; the 75 / 76 / 50 / 36 bp of older runs): the 5-word frame -- 5 transposes, 20 prefetch registers
        if (4u * L + mis <= 64u * 5u) {
            // (up to 32 windows -- 50 bp at k = 31, 36 bp at any k: one window per lane; two would leave half of phase D's lanes idle and double its passes)
            if (W <= 32u) return launch_bs<K, 5, 1, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
            if (W <= 64u) return launch_bs<K, 5, 2, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
            return launch_bs<K, 5, 3, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
        }
        if (4u * L + mis <= 64u * 7u) {
            if (W <= 64u) return launch_bs<K, 7, 2, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
            if (W <= 96u) return launch_bs<K, 7, 3, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
            return launch_bs<K, 7, 4, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
        }
        // 113..128 bp (round 3; the 125 / 126 bp of HiSeq runs): the 8-word frame instead of the 10-word one
        if (4u * L + mis <= 64u * 8u) {
            if (W <= 96u) return launch_bs<K, 8, 3, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
            return launch_bs<K, 8, 4, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
        }
    }
    if (L > 160) {   // 161..256 bp: the 16-word frame, as few windows per lane as keep the 2*ceil(W/WPL) items inside 64 lanes
        if constexpr (!PACKED) {
            // 161..208 bp (round 3): the 13-word frame -- 13 instead of 16 transposes per half-wave, 52 instead of 64 prefetch
            // registers (4 waves/SIMD), four fifths of the plane area
            const u32 mis13 = (reinterpret_cast<uintptr_t>(bases) & 15u) ? 1u : 0u;
            // 161..176 bp (round 6): the 11-word frame -- reads of 161 bases filled 77 % of the 13-word frame's 208
            // (0.70 of the roofline between 0.77 at 150 and 0.76 at 200 bases: profiles/r05_len_sweep.txt)
            if (4u * L + mis13 <= 64u * 11u && W <= 160u)
                return launch_bs<K, 11, 5, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
            if (4u * L + mis13 <= 64u * 13u) {
                if (W <= 160u) return launch_bs<K, 13, 5, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
                if (W <= 192u) return launch_bs<K, 13, 6, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
                return launch_bs<K, 13, 7, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
            }
            if (W <= 160u) return launch_bs<K, 16, 5, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
            if (W <= 192u) return launch_bs<K, 16, 6, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
            if (W <= 224u) return launch_bs<K, 16, 7, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
        }
        return launch_bs<K, 16, 8, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
    }
    if (W <= 96u) return launch_bs<K, 10, 3, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
    if (W <= 128u) return launch_bs<K, 10, 4, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
    return launch_bs<K, 10, 5, PACKED>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);
}

// [u64;2] k-mers (k in 33..64): WPL = ceil(W / 32) on the 10-word frame; reads of 161..208 / 209..256 bases (round 4) take the
// 13- / 16-word frame with ONE instantiation each -- 6 / 7 windows per lane hold every W a two-word k leaves there (<= 176 / 224),
// the window blocks past W are skipped at run time
template <int K>
static hipError_t launch_bs2_any(const uint8_t* bases, u64 n_reads, u32 L, u32 want_hash, void* out, unsigned long long* queue,
                                 int n_cu, hipStream_t stream) {
    const u32 W = L - (u32)K + 1u;
    if (L > 256u) return launch_bs_seg<K>(bases, n_reads, L, want_hash, 0, out, queue, n_cu, stream);
    if (L > 160) {
        const u32 mis = (reinterpret_cast<uintptr_t>(bases) & 15u) ? 1u : 0u;
        if (4u * L + mis <= 64u * 13u) return launch_bs<K, 13, 6>(bases, n_reads, L, want_hash, 0, out, queue, n_cu, stream);
        return launch_bs<K, 16, 7>(bases, n_reads, L, want_hash, 0, out, queue, n_cu, stream);
    }
    if (W <= 64u) return launch_bs<K, 10, 2>(bases, n_reads, L, want_hash, 0, out, queue, n_cu, stream);
    if (W <= 96u) return launch_bs<K, 10, 3>(bases, n_reads, L, want_hash, 0, out, queue, n_cu, stream);
    return launch_bs<K, 10, 4>(bases, n_reads, L, want_hash, 0, out, queue, n_cu, stream);
}
#define KMX_BS2_DECLARE_K(K) \
    hipError_t launch_bs2_k##K(const uint8_t* bases, u64 n_reads, u32 L, u32 want_hash, void* out, unsigned long long* queue, \
                               int n_cu, hipStream_t stream);
#define KMX_BS2_DEFINE_K(K)                                                                                                 \
    hipError_t launch_bs2_k##K(const uint8_t* bases, u64 n_reads, u32 L, u32 want_hash, void* out, unsigned long long* queue, \
                               int n_cu, hipStream_t stream) {                                                               \
        return launch_bs2_any<K>(bases, n_reads, L, want_hash, out, queue, n_cu, stream);                                    \
    }
// two-word k with a bit-sliced kernel: every k from 33 to 64
#define KMX_BS2_FOR_EACH_K(X) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) \
    X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) X(64)

// ragged reads: Lf = the frame (longest read a tile may hold; a tile with a longer read rolls per lane)
template <int K>
static hipError_t launch_bs_ragged_any(const uint8_t* bases, const u64* offsets, u64 n_reads, u32 Lf, u32 want_hash, void* out,
                                       unsigned long long* queue, int n_cu, hipStream_t stream, u32 want_sumfw, const u64* ends) {
    const u32 W = Lf - (u32)K + 1u;
    if (Lf > 160) {
        if (W <= 160u) return launch_bs<K, 16, 5, false, true>(bases, n_reads, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, offsets, ends);
        if (W <= 192u) return launch_bs<K, 16, 6, false, true>(bases, n_reads, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, offsets, ends);
        if (W <= 224u) return launch_bs<K, 16, 7, false, true>(bases, n_reads, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, offsets, ends);
        return launch_bs<K, 16, 8, false, true>(bases, n_reads, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, offsets, ends);
    }
    if (Lf <= 111u && Lf >= (u32)K) {   // short reads (a tile of 64 spans at most 448 chunks): the 7-word frame
        if (W <= 96u) return launch_bs<K, 7, 3, false, true>(bases, n_reads, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, offsets, ends);
        return launch_bs<K, 7, 4, false, true>(bases, n_reads, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, offsets, ends);
    }
    if (W <= 96u) return launch_bs<K, 10, 3, false, true>(bases, n_reads, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, offsets, ends);
    if (W <= 128u) return launch_bs<K, 10, 4, false, true>(bases, n_reads, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, offsets, ends);
    return launch_bs<K, 10, 5, false, true>(bases, n_reads, Lf, want_hash, want_sumfw, out, queue, n_cu, stream, offsets, ends);
}
#define KMX_BSR_DECLARE_K(K) \
    hipError_t launch_bs_ragged_k##K(const uint8_t* bases, const u64* offsets, u64 n_reads, u32 Lf, u32 want_hash, void* out, \
                                     unsigned long long* queue, int n_cu, hipStream_t stream, u32 want_sumfw, const u64* ends);
#define KMX_BSR_DEFINE_K(K)                                                                                                  \
    hipError_t launch_bs_ragged_k##K(const uint8_t* bases, const u64* offsets, u64 n_reads, u32 Lf, u32 want_hash, void* out, \
                                     unsigned long long* queue, int n_cu, hipStream_t stream, u32 want_sumfw, const u64* ends) { \
        return launch_bs_ragged_any<K>(bases, offsets, n_reads, Lf, want_hash, out, queue, n_cu, stream, want_sumfw, ends);  \
    }
// k with a bit-sliced kernel for ragged reads: 9..31, like the uniform kernel
#define KMX_BSR_FOR_EACH_K(X) \
    X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
// ... and two-word k (33..64; round 4): ONE instantiation each, the 10-word frame at four windows per lane -- a two-word k leaves
// at most 128 windows in 160 bases; the window blocks past W are skipped at run time
#define KMX_BSR2_DECLARE_K(K) \
    hipError_t launch_bs2_ragged_k##K(const uint8_t* bases, const u64* offsets, u64 n_reads, u32 Lf, u32 want_hash, void* out, \
                                      unsigned long long* queue, int n_cu, hipStream_t stream, const u64* ends);
#define KMX_BSR2_DEFINE_K(K)                                                                                                  \
    hipError_t launch_bs2_ragged_k##K(const uint8_t* bases, const u64* offsets, u64 n_reads, u32 Lf, u32 want_hash, void* out, \
                                      unsigned long long* queue, int n_cu, hipStream_t stream, const u64* ends) {              \
        return launch_bs<K, 10, 4, false, true>(bases, n_reads, Lf, want_hash, 0, out, queue, n_cu, stream, offsets, ends);    \
    }

#define KMX_BS_DECLARE_K(K) \
    hipError_t launch_bs_k##K(const uint8_t* bases, u64 n_reads, u32 L, bool packed, u32 want_hash, u32 want_sumfw, void* out, \
                              unsigned long long* queue, int n_cu, hipStream_t stream);
#define KMX_BS_DEFINE_K(K, WITH_PACKED)                                                                                       \
    hipError_t launch_bs_k##K(const uint8_t* bases, u64 n_reads, u32 L, bool packed, u32 want_hash, u32 want_sumfw, void* out, \
                              unsigned long long* queue, int n_cu, hipStream_t stream) {                                      \
        if constexpr (WITH_PACKED) {                                                                                          \
            if (packed) return launch_bs_any<K, true>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);    \
        } else if (packed) {                                                                                                  \
            return hipErrorInvalidValue;                                                                                      \
        }                                                                                                                     \
        return launch_bs_any<K, false>(bases, n_reads, L, want_hash, want_sumfw, out, queue, n_cu, stream);                   \
    }
// k with a bit-sliced kernel (u64 k-mers), ASCII and packed (SeqVector) input: 13..31
#define KMX_BS_FOR_EACH_K(X) \
    X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
KMX_BS_FOR_EACH_K(KMX_BS_DECLARE_K)
KMX_BS2_FOR_EACH_K(KMX_BS2_DECLARE_K)
KMX_BSR_FOR_EACH_K(KMX_BSR_DECLARE_K)
KMX_BS2_FOR_EACH_K(KMX_BSR2_DECLARE_K)

}  // namespace kmx
