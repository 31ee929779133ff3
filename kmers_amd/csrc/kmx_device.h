// kmx_device.h -- gfx950 device helpers shared by the kmx kernels.
// Pure integer bit manipulation; wave64; no MFMA (the path is HBM/VALU-bound byte work).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kmx.h"

namespace kmx {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int WAVE = 64;

// ({hi,lo} >> sh[4:0])[31:0]  -> v_alignbit_b32
__device__ __forceinline__ u32 alignbit(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// ASCII -> 2-bit.  (c>>1)&3 gives the "internal" code A0 C1 T2 G3 for both cases
// (reference src/encoding/naive.rs:14-16); naive_impl codes are A0 C1 G2 T3
// (src/naive_impl/mod.rs:20-24), obtained per 2-bit group as i ^ (i>>1).
//
// encode16: 16 ASCII bytes (one dwordx4) -> 16 bases packed in one dword, first base lowest,
// in naive_impl (ACGT) codes, plus `bad`: OR of (byte ^ expected upper-case letter) over the 16
// bytes -- the chunk is all-ACGTacgt  <=>  (bad & 0xDFDFDFDF) == 0  (exact, case-insensitive,
// same accept set as encode_binary_u8, src/naive_impl/mod.rs:40-50).
// a 32-bit constant materialised in a VGPR (pure, so it is hoisted out of loops and shared)
template <u32 C>
__device__ __forceinline__ u32 vgpr_const() {
    u32 r;
    asm("v_mov_b32 %0, %1" : "=v"(r) : "i"(C));
    return r;
}

__device__ __forceinline__ u32 encode16(uint4 w, u32& bad) {
    // expected letter by internal code*2 as v_perm selector: 0->'A' 2->'C' 4->'T' 6->'G'
    constexpr u32 TBL_LO = 0x00430041u;  // bytes 0..3 : 'A', -, 'C', -
    constexpr u32 TBL_HI = 0x00470054u;  // bytes 4..7 : 'T', -, 'G', -
    constexpr u32 W4 = 0x40100401u;      // dot4 weights 1,4,16,64
    const u32 t0 = w.x & 0x06060606u, t1 = w.y & 0x06060606u, t2 = w.z & 0x06060606u, t3 = w.w & 0x06060606u;
    const u32 x0 = w.x ^ __builtin_amdgcn_perm(TBL_HI, TBL_LO, t0);
    const u32 x1 = w.y ^ __builtin_amdgcn_perm(TBL_HI, TBL_LO, t1);
    const u32 x2 = w.z ^ __builtin_amdgcn_perm(TBL_HI, TBL_LO, t2);
    const u32 x3 = w.w ^ __builtin_amdgcn_perm(TBL_HI, TBL_LO, t3);
    bad = __builtin_amdgcn_bitop3_b32(bad, x0, x1, 0xFE);   // 3-input OR at full rate (v_or3_b32 is half rate)
    bad = __builtin_amdgcn_bitop3_b32(bad, x2, x3, 0xFE);
    // v_dot4_u32_u8: sum of (2*code_i) * 4^i  = 2 * (4 bases packed in 8 bits)
    const u32 d0 = __builtin_amdgcn_udot4(t0, W4, 0u, false);
    const u32 d1 = __builtin_amdgcn_udot4(t1, W4, 0u, false);
    const u32 d2 = __builtin_amdgcn_udot4(t2, W4, 0u, false);
    const u32 d3 = __builtin_amdgcn_udot4(t3, W4, 0u, false);
    u32 p = (d1 << 8) | d0;          // v_lshl_or_b32 chain: 2 half-rate ops instead of lshl, lshl, or3
    p = (d2 << 16) | p;
    p = (d3 << 23) | (p >> 1);
    // internal (ACTG) -> naive_impl (ACGT) codes: p ^ ((p >> 1) & 0x55555555) as one v_bitop3_b32 whose
    // constant sits in a VGPR (an SGPR source would halve the issue rate)
    return __builtin_amdgcn_bitop3_b32(p >> 1, p, vgpr_const<0x55555555u>(), 0x6c);
}

__device__ __forceinline__ bool chunk_has_invalid(u32 bad) { return (bad & 0xDFDFDFDFu) != 0u; }

// encode16 (kmx_device.h) + one bit per byte: "not one of ACGTacgt" (the accept set of encode_binary_u8, mod.rs:40-50)
__device__ __forceinline__ u32 encode16_inv(const uint4 w, u32& inv16) {
    constexpr u32 TBL_LO = 0x00430041u, TBL_HI = 0x00470054u, W4 = 0x40100401u;   // as in encode16
    const u32 t0 = w.x & 0x06060606u, t1 = w.y & 0x06060606u, t2 = w.z & 0x06060606u, t3 = w.w & 0x06060606u;
    const u32 x0 = w.x ^ __builtin_amdgcn_perm(TBL_HI, TBL_LO, t0);
    const u32 x1 = w.y ^ __builtin_amdgcn_perm(TBL_HI, TBL_LO, t1);
    const u32 x2 = w.z ^ __builtin_amdgcn_perm(TBL_HI, TBL_LO, t2);
    const u32 x3 = w.w ^ __builtin_amdgcn_perm(TBL_HI, TBL_LO, t3);
    // a byte of x is 0x00 or 0x20 for a valid letter: bit 7 of ((x & 0x5F) + 0x7F) | x is set <=> the byte is anything else
    auto nz = [](u32 x) { return __builtin_amdgcn_bitop3_b32((x & 0x5F5F5F5Fu) + 0x7F7F7F7Fu, x, 0x80808080u, 0xA8 /* (a | b) & c */); };
    // v_dot4_u32_u8 gathers the four marks of a dword: 0x80 * (b0 + 2 b1 + 4 b2 + 8 b3), the second dword on top at 16 .. 128
    const u32 a = __builtin_amdgcn_udot4(nz(x1), 0x80402010u, __builtin_amdgcn_udot4(nz(x0), 0x08040201u, 0u, false), false);
    const u32 b = __builtin_amdgcn_udot4(nz(x3), 0x80402010u, __builtin_amdgcn_udot4(nz(x2), 0x08040201u, 0u, false), false);
    inv16 = (a >> 7) | (b << 1);
    const u32 d0 = __builtin_amdgcn_udot4(t0, W4, 0u, false);
    const u32 d1 = __builtin_amdgcn_udot4(t1, W4, 0u, false);
    const u32 d2 = __builtin_amdgcn_udot4(t2, W4, 0u, false);
    const u32 d3 = __builtin_amdgcn_udot4(t3, W4, 0u, false);
    u32 p = (d1 << 8) | d0;
    p = (d2 << 16) | p;
    p = (d3 << 23) | (p >> 1);
    return __builtin_amdgcn_bitop3_b32(p >> 1, p, vgpr_const<0x55555555u>(), 0x6c);   // internal (ACTG) -> naive_impl (ACGT) codes
}

// does the read s[0, len) hold a byte outside ACGTacgt?  (same accept set as encode16.)  All the 16-byte chunks of the
// aligned span around the read -- at most MAXV, the caller guarantees that span is readable, as it is inside a tile of the
// scan kernels -- are requested before the first is looked at: one memory round trip, not one per chunk.
template <int MAXV>
__device__ __forceinline__ bool read_has_invalid(const uint8_t* __restrict__ s, u32 len) {
    constexpr u32 TBL_LO = 0x00430041u, TBL_HI = 0x00470054u;   // as in encode16
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(s), a1 = a0 + len, base = a0 & ~(uintptr_t)15;
    const u32 nch = (u32)((a1 - base + 15u) >> 4);
    if (len == 0u) return false;
    uint4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) v[i] = *reinterpret_cast<const uint4*>(base + 16u * ((u32)i < nch ? (u32)i : nch - 1u));
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const u32 wv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uintptr_t a = base + 16u * i + 4u * j;
            u32 m = ((u32)i < nch && a < a1 && a + 4 > a0) ? ~0u : 0u;
            if (a < a0) m &= ~0u << (8u * (u32)(a0 - a));            // bytes before the read (1..3 here: a + 4 > a0)
            if (a + 4 > a1 && a < a1) m &= ~0u >> (8u * (u32)(a + 4 - a1));   // bytes after it
            acc |= (wv[j] ^ __builtin_amdgcn_perm(TBL_HI, TBL_LO, wv[j] & 0x06060606u)) & 0xDFDFDFDFu & m;
        }
    }
    return acc != 0u;
}

// reverse the 16 2-bit groups of a dword (v_bfrev_b32 + swap the two bits of each pair)
__device__ __forceinline__ u32 revgroups32(u32 x) {
    const u32 y = __builtin_bitreverse32(x);
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}

// reverse the 32 2-bit groups of a u64
__device__ __forceinline__ u64 revgroups64(u64 x) {
    return ((u64)revgroups32((u32)x) << 32) | (u64)revgroups32((u32)(x >> 32));
}

// Kmer::to_reverse_complement (src/naive_impl/kmer.rs:124-136), k in [1,32]
__device__ __forceinline__ u64 revcomp_word(u64 w, u32 k) { return revgroups64(~w) >> (2u * (32u - k)); }

// LexHasher::write_u64 (src/naive_impl/hash.rs:60-71), hasher_k in [1,32]
__device__ __forceinline__ u64 lex_hash(u64 w, u32 hk) { return revgroups64(w) >> (2u * (32u - hk)); }

// MASK_TABLE[k] for k in [0,31] (src/naive_impl/kmer.rs:584-616)
__device__ __host__ __forceinline__ u64 mask2k(u32 k) { return k >= 32 ? ~0ull : ((1ull << (2u * k)) - 1ull); }

// encode_binary_u8 (src/naive_impl/mod.rs:40-50): code 0..3, or 4 for "invalid" (reference: u64::MAX)
__device__ __forceinline__ u32 encode_base(u32 c) {
    const u32 u = c & 0xDFu;
    const u32 i = (c >> 1) & 3u;          // A0 C1 T2 G3
    const u32 expect = (0x47544341u >> (8u * i)) & 0xFFu;  // 'A','C','T','G'
    return (u == expect) ? (i ^ (i >> 1)) : 4u;
}

__device__ __forceinline__ u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// BUILD-DEFINED bucket of a hash value (include/kmx.h, oracle kmo_bucket_of): the top log2_buckets bits of the 32-bit sum of the
// two halves, each times its own odd constant.  One v_mul_lo_u32 and one multiply-add: the partition pass of the histogram
// is bound by VALU issue, and the 64-bit Fibonacci product of round 1 (top bits of h * 0x9E3779B97F4A7C15) cost it 6
// instructions per k-mer.  Every bit of h reaches the top bits of the sum (a multiplicative hash carries upwards only).
__device__ __forceinline__ u32 bucket_mix(u32 lo, u32 hi) { return lo * 0x9E3779B1u + hi * 0x85EBCA6Bu; }
__device__ __forceinline__ u64 bucket_of(u64 h, u32 log2_buckets) {
    return log2_buckets ? (u64)(bucket_mix((u32)h, (u32)(h >> 32)) >> (32u - log2_buckets)) : 0ull;
}

// max over the 64 lanes, returned wave-uniform: DPP inside the 16-lane rows (no LDS round trips), then 4 v_readlane
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
    auto mx = [](u32 a, u32 b) { return a > b ? a : b; };
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true));
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true));
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x124 /* row_ror:4 */, 0xF, 0xF, true));
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128 /* row_ror:8 */, 0xF, 0xF, true));
    const u32 a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const u32 c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return mx(mx(a, b), mx(c, d));
}

// Wave-wide sum / xor of a 64-bit value, the same in every lane (scalar).  Within a row of 16 lanes through DPP (no LDS round trip),
// the four rows through v_readlane.  (Round 6: six ds_bpermute pairs each, one behind the other, used to make up most of a scan's
// epilogue -- a dozen reductions, ~0.3 us apiece: profiles/r06_small_batches.txt.)  Every caller is at a wave-uniform point.
#define KMX_DPP64(v, ctrl)                                                                                         \
    (((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)((v) >> 32), (ctrl), 0xF, 0xF, true) << 32) |          \
     (u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ u64 wave_lanes_0_16_32_48(u64 v, bool add) {
    const u32 lo = (u32)v, hi = (u32)(v >> 32);
    u64 r = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const u64 x = ((u64)(u32)__builtin_amdgcn_readlane((int)hi, l) << 32) | (u32)__builtin_amdgcn_readlane((int)lo, l);
        r = add ? r + x : r ^ x;
    }
    return r;
}
__device__ __forceinline__ u64 wave_sum(u64 v) {
    v += KMX_DPP64(v, 0xB1 /* quad_perm:[1,0,3,2] */);
    v += KMX_DPP64(v, 0x4E /* quad_perm:[2,3,0,1] */);
    v += KMX_DPP64(v, 0x124 /* row_ror:4 */);
    v += KMX_DPP64(v, 0x128 /* row_ror:8 */);
    return wave_lanes_0_16_32_48(v, true);
}
__device__ __forceinline__ u64 wave_xor(u64 v) {
    v ^= KMX_DPP64(v, 0xB1);
    v ^= KMX_DPP64(v, 0x4E);
    v ^= KMX_DPP64(v, 0x124);
    v ^= KMX_DPP64(v, 0x128);
    return wave_lanes_0_16_32_48(v, false);
}

// Inclusive running sum of a 64-bit value within each half-wave (lanes 0..31 and 32..63 apart), and the total of the lane's
// half-wave in each of its lanes: DPP again (row_shr, then lane 15 of rows 0 / 2 broadcast into rows 1 / 3).
__device__ __forceinline__ u64 half_scan_sum(u64 v) {
    v += KMX_DPP64(v, 0x111 /* row_shr:1 */);
    v += KMX_DPP64(v, 0x112 /* row_shr:2 */);
    v += KMX_DPP64(v, 0x114 /* row_shr:4 */);
    v += KMX_DPP64(v, 0x118 /* row_shr:8 */);
    const u64 prev_row = ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x142 /* row_bcast:15 */, 0xA, 0xF, false) << 32) |
                         (u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x142, 0xA /* rows 1 and 3 */, 0xF, false);
    return v + prev_row;
}
__device__ __forceinline__ u64 half_sum(u64 v) {
    v += KMX_DPP64(v, 0xB1);
    v += KMX_DPP64(v, 0x4E);
    v += KMX_DPP64(v, 0x124);
    v += KMX_DPP64(v, 0x128);
    const u32 lo = (u32)v, hi = (u32)(v >> 32);
    auto at = [&](int l) { return ((u64)(u32)__builtin_amdgcn_readlane((int)hi, l) << 32) | (u32)__builtin_amdgcn_readlane((int)lo, l); };
    const u64 h0 = at(0) + at(16), h1 = at(32) + at(48);
    return (threadIdx.x & 32u) ? h1 : h0;
}

// A read of 2^31 bases or more (the iterator's positions are i32, canonical_kmer_iterator.rs:15; kmx.h "Limits") is not
// scanned: the kernels skip it and raise the context's sticky flag, which kmx_ctx_synchronize reports as KMX_E_ARG.
// `queue` = the tile-queue block of the context (d_scratch + 16); the flag lives 8 words below it (kmx_internal.h).
constexpr int KMX_TOOLONG_FROM_QUEUE = -8;
__device__ __forceinline__ bool read_too_long(u64 len, unsigned long long* flag) {
    if (len < (1ull << 31)) return false;
    if (flag) *flag = 1ull;   // (a plain store of one constant: every writer agrees)
    return true;
}

// ---- the context's queue block (`queue` = d_scratch + 16, u64 words; kmx_internal.h), as the bit-sliced scan uses it (round 6):
//   [q * 16], q < 32   the ticket heads, 128 bytes apart
//   [512]              reads marked so far (running count of the launch);  [513] the uniform / ragged gate;  [515] the mask array
//   [516]              the marked reads of the LAST bit-sliced launch, for its sweep (overwritten, never cleared)
//   [517]              the context's pinned host words as the device sees them (written once, kmx_ctx_create)
//   [544..559]         a quiet line (stand-in source of loads that must not fault)
//   [560]              blocks of the launch that have handed in their sums
//   [576 + 16 s + i]   partial summary s (s < 16: block b adds into s = b & 15), word i < 6
// The scan CLOSES its own launch: the last block to hand in adds the sixteen partial summaries up, writes the result, and puts
// the heads, [512], [560] and the partials back to zero -- so a caller that knows only such launches ran since its last clear
// need not clear again (two fill kernels and their gaps: 11 us of a small batch's 70, profiles/r06_small_batches.txt).
constexpr u32 KMX_Q_MARKED = 512, KMX_Q_MARKED_OUT = 516, KMX_Q_HOST = 517, KMX_Q_DONE = 560, KMX_Q_SLOTS = 576;
// `want_sumfw` of scan_bitsliced_kernel / launch_bs carries the launch's mode: bit 0 = the sum of the forward words is wanted;
// STORE = the last block stores the summary (the caller did not zero `out`; default: adds to it); PUBLISH = ... and leaves
// {token, marked reads, the summary's words} in the pinned host words, the token (bits 8..31) last; NO_SWEEP = launch_bs does
// not enqueue the sweep (the caller does, when it has seen the count)
constexpr u32 KMX_BS_SUMFW = 1u, KMX_BS_STORE = 2u, KMX_BS_PUBLISH = 4u, KMX_BS_NO_SWEEP = 8u;

// accumulators of one reduce pass, per lane
struct Acc {
    u64 n_valid = 0, sum_canon = 0, xor_hash = 0, sum_fw = 0;
};

// one set of 64-bit atomics per wave; want_hash / want_sumfw are wave-uniform
__device__ __forceinline__ void flush_acc(const Acc& a, kmx_summary* out, bool want_hash, bool want_sumfw) {
    const u64 n = wave_sum(a.n_valid);
    const u64 s = wave_sum(a.sum_canon);
    u64 x = 0, f = 0;
    if (want_hash) x = wave_xor(a.xor_hash);
    if (want_sumfw) f = wave_sum(a.sum_fw);
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        atomicAdd((unsigned long long*)&out->n_valid, (unsigned long long)n);
        atomicAdd((unsigned long long*)&out->sum_canon, (unsigned long long)s);
        if (want_hash) atomicXor((unsigned long long*)&out->xor_hash, (unsigned long long)x);
        if (want_sumfw) atomicAdd((unsigned long long*)&out->sum_fw, (unsigned long long)f);
    }
}

// ---------------------------------------------------------------------------
// Reference-shaped per-lane rolling over one read (generic path: ragged reads, reads
// with invalid bytes, partial tiles, any k in [1,31]).  Direct restatement of
// CanonicalKmerIterator::find_next (src/naive_impl/canonical_kmer_iterator.rs:42-70) with
// CanonicalKmer::append_base (canonical_kmer.rs:90-94): one lane walks one read.
// `emit(pos, fw, rc)` is called for every yielded window, in increasing pos.
template <bool AHEAD = true, typename Emit>
__device__ __forceinline__ void roll_read(const uint8_t* __restrict__ s, u32 len, u32 k, Emit&& emit) {
    const u64 mask = mask2k(k);
    const u32 top = 2u * k - 2u;
    u64 fw = 0, rc = ~0ull;  // CanonicalKmer::blank_of_size (canonical_kmer.rs:22-29)
    int last_invalid = -1;
    auto step = [&](u32 c, u32 l) {
        const u32 b = encode_base(c);
        if (b < 4u) {
            fw = (fw >> 2) | ((u64)b << top);                  // kmer.rs:98-102
            rc = mask & ((rc << 2) | (u64)(3u - b));           // kmer.rs:91-95, mod.rs:81-84
            if ((int)l - last_invalid >= (int)k) emit(l + 1u - k, fw, rc);
        } else {
            last_invalid = (int)l;
        }
    };
    // 8 bases per (unaligned) global_load_dwordx2, the loads TWO groups ahead of the walk (round 5): a lane's walk is a chain of dependent
    // steps, and with the load issued where its bytes were wanted every group of 8 bases began with a full memory round trip -- the rolled
    // reads of a dirty batch cost what their latencies add up to, not what the walk computes (profiles/r05_dirty_bench.txt)
    // (AHEAD = false: the rare per-lane path INSIDE the tiled scan, where four more registers across the walk push the widest variants into
    // scratch: there the load sits where its bytes are wanted, as it always did)
    u32 l = 0;
    if constexpr (AHEAD) {
        u64 v1 = 0, v2 = 0;
        if (8u <= len) __builtin_memcpy(&v1, s, 8);
        if (16u <= len) __builtin_memcpy(&v2, s + 8u, 8);
        for (; l + 8u <= len; l += 8u) {
            const u64 v = v1;
            v1 = v2;
            if (l + 24u <= len) __builtin_memcpy(&v2, s + l + 16u, 8);
#pragma unroll
            for (u32 j = 0; j < 8u; ++j) step((u32)(v >> (8u * j)) & 0xFFu, l + j);
        }
    } else {
        for (; l + 8u <= len; l += 8u) {
            u64 v;
            __builtin_memcpy(&v, s + l, 8);
#pragma unroll
            for (u32 j = 0; j < 8u; ++j) step((u32)(v >> (8u * j)) & 0xFFu, l + j);
        }
    }
    for (; l < len; ++l) step(s[l], l);
}

// The same walk for a whole wave in lockstep: every lane rolls its own read, and `block(wb)` runs with the wave converged
// each time the 16 windows [16 wb, 16 wb + 16) are complete in every lane (after base k-1 + 16 (wb+1)); `maxlen` = the
// longest read of the wave.  The partitioned histogram drains its LDS rings there (without that a rolled tile overflows
// them and falls back to global atomics); materialise mode writes the block out as it does on its fast path.
template <typename Emit, typename Block>
__device__ __forceinline__ void roll_read_stepped(const uint8_t* __restrict__ s, u32 len, u32 maxlen, u32 k, Emit&& emit, Block&& block) {
    const u64 mask = mask2k(k);
    const u32 top = 2u * k - 2u;
    u64 fw = 0, rc = ~0ull;
    int last_invalid = -1;
    u32 l = 0;
    for (u32 wb = 0;; ++wb) {
        const u32 lend = k - 1u + 16u * (wb + 1u);
        const u32 stop = lend < len ? lend : len;
        auto step = [&](u32 c, u32 at) {
            const u32 b = encode_base(c);
            if (b < 4u) {
                fw = (fw >> 2) | ((u64)b << top);
                rc = mask & ((rc << 2) | (u64)(3u - b));
                if ((int)at - last_invalid >= (int)k) emit(at + 1u - k, fw, rc);
            } else {
                last_invalid = (int)at;
            }
        };
        for (; l + 8u <= stop; l += 8u) {   // 8 bases per (unaligned) global_load_dwordx2
            u64 v;
            __builtin_memcpy(&v, s + l, 8);
#pragma unroll
            for (u32 j = 0; j < 8u; ++j) step((u32)(v >> (8u * j)) & 0xFFu, l + j);
        }
        for (; l < stop; ++l) step(s[l], l);
        block(wb);
        if (lend >= maxlen) break;
    }
}

// The same walk over a read held in a SeqVector (seq_vector.rs: base i at flat bits [2i,2i+1] of `words`): read =
// bases [first, first+len).  Every 2-bit code is a valid base, so every window is yielded.
template <typename Emit>
__device__ __forceinline__ void roll_read_packed(const u64* __restrict__ words, u64 first, u32 len, u32 k, Emit&& emit) {
    const u64 mask = mask2k(k);
    const u32 top = 2u * k - 2u;
    u64 fw = 0, rc = ~0ull;
    u64 wi = first >> 5;
    u32 sh = 2u * (u32)(first & 31u);
    u64 cur = len ? words[wi] : 0ull;
    for (u32 l = 0; l < len; ++l) {
        const u32 b = (u32)(cur >> sh) & 3u;
        fw = (fw >> 2) | ((u64)b << top);
        rc = mask & ((rc << 2) | (u64)(3u - b));
        if (l + 1u >= k) emit(l + 1u - k, fw, rc);
        sh += 2u;
        if (sh == 64u && l + 1u < len) {
            sh = 0;
            cur = words[++wi];
        }
    }
}

// ---------------------------------------------------------------- [u64;2] k-mers (k in 33..64), BUILD-DEFINED
struct U128 {
    u64 lo, hi;
};
__device__ __forceinline__ bool lt128(U128 a, U128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

// 2-bit-group reversal of the 2k-bit value (the [u64;2] analogue of LexHasher, hash.rs:60-71)
__device__ __forceinline__ U128 lex_hash128(U128 c, u32 k) {
    const u32 sh = 2u * (64u - k);  // 0..62
    const u64 rl = revgroups64(c.hi), rh = revgroups64(c.lo);
    U128 h;
    h.lo = sh ? ((rl >> sh) | (rh << (64u - sh))) : rl;
    h.hi = rh >> sh;
    return h;
}

// same control flow as the iterator (canonical_kmer_iterator.rs:42-70), arithmetic of kmer.rs:91-102 on 128 bits
template <bool AHEAD = true, typename Emit>
__device__ __forceinline__ void roll_read2(const uint8_t* __restrict__ s, u32 len, u32 k, Emit&& emit) {
    const u32 kb = 2u * k;  // 66..128
    const U128 mask = {~0ull, kb >= 128u ? ~0ull : ((1ull << (kb - 64u)) - 1ull)};
    const u32 top = kb - 2u - 64u;  // bit position of the newest base inside .hi (k>=33)
    U128 fw = {0, 0}, rc = {~0ull, ~0ull};
    int last_invalid = -1;
    auto step = [&](u32 c, u32 l) {
        const u32 b = encode_base(c);
        if (b < 4u) {
            fw.lo = (fw.lo >> 2) | (fw.hi << 62);
            fw.hi = (fw.hi >> 2) | ((u64)b << top);
            rc.hi = ((rc.hi << 2) | (rc.lo >> 62)) & mask.hi;
            rc.lo = (rc.lo << 2) | (u64)(3u - b);
            if ((int)l - last_invalid >= (int)k) emit(l + 1u - k, fw, rc);
        } else {
            last_invalid = (int)l;
        }
    };
    u32 l = 0;
    if constexpr (AHEAD) {        // (the loads two groups ahead of the walk: roll_read)
        u64 v1 = 0, v2 = 0;
        if (8u <= len) __builtin_memcpy(&v1, s, 8);
        if (16u <= len) __builtin_memcpy(&v2, s + 8u, 8);
        for (; l + 8u <= len; l += 8u) {
            const u64 v = v1;
            v1 = v2;
            if (l + 24u <= len) __builtin_memcpy(&v2, s + l + 16u, 8);
#pragma unroll
            for (u32 j = 0; j < 8u; ++j) step((u32)(v >> (8u * j)) & 0xFFu, l + j);
        }
    } else {
        for (; l + 8u <= len; l += 8u) {
            u64 v;
            __builtin_memcpy(&v, s + l, 8);
#pragma unroll
            for (u32 j = 0; j < 8u; ++j) step((u32)(v >> (8u * j)) & 0xFFu, l + j);
        }
    }
    for (; l < len; ++l) step(s[l], l);
}

}  // namespace kmx
