// kmx_fastx.hip -- FASTA / FASTQ record splitting on the device.
//
// BUILD-DEFINED: the reference has no parser (its callers hand it `&[u8]` reads); this is the step that turns a file
// image in HBM into what kmx_canonical_reduce & co. take for real data: the reads back to back (`bases`) and the
// `offsets` array (read r = bases[offsets[r], offsets[r+1])).  Semantics (oracle: kmo_fastx_parse / oracle.fastx_parse):
//   * lines end at '\n'; every '\r' on a sequence line is dropped; the last line may lack its '\n';
//   * FASTQ: strict 4-line records, line i is a read iff i % 4 == 1 (quality lines may hold any byte but '\n');
//   * FASTA: a line starting with '>' opens a record, every other line up to the next '>' line is its sequence;
//   * the text must start with '@' / '>'; bases are copied verbatim (case, N: the k-mer kernels deal with them).
//
// It is a stream compaction whose keep/drop decision depends on state carried from the start of the file (line number
// mod 4; "is this line a header"), so it runs as the usual three passes over 128 KiB chunks:
//   1. fastx_summarise_kernel: per chunk, what it does to the state and how many bytes / records it emits for each
//      state it may be entered in (FASTQ: 4 phases; FASTA: a chunk whose first line start is known emits a fixed amount
//      plus what its leading partial line adds if that line is sequence);
//   2. fastx_scan_blocks_kernel + fastx_scan_chunks_kernel: the same summary per block of 1024 chunks, then every block composes
//      the aggregates before it and runs the state through its chunk summaries: entry state, exclusive sums of the emitted
//      bytes and records per chunk, totals;
//   3. fastx_emit_kernel: every chunk again, now knowing its entry state and output positions: compacts the bases
//      (through an LDS image of each row's output, written back in whole aligned 16-byte pieces) and writes the offsets.
// Traffic: the text is read twice, the bases written once (HBM-bound byte work: no LDS staging of the text, a lane
// owns 64 consecutive bytes per step and turns them into 64-bit masks -- newlines, line classes, bytes to keep -- a dword
// at a time; only the 16-byte pieces with a line end inside are then copied byte by byte).  Both passes run near both of their limits
// (5.4-6.1 TB/s, 70-80 % VALU issue: profiles/r05_pmc_fastq.txt).  One pass over the text with chained tile descriptors was built and
// measured twice: 16 KiB tiles, two look-backs (round 3, profiles/r03_fastx_one_pass.txt): 6x slower; 64 KiB tiles held in registers, one
// look-back of 1024 descriptors per round trip (round 5, profiles/r05_fastq_one_pass.txt, tools/patches/fastq_one_pass.patch): 1.5x slower.
#include "kmx_device.h"

namespace kmx {

namespace {

// Pass 3 writes its bytes through an LDS image of the row's output; the aligned 16-byte stores of that write-back carry the nt hint
// (whole lines from consecutive lanes: FASTA +3 %, FASTQ unchanged -- profiles/r03_fastx_staged_emit.txt).
constexpr u32 FX_THREADS = 256;
constexpr u32 FX_LANE = 64;                  // consecutive bytes a lane owns per step (four dwordx4 loads)
constexpr u32 FX_ROW = FX_THREADS * FX_LANE; // bytes a block handles per step (16 KiB)
constexpr u32 FX_ROWS = 8;                   // steps per chunk
constexpr u32 FX_OB = FX_THREADS * 64u + 32u;   // staged output of a row: its bytes behind up to 15 of alignment
constexpr u64 FX_CHUNK = (u64)FX_ROW * FX_ROWS;   // 128 KiB
constexpr u32 FX_SUM_WORDS = 12;             // u32 per chunk summary
constexpr u32 FX_PFX_WORDS = 4;              // u64 per chunk prefix: entry state, first output byte, first record, -

// line types of the FASTA state (0 = "whatever the chunk / row was entered with")
constexpr u32 T_NONE = 0, T_SEQ = 2, T_HDR = 3;

// inclusive scan over the block of NW waves; `tmp` holds NW elements
template <u32 NW, class T, class Op>
__device__ __forceinline__ T block_scan_incl(T v, Op op, T* tmp, T& total) {
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T o = __shfl_up(v, d, WAVE);
        if (lane >= (u32)d) v = op(o, v);
    }
    __syncthreads();                  // tmp may still be read from the previous call
    if (lane == 63u) tmp[wv] = v;
    __syncthreads();
    T pre = v, acc = v;
    bool have = false;
#pragma unroll
    for (u32 w = 0; w < NW; ++w) {
        const T x = tmp[w];
        if (w < wv) { pre = have ? op(pre, x) : x; have = true; }
        acc = w == 0 ? x : op(acc, x);
    }
    total = acc;
    return have ? op(pre, v) : v;
}

struct OpAdd32 { __device__ u32 operator()(u32 a, u32 b) const { return a + b; } };
struct OpMax32 { __device__ u32 operator()(u32 a, u32 b) const { return a > b ? a : b; } };

// ---- the 64 bytes of a lane as bit masks (bit i = byte i), computed a dword at a time (SWAR), never a byte at a time
struct Lane64 {
    u32 w[16];
    u64 val;        // bytes inside the text
    u64 nextval;    // bytes whose successor is inside the text
    u32 next;       // text[p + 64] (0 past the end)
};

__device__ __forceinline__ Lane64 load_lane(const uint8_t* __restrict__ text, u64 n, u64 p) {
    Lane64 r;
    r.next = 0;
    if (p + FX_LANE <= n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(text + p + 16u * q);   // (plain: a lane's four loads share their lines with its neighbours'; with the nt hint the parse is 25 % slower)
            r.w[4 * q] = v.x; r.w[4 * q + 1] = v.y; r.w[4 * q + 2] = v.z; r.w[4 * q + 3] = v.w;
        }
        r.val = ~0ull;
        const bool more = p + FX_LANE < n;
        if (more) r.next = text[p + FX_LANE];
        r.nextval = more ? ~0ull : (~0ull >> 1);
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) r.w[q] = 0;
        const u32 nv = p < n ? (u32)(n - p) : 0u;
        for (u32 i = 0; i < nv; ++i) {
            const u32 b = text[p + i];
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if ((i >> 2) == (u32)q) r.w[q] |= b << (8u * (i & 3u));
        }
        r.val = nv ? (~0ull >> (64u - nv)) : 0ull;
        r.nextval = r.val >> 1;
    }
    return r;
}

// exact: bit i set iff byte i of the 64 equals c
__device__ __forceinline__ u64 eq_mask(const Lane64& d, u32 c) {
    const u32 pat = c * 0x01010101u;
    u32 m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u32 ind[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 x = d.w[4 * q + j] ^ pat;
            const u32 z = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;     // bit 7 of a byte set iff the byte is non-zero
            ind[j] = (~z >> 7) & 0x01010101u;
        }
        // v_dot4_u32_u8 gathers the four 0/1 bytes of a dword into a nibble
        const u32 lo = __builtin_amdgcn_udot4(ind[1], 0x80402010u, __builtin_amdgcn_udot4(ind[0], 0x08040201u, 0u, false), false);
        const u32 hi = __builtin_amdgcn_udot4(ind[3], 0x80402010u, __builtin_amdgcn_udot4(ind[2], 0x08040201u, 0u, false), false);
        m[q] = lo | (hi << 8);
    }
    return (u64)(m[0] | (m[1] << 16)) | ((u64)(m[2] | (m[3] << 16)) << 32);
}
// cheaper, with false positives only on bytes c^1 directly above (in the same dword) a byte that does equal c
__device__ __forceinline__ u64 eq_mask_approx(const Lane64& d, u32 c) {
    const u32 pat = c * 0x01010101u;
    u32 m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u32 ind[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 x = d.w[4 * q + j] ^ pat;
            ind[j] = (((x - 0x01010101u) & ~x) >> 7) & 0x01010101u;
        }
        const u32 lo = __builtin_amdgcn_udot4(ind[1], 0x80402010u, __builtin_amdgcn_udot4(ind[0], 0x08040201u, 0u, false), false);
        const u32 hi = __builtin_amdgcn_udot4(ind[3], 0x80402010u, __builtin_amdgcn_udot4(ind[2], 0x08040201u, 0u, false), false);
        m[q] = lo | (hi << 8);
    }
    return (u64)(m[0] | (m[1] << 16)) | ((u64)(m[2] | (m[3] << 16)) << 32);
}
// The '\n' mask, exact, in 4 instructions a dword (eq_mask takes 6): with x = w ^ 0x0A0A0A0A the classic
// (x - 0x01010101) & ~x & 0x80808080 flags every zero byte and, falsely, only bytes equal to 1 above one (the borrow); those
// have bit 0 set, a zero byte has not: & ~(x << 7) drops them.  The indicators stay at bit 7 of their bytes -- v_dot4_u32_u8
// gathers them all the same, the nibbles come out shifted by 7.
// CTL: also says (`ctl` != 0) whether a byte of 0x08..0x0F other than '\n' MAY be among the 64 -- what a '\r' looks like; almost no
// file has one, so the exact '\r' mask is computed only by the waves that may hold one.  s = x - 0x01010101 is below 7 in exactly
// those bytes (a borrow from the byte below only lowers it: never a false negative), and "some byte below 7" is the same
// subtract-and-mask once more: 2 instructions a dword on top of the newline test.
template <bool CTL>
__device__ __forceinline__ u64 nl_mask(const Lane64& d, u32& ctl) {
    u32 m[4];
    u32 acc = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u32 ind[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 x = d.w[4 * q + j] ^ 0x0A0A0A0Au;
            const u32 sb = x - 0x01010101u;
            const u32 a = (x << 7) | x;
            ind[j] = __builtin_amdgcn_bitop3_b32(sb, a, 0x80808080u, 0x20);          // sb & ~a & 0x80808080
            if constexpr (CTL) acc = __builtin_amdgcn_bitop3_b32(acc, sb - 0x07070707u, sb, 0xF4);   // acc | (t & ~sb)
        }
        const u32 lo = __builtin_amdgcn_udot4(ind[1], 0x80402010u, __builtin_amdgcn_udot4(ind[0], 0x08040201u, 0u, false), false);
        const u32 hi = __builtin_amdgcn_udot4(ind[3], 0x80402010u, __builtin_amdgcn_udot4(ind[2], 0x08040201u, 0u, false), false);
        m[q] = (lo | (hi << 8)) >> 7;
    }
    ctl = acc & 0x80808080u;
    return (u64)(m[0] | (m[1] << 16)) | ((u64)(m[2] | (m[3] << 16)) << 32);
}
// newlines and the bytes that are neither '\n' nor '\r'.  CR = false: the caller knows that the text around has no '\r'
// (the chunk summary of the counting pass says so)
template <bool CR>
__device__ __forceinline__ u64 nl_and_keep(const Lane64& d, u64& keep, u32& suspect) {
    u32 ctl = 0;
    const u64 nl = nl_mask<CR>(d, ctl);
    u64 cr = 0;
    if constexpr (CR) {
        if (__any(ctl != 0u)) cr = eq_mask(d, '\r');
        suspect |= ctl;
    }
    keep = ~nl & ~cr & d.val;
    return nl;
}
__device__ __forceinline__ u64 prefix_xor_excl(u64 x) {   // bit i = parity of the bits of x below i
    x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
    return x << 1;
}
__device__ __forceinline__ u32 pc64(u64 x) { return (u32)__builtin_popcountll(x); }

// ---- FASTQ: state = line number mod 4.  Per lane: its newlines, and for every line number j (relative to the line
// the lane's first byte lies on, mod 4) the mask of the bytes on such lines.
struct FqLane {
    u64 nl, keep, start;   // newlines; bytes that are neither \n nor \r; newlines after which the text goes on (a line starts there)
    u64 b0, b1;            // bits 0 and 1 of the number of newlines before a byte (its line number relative to the lane's first byte)
};
template <bool CR>
__device__ __forceinline__ FqLane fq_analyse(const Lane64& d, u32& suspect) {
    FqLane r;
    r.nl = nl_and_keep<CR>(d, r.keep, suspect);
    r.start = r.nl & d.nextval;
    r.b0 = prefix_xor_excl(r.nl);                    // bit 0 of the number of newlines before a byte
    r.b1 = prefix_xor_excl(r.nl & r.b0);             // bit 1: toggles after a newline that makes the count even again
    return r;
}
// (c1, c0) = the line number of every byte mod 4, given the one of the lane's first byte: a 2-bit add on the bit planes
__device__ __forceinline__ void fq_classes(const FqLane& a, u32 first, u64& c0, u64& c1) {
    const u64 p0 = (first & 1u) ? ~0ull : 0ull, p1 = (first & 2u) ? ~0ull : 0ull;
    c0 = a.b0 ^ p0;
    c1 = a.b1 ^ p1 ^ (a.b0 & p0);
}
// newlines before the lane in the block's row, mod 4 (all the FASTQ state there is), from two ballots of the lanes' counts
// instead of a shuffle scan; `tmp`: one word per wave; total = the row's newlines mod 4
// PRE = false: the caller vouches that nobody still reads `tmp` from an earlier call (another barrier lies between, or the calls alternate
// between two arrays) -- one barrier per call instead of two
template <u32 NW, bool PRE = true>
__device__ __forceinline__ u32 block_prefix_mod4(u32 c, u32* tmp, u32& total) {
    const u32 wv = threadIdx.x >> 6;
    const u64 e0 = __ballot((c & 1u) != 0u), e1 = __ballot((c & 2u) != 0u);
    const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(e0 >> 32), __builtin_amdgcn_mbcnt_lo((u32)e0, 0u)) +
                      2u * __builtin_amdgcn_mbcnt_hi((u32)(e1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)e1, 0u));
    if constexpr (PRE) __syncthreads();                  // tmp may still be read from the previous call
    if ((threadIdx.x & 63u) == 0u) tmp[wv] = pc64(e0) + 2u * pc64(e1);
    __syncthreads();
    u32 pre = 0, acc = 0;
#pragma unroll
    for (u32 w = 0; w < NW; ++w) {    // (NW a constant: with blockDim.x >> 6 hipcc builds a vectorised loop and its remainders -- 90 instructions -- around these four words)
        const u32 x = tmp[w];
        pre += w < wv ? x : 0u;
        acc += x;
    }
    total = acc & 3u;
    return (pre + below) & 3u;
}
// inclusive add scan over the block, the wave part as six DPP adds (row shifts, then the two row broadcasts)
template <u32 NW, bool PRE = true>
__device__ __forceinline__ u32 block_scan_add_dpp(u32 v, u32* tmp, u32& total) {
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    if constexpr (PRE) __syncthreads();
    if (lane == 63u) tmp[wv] = v;
    __syncthreads();
    u32 pre = 0, acc = 0;
#pragma unroll
    for (u32 w = 0; w < NW; ++w) {
        const u32 x = tmp[w];
        pre += w < wv ? x : 0u;
        acc += x;
    }
    total = acc;
    return pre + v;
}

// ---- FASTA: state = type of the current line.  The type of a line is decided by its first byte; a line start is
// attributed to the newline before it.
struct FaLane {
    u64 nl, keep;
    u64 hdr_nl;     // newlines followed by '>' (a record opens there)
    u32 def;        // type of the last line that starts after a newline of this lane (T_NONE: none does)
    u64 inh;        // bytes before the lane's first newline: their line began earlier
    u64 hdr;        // bytes on header lines that start inside the lane
};
template <bool CR>
__device__ __forceinline__ FaLane fa_analyse(const Lane64& d, u32& suspect) {
    FaLane r;
    r.nl = nl_and_keep<CR>(d, r.keep, suspect);
    const u64 start = r.nl & d.nextval;
    // the false positives of the cheap mask sit right above a '>' (never after a newline), so they drop out here
    const u64 gt_next = (eq_mask_approx(d, '>') >> 1) | ((u64)(d.next == '>') << 63);
    r.hdr_nl = start & gt_next;
    const u64 seq_nl = start & ~gt_next;
    r.def = start ? (r.hdr_nl > seq_nl ? T_HDR : T_SEQ) : T_NONE;
    r.inh = r.nl ? ((r.nl & (0ull - r.nl)) - 1ull) : ~0ull;
    // header lines: from the byte after a hdr_nl newline up to the next newline; subtracting the opening bits from the
    // newline mask borrows from exactly the newline that closes each (or runs off the top: the line goes on)
    r.hdr = (r.nl - (r.hdr_nl << 1)) & ~r.nl;
    return r;
}

// copy the bytes of the lane selected by `ks` (ascending) to dst[0..popcount(ks))
__device__ __forceinline__ void emit_bytes(const Lane64& d, u64 ks, uint8_t* __restrict__ dst) {
    struct __attribute__((packed)) P16 { u32 a, b, c, e; };
    u32 partial = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32 k = (u32)(ks >> (16 * q)) & 0xFFFFu;
        if (k == 0xFFFFu) {          // the common case: one unaligned 16-byte store (plain: the L2 merges the four pieces of a line; with the nt hint pass 3 takes 6-7 instead of 1.9 ms: profiles/r03_fastx_variants.txt)
            P16 v{d.w[4 * q], d.w[4 * q + 1], d.w[4 * q + 2], d.w[4 * q + 3]};
            *reinterpret_cast<P16*>(dst + pc64(ks & ((1ull << (16 * q)) - 1ull))) = v;
        } else if (k) {
            partial |= 1u << q;
        }
    }
    while (partial) {                // pieces with a line end inside
        const u32 q = (u32)__builtin_ctz(partial);
        partial &= partial - 1u;
        u32 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = q == 0 ? d.w[j] : q == 1 ? d.w[4 + j] : q == 2 ? d.w[8 + j] : d.w[12 + j];
        u32 k = (u32)(ks >> (16u * q)) & 0xFFFFu;
        uint8_t* o = dst + pc64(ks & ((1ull << (16u * q)) - 1ull));
        const u64 lo0 = (u64)a[0] | ((u64)a[1] << 32), hi0 = (u64)a[2] | ((u64)a[3] << 32);
        while (k) {
            // one run of bytes (the end of a line, the beginning of the next ...): shift it down to byte 0, then 8 + 4 + 2 + 1
            struct __attribute__((packed)) P8 { u64 v; };
            struct __attribute__((packed)) P4 { u32 v; };
            struct __attribute__((packed)) P2 { uint16_t v; };
            u32 sb = (u32)__builtin_ctz(k);
            const u32 len = (u32)__builtin_ctz(~(k >> sb));
            k &= ~(((1u << len) - 1u) << sb);
            u64 lo = lo0, hi = hi0;
            if (sb >= 8u) { lo = hi; hi = 0; sb -= 8u; }
            if (sb) { lo = (lo >> (8u * sb)) | (hi << (64u - 8u * sb)); hi >>= 8u * sb; }
            if (len & 8u) { reinterpret_cast<P8*>(o)->v = lo; o += 8; lo = hi; }
            if (len & 4u) { reinterpret_cast<P4*>(o)->v = (u32)lo; o += 4; lo >>= 32; }
            if (len & 2u) { reinterpret_cast<P2*>(o)->v = (uint16_t)lo; o += 2; lo >>= 16; }
            if (len & 1u) { *o = (uint8_t)lo; o += 1; }
        }
    }
}

}  // namespace

// pass 1.  Summary words of a chunk: [0..3] FASTQ: bytes kept on the lines of (chunk-relative) number j mod 4 / FASTA: [0] bytes
// emitted whatever the state, [1] more if the chunk is entered on a sequence line, [2] records opened; [4..7] FASTQ: lines of
// number j mod 4 that START in the chunk; [8] what the chunk does to the state; [9] non-zero: the chunk may hold a '\r'
template <bool FASTA>
__global__ void __launch_bounds__(FX_THREADS)
fastx_summarise_kernel(const uint8_t* __restrict__ text, u64 n, u32* __restrict__ summ) {
    __shared__ u32 tmp32[2][FX_THREADS / 64];
    __shared__ u32 wave_last[FX_THREADS / 64];
    __shared__ u32 red[8];
    const u64 chunk = blockIdx.x;
    const u64 c0 = chunk * FX_CHUNK;
    if (threadIdx.x < 8) red[threadIdx.x] = 0;
    __syncthreads();
    u32 acc[6] = {0, 0, 0, 0, 0, 0};   // [4]: FASTQ: line starts; [5]: '\r' suspects
    u32 state = 0;     // FASTQ: newlines so far in the chunk, mod 4; FASTA: type of the current line (T_NONE: as entered)
    // (the next row's 64 bytes are requested before this row is analysed: one HBM round trip per row hidden)
    Lane64 d_next = c0 < n ? load_lane(text, n, c0 + threadIdx.x * FX_LANE) : Lane64{};
    for (u32 row = 0; row < FX_ROWS; ++row) {
        const u64 p = c0 + (u64)row * FX_ROW + threadIdx.x * FX_LANE;
        if (c0 + (u64)row * FX_ROW >= n) break;
        const Lane64 d = d_next;
        if (row + 1u < FX_ROWS && c0 + (u64)(row + 1u) * FX_ROW < n) d_next = load_lane(text, n, p + FX_ROW);
        if constexpr (!FASTA) {
            const FqLane a = fq_analyse<true>(d, acc[5]);
            u32 tot;
            const u32 ph = (state + block_prefix_mod4<FX_THREADS / 64, false>(pc64(a.nl), tmp32[row & 1u], tot)) & 3u;   // line number (relative to the chunk) at the lane's first byte (the rows alternate between two arrays: one barrier per row)
            u64 c0m, c1m;
            fq_classes(a, ph, c0m, c1m);
            acc[0] += pc64(a.keep & ~c1m & ~c0m);
            acc[1] += pc64(a.keep & ~c1m & c0m);
            acc[2] += pc64(a.keep & c1m & ~c0m);
            acc[3] += pc64(a.keep & c1m & c0m);
            acc[4] += pc64(a.start);
            state = (state + tot) & 3u;
        } else {
            const FaLane a = fa_analyse<true>(d, acc[5]);
            u32 tot;
            const u32 key = a.def ? (((threadIdx.x + 1u) << 2) | a.def) : 0u;
            const u32 incl = block_scan_incl<FX_THREADS / 64>(key, OpMax32{}, tmp32[0], tot);
            const u32 ex = __shfl_up(incl, 1, WAVE);            // exclusive: the lane before (across waves through LDS)
            __syncthreads();
            if ((threadIdx.x & 63u) == 63u) wave_last[threadIdx.x >> 6] = incl;
            __syncthreads();
            const u32 prev = (threadIdx.x & 63u) ? ex : (threadIdx.x ? wave_last[(threadIdx.x >> 6) - 1u] : 0u);
            const u32 in = prev ? (prev & 3u) : state;          // type of the line the lane's first bytes lie on
            const u32 inh = pc64(a.keep & a.inh), fixed = pc64(a.keep & ~a.inh & ~a.hdr);
            acc[0] += fixed + (in == T_SEQ ? inh : 0u);         // emitted whatever the chunk is entered with
            acc[1] += in == T_NONE ? inh : 0u;                  // emitted if the chunk is entered on a sequence line
            acc[2] += pc64(a.hdr_nl);
            if (tot) state = tot & 3u;
        }
    }
    // block totals
    acc[5] = acc[5] ? 1u : 0u;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        u32 v = acc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&red[q], v);
    }
    __syncthreads();
    u32* s = summ + chunk * FX_SUM_WORDS;
    if (threadIdx.x < 4) s[threadIdx.x] = red[threadIdx.x];
    if constexpr (!FASTA) {
        // The k-th line start of the chunk (k = 1, 2, ...) opens a line of chunk-relative number k: of C starts, those of number
        // j mod 4 follow from C alone.  (`start` leaves out only a newline that ends the text: the last of them.)
        if (threadIdx.x >= 4 && threadIdx.x < 8) {
            const u32 j = threadIdx.x - 4u, C = red[4];
            s[threadIdx.x] = j == 0u ? (C >> 2) : ((C + 4u - j) >> 2);
        }
    } else {
        if (threadIdx.x >= 4 && threadIdx.x < 8) s[threadIdx.x] = 0;
    }
    if (threadIdx.x == 8) s[8] = state;
    if (threadIdx.x == 9) s[9] = red[5];
}

// pass 2, in two small kernels over blocks of 1024 chunk summaries (one thread per chunk; the single block that walked all the
// summaries took 0.24 ms for the 42 000 chunks of a 5.5 GB text):
//   (a) fastx_scan_blocks_kernel: what a BLOCK of chunks does to the state and emits for every state it may be entered in -- the
//       same summary one level up (block_agg, FX_AGG_WORDS words per block);
//   (b) fastx_scan_chunks_kernel: every block composes the aggregates of the blocks before it (a few dozen), then places its chunks:
//       prefix[chunk] = {entry state, first output byte, first record index}; the last block writes totals[0..1] = reads, bases.
constexpr u32 FX_SCAN_THREADS = 1024;
constexpr u32 FX_AGG_WORDS = 12;
// exclusive max-scan of the line-type keys over the block: the key of the nearest thread before this one that has one (0: none)
__device__ __forceinline__ u32 block_prev_key(u32 key, u32* tmp, u32* wave_last, u32& tot) {
    const u32 incl = block_scan_incl<FX_SCAN_THREADS / 64>(key, OpMax32{}, tmp, tot);
    const u32 ex = __shfl_up(incl, 1, WAVE);
    __syncthreads();
    if ((threadIdx.x & 63u) == 63u) wave_last[threadIdx.x >> 6] = incl;
    __syncthreads();
    return (threadIdx.x & 63u) ? ex : (threadIdx.x ? wave_last[(threadIdx.x >> 6) - 1u] : 0u);
}
__device__ __forceinline__ u32 pick4(const uint4& v, u32 j) { return j == 0u ? v.x : j == 1u ? v.y : j == 2u ? v.z : v.w; }

template <bool FASTA>
__global__ void __launch_bounds__(FX_SCAN_THREADS)
fastx_scan_blocks_kernel(const u32* __restrict__ summ, u64 n_chunks, u32* __restrict__ block_agg) {
    __shared__ u32 tmp32[FX_SCAN_THREADS / 64];
    __shared__ u32 wave_last[FX_SCAN_THREADS / 64];
    __shared__ unsigned long long red[8];
    const u64 c = (u64)blockIdx.x * FX_SCAN_THREADS + threadIdx.x;
    const bool live = c < n_chunks;
    const u32* sp = summ + (live ? c : n_chunks - 1u) * FX_SUM_WORDS;
    uint4 K = *reinterpret_cast<const uint4*>(sp), R = *reinterpret_cast<const uint4*>(sp + 4);
    u32 S = sp[8];
    if (!live) { K = make_uint4(0, 0, 0, 0); R = K; S = 0; }
    if (threadIdx.x < 8) red[threadIdx.x] = 0;
    u64 v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 what;     // what the block does to the state
    if constexpr (!FASTA) {
        const u32 pre = block_prefix_mod4<FX_SCAN_THREADS / 64>(S & 3u, tmp32, what);     // (its barriers also cover red[])
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {      // the block's line number j is the chunk's j - pre
            v[j] = pick4(K, (j - pre) & 3u);
            v[4 + j] = pick4(R, (j - pre) & 3u);
        }
    } else {
        u32 tot;
        const u32 prev = block_prev_key(S ? (((threadIdx.x + 1u) << 2) | S) : 0u, tmp32, wave_last, tot);
        const u32 in = prev ? (prev & 3u) : T_NONE;
        v[0] = (u64)K.x + (in == T_SEQ ? K.y : 0u);       // emitted whatever the block is entered with
        v[1] = in == T_NONE ? K.y : 0u;                    // more if it is entered on a sequence line
        v[2] = K.z;
        what = tot ? (tot & 3u) : 0u;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        u64 x = v[q];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, WAVE);
        if ((threadIdx.x & 63u) == 0 && x) atomicAdd(&red[q], (unsigned long long)x);
    }
    __syncthreads();
    // (a block of 1024 chunks emits < 2^27 bytes: the aggregates fit their 32-bit words)
    u32* ag = block_agg + (u64)blockIdx.x * FX_AGG_WORDS;
    if (threadIdx.x < 8) ag[threadIdx.x] = (u32)red[threadIdx.x];
    if (threadIdx.x == 8) ag[8] = what;
}

template <bool FASTA>
__global__ void __launch_bounds__(FX_SCAN_THREADS)
fastx_scan_chunks_kernel(const u32* __restrict__ summ, u64 n_chunks, const u32* __restrict__ block_agg, u64* __restrict__ prefix,
                         unsigned long long* __restrict__ totals) {
    __shared__ u32 tmp32[FX_SCAN_THREADS / 64];
    __shared__ u32 wave_last[FX_SCAN_THREADS / 64];
    __shared__ u32 ags[FX_SCAN_THREADS][9];
    __shared__ u64 sh_entry[3];
    // ---- where the block begins: the aggregates of the blocks before it, composed in order (the text starts with a header line
    // ('>' checked by the host) / on line 0; FASTA: record 0 opens at byte 0, no newline announces it)
    u64 state = FASTA ? T_HDR : 0, out_pos = 0, rec = FASTA ? 1 : 0;
    for (u64 base = 0; base < blockIdx.x; base += FX_SCAN_THREADS) {
        const u64 bb = base + threadIdx.x;
        __syncthreads();
        if (bb < blockIdx.x) {
#pragma unroll
            for (int q = 0; q < 9; ++q) ags[threadIdx.x][q] = block_agg[bb * FX_AGG_WORDS + q];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const u64 cnt = blockIdx.x - base < FX_SCAN_THREADS ? blockIdx.x - base : FX_SCAN_THREADS;
            for (u64 i = 0; i < cnt; ++i) {
                const u32* a = ags[i];
                if constexpr (FASTA) {
                    out_pos += (u64)a[0] + (state == T_SEQ ? a[1] : 0u);
                    rec += a[2];
                    state = a[8] ? a[8] : state;
                } else {
                    const u32 j = (1u - (u32)state) & 3u;       // file lines 1 (mod 4) are lines 1 - state of the block
                    out_pos += a[j];
                    rec += a[4u + j];
                    state = (state + a[8]) & 3u;
                }
            }
        }
    }
    if (threadIdx.x == 0) { sh_entry[0] = state; sh_entry[1] = out_pos; sh_entry[2] = rec; }
    __syncthreads();
    state = sh_entry[0]; out_pos = sh_entry[1]; rec = sh_entry[2];
    // ---- the block's chunks
    const u64 c = (u64)blockIdx.x * FX_SCAN_THREADS + threadIdx.x;
    const bool live = c < n_chunks;
    const u32* sp = summ + (live ? c : n_chunks - 1u) * FX_SUM_WORDS;
    uint4 K = *reinterpret_cast<const uint4*>(sp), R = *reinterpret_cast<const uint4*>(sp + 4);
    u32 S = sp[8];
    if (!live) { K = make_uint4(0, 0, 0, 0); R = K; S = 0; }
    u32 in, kept, recs;
    if constexpr (!FASTA) {
        u32 tot;
        in = ((u32)state + block_prefix_mod4<FX_SCAN_THREADS / 64>(S & 3u, tmp32, tot)) & 3u;
        kept = pick4(K, (1u - in) & 3u);
        recs = pick4(R, (1u - in) & 3u);
    } else {
        u32 tot;
        const u32 prev = block_prev_key(S ? (((threadIdx.x + 1u) << 2) | S) : 0u, tmp32, wave_last, tot);
        in = prev ? (prev & 3u) : (u32)state;
        kept = K.x + (in == T_SEQ ? K.y : 0u);
        recs = K.z;
    }
    u32 tk, tr;
    const u32 ik = block_scan_add_dpp<FX_SCAN_THREADS / 64>(kept, tmp32, tk);
    const u32 ir = block_scan_add_dpp<FX_SCAN_THREADS / 64>(recs, tmp32, tr);
    if (live) {
        u64* pf = prefix + c * FX_PFX_WORDS;
        *reinterpret_cast<ulonglong2*>(pf) = make_ulonglong2((u64)in, out_pos + (ik - kept));
        pf[2] = rec + (ir - recs);
    }
    if (blockIdx.x + 1u == gridDim.x && threadIdx.x == 0) {
        totals[0] = rec + tr;
        totals[1] = out_pos + tk;
    }
}

// pass 3
template <bool FASTA, bool CR>
__device__ __forceinline__ void fastx_emit_rows(const uint8_t* __restrict__ text, u64 n, u64 c0, u32 state, u64 out_pos, u64 rec,
                                                uint8_t* __restrict__ bases, u64* __restrict__ offsets, u32* tmp32, u32* wave_last, uint8_t* stage) {
    // (the next row's 64 bytes are requested before this row is analysed: one HBM round trip per row hidden)
    Lane64 d_next = c0 < n ? load_lane(text, n, c0 + threadIdx.x * FX_LANE) : Lane64{};
    u32 unused = 0;
    u32 foreign = 0;      // leading bytes of the current row's piece 0 that belong to the chunk before
    for (u32 row = 0; row < FX_ROWS; ++row) {
        const u64 p = c0 + (u64)row * FX_ROW + threadIdx.x * FX_LANE;
        if (c0 + (u64)row * FX_ROW >= n) break;
        const Lane64 d = d_next;
        if (row + 1u < FX_ROWS && c0 + (u64)(row + 1u) * FX_ROW < n) d_next = load_lane(text, n, p + FX_ROW);
        u64 ks, rs;       // bytes to emit; newlines after which a read begins
        u32 tot;
        if constexpr (!FASTA) {
            const FqLane a = fq_analyse<CR>(d, unused);
            const u32 in = (state + block_prefix_mod4<FX_THREADS / 64, false>(pc64(a.nl), tmp32, tot)) & 3u;   // the file's line number (mod 4) at the lane's first byte (no barrier before: the row's other two lie between two uses of tmp32)
            state = (state + tot) & 3u;
            u64 c0m, c1m;
            fq_classes(a, in, c0m, c1m);
            ks = a.keep & ~c1m & c0m;      // lines 1 (mod 4): the reads
            rs = a.start & ~c1m & ~c0m;    // newlines that end a header line (0 mod 4)
        } else {
            const FaLane a = fa_analyse<CR>(d, unused);
            const u32 key = a.def ? (((threadIdx.x + 1u) << 2) | a.def) : 0u;
            const u32 incl = block_scan_incl<FX_THREADS / 64>(key, OpMax32{}, tmp32, tot);
            const u32 ex = __shfl_up(incl, 1, WAVE);
            __syncthreads();
            if ((threadIdx.x & 63u) == 63u) wave_last[threadIdx.x >> 6] = incl;
            __syncthreads();
            const u32 prev = (threadIdx.x & 63u) ? ex : (threadIdx.x ? wave_last[(threadIdx.x >> 6) - 1u] : 0u);
            const u32 in = prev ? (prev & 3u) : state;
            if (tot) state = tot & 3u;
            ks = a.keep & ((in == T_SEQ ? a.inh : 0ull) | (~a.inh & ~a.hdr));
            rs = a.hdr_nl;
        }
        // output positions: bytes and records packed into one scan (a row emits <= 16384 of either)
        const u32 kept = pc64(ks), recs = pc64(rs);
        u32 tkr;
        // (FASTQ: through wave_last, which that path does not use otherwise -- again the row's other barriers lie between two uses)
        const u32 ikr = FASTA ? block_scan_add_dpp<FX_THREADS / 64>(kept | (recs << 16), tmp32, tkr)
                              : block_scan_add_dpp<FX_THREADS / 64, false>(kept | (recs << 16), wave_last, tkr);
        const u64 o = out_pos + ((ikr & 0xFFFFu) - kept);
        u64 r = rec + ((ikr >> 16) - recs);
        {
            // The row's bytes go out through LDS: the lanes deposit their pieces where they will lie (same offset modulo 16 as in
            // memory), then the block writes whole 16-byte pieces, consecutive lanes consecutive addresses -- full lines instead of
            // four 16-byte stores per lane at a 64-byte stride.  Two buffers: the next row deposits while this one is still read.
            // The row's LAST piece, if the row ends inside it, is not written: it is handed to the next row's buffer (piece 0 there:
            // the next row begins at that offset modulo 16) and goes out whole with it -- only the chunk's first and last piece,
            // shared with other blocks, are written byte by byte (round 5: until then every row's were, up to 30 single-byte copies
            // by one lane of two of the four waves).
            uint8_t* const ob = stage + (row & 1u) * FX_OB;
            uint8_t* const ob_next = stage + ((row + 1u) & 1u) * FX_OB;
            const u32 row_kept = tkr & 0xFFFFu;
            const u32 al = (u32)((reinterpret_cast<uintptr_t>(bases) + out_pos) & 15u);       // where the row's first byte sits in its 16-byte piece
            if (row == 0u) foreign = al;
            if (ks) emit_bytes(d, ks, ob + al + (u32)(o - out_pos));
            __syncthreads();
            uint8_t* const g0 = bases + out_pos - al;                                          // 16-byte aligned
            const u32 end = al + row_kept;
            const bool last_row = row + 1u == FX_ROWS || c0 + (u64)(row + 1u) * FX_ROW >= n;
            for (u32 lo = 16u * threadIdx.x; lo < end; lo += 16u * FX_THREADS) {
                typedef u32 v4u __attribute__((ext_vector_type(4)));
                const u32 b0 = lo == 0u ? foreign : lo;       // the first byte of the piece that is this chunk's
                if (b0 == lo && lo + 16u <= end) {
                    __builtin_nontemporal_store(*reinterpret_cast<const v4u*>(ob + lo), reinterpret_cast<v4u*>(g0 + lo));
                } else if (lo + 16u > end && !last_row) {
                    *reinterpret_cast<v4u*>(ob_next) = *reinterpret_cast<const v4u*>(ob + lo);
                } else {                       // the chunk's first and last piece: shared with the chunks around it, byte by byte
                    const u32 b1 = lo + 16u < end ? lo + 16u : end;
                    for (u32 b = b0; b < b1; ++b) g0[b] = ob[b];
                }
            }
            // (what of the next row's piece 0 is not this chunk's: still the chunk's first piece, or nothing)
            if (end >= 16u) foreign = 0u;
        }
        out_pos += tkr & 0xFFFFu;
        rec += tkr >> 16;
        while (rs) {      // the read that begins after newline q starts at the output position of the bytes kept so far
            const u32 q = (u32)__builtin_ctzll(rs);
            rs &= rs - 1ull;
            offsets[r++] = o + pc64(ks & ((1ull << q) - 1ull));
        }
    }
}
template <bool FASTA>
__global__ void __launch_bounds__(FX_THREADS)
fastx_emit_kernel(const uint8_t* __restrict__ text, u64 n, const u64* __restrict__ prefix, const u32* __restrict__ summ,
                  uint8_t* __restrict__ bases, u64* __restrict__ offsets) {
    __shared__ u32 tmp32[FX_THREADS / 64];
    __shared__ u32 wave_last[FX_THREADS / 64];
    __shared__ __attribute__((aligned(16))) uint8_t stage[2 * FX_OB];
    const u64 chunk = blockIdx.x;
    const u64 c0 = chunk * FX_CHUNK;
    const u64* pf = prefix + chunk * FX_PFX_WORDS;
    if (chunk == 0 && threadIdx.x == 0 && FASTA) offsets[0] = 0;     // record 0 (see fastx_scan_kernel)
    // (a chunk the counting pass found free of '\r' suspects skips that test)
    if (summ[chunk * FX_SUM_WORDS + 9] != 0u)
        fastx_emit_rows<FASTA, true>(text, n, c0, (u32)pf[0], pf[1], pf[2], bases, offsets, tmp32, wave_last, stage);
    else
        fastx_emit_rows<FASTA, false>(text, n, c0, (u32)pf[0], pf[1], pf[2], bases, offsets, tmp32, wave_last, stage);
}

__global__ void fastx_last_offset_kernel(const unsigned long long* __restrict__ totals, u64* __restrict__ offsets) {
    offsets[totals[0]] = totals[1];
}

size_t fastx_scratch_bytes(u64 n_bytes) {
    const u64 n_chunks = (n_bytes + FX_CHUNK - 1u) / FX_CHUNK;
    const u64 n_blocks = (n_chunks + FX_SCAN_THREADS - 1u) / FX_SCAN_THREADS;
    return (size_t)(n_chunks * (FX_SUM_WORDS * 4u + FX_PFX_WORDS * 8u) + n_blocks * FX_AGG_WORDS * 4u + 64u);
}

// counts: passes 1 and 2; totals (device) receives {n_reads, n_bases}
hipError_t launch_fastx_count(const uint8_t* text, u64 n, bool fasta, void* scratch, unsigned long long* totals, hipStream_t st) {
    const u64 n_chunks = (n + FX_CHUNK - 1u) / FX_CHUNK;
    const u64 n_blocks = (n_chunks + FX_SCAN_THREADS - 1u) / FX_SCAN_THREADS;
    u64* prefix = static_cast<u64*>(scratch);
    u32* summ = reinterpret_cast<u32*>(prefix + n_chunks * FX_PFX_WORDS);
    u32* agg = summ + n_chunks * FX_SUM_WORDS;
    if (fasta) {
        hipLaunchKernelGGL(fastx_summarise_kernel<true>, dim3((unsigned)n_chunks), dim3(FX_THREADS), 0, st, text, n, summ);
        hipLaunchKernelGGL(fastx_scan_blocks_kernel<true>, dim3((unsigned)n_blocks), dim3(FX_SCAN_THREADS), 0, st, summ, n_chunks, agg);
        hipLaunchKernelGGL(fastx_scan_chunks_kernel<true>, dim3((unsigned)n_blocks), dim3(FX_SCAN_THREADS), 0, st, summ, n_chunks, agg, prefix, totals);
    } else {
        hipLaunchKernelGGL(fastx_summarise_kernel<false>, dim3((unsigned)n_chunks), dim3(FX_THREADS), 0, st, text, n, summ);
        hipLaunchKernelGGL(fastx_scan_blocks_kernel<false>, dim3((unsigned)n_blocks), dim3(FX_SCAN_THREADS), 0, st, summ, n_chunks, agg);
        hipLaunchKernelGGL(fastx_scan_chunks_kernel<false>, dim3((unsigned)n_blocks), dim3(FX_SCAN_THREADS), 0, st, summ, n_chunks, agg, prefix, totals);
    }
    return hipGetLastError();
}

// pass 3 (after launch_fastx_count on the same scratch)
hipError_t launch_fastx_emit(const uint8_t* text, u64 n, bool fasta, const void* scratch, const unsigned long long* totals,
                             uint8_t* bases, u64* offsets, hipStream_t st) {
    const u64 n_chunks = (n + FX_CHUNK - 1u) / FX_CHUNK;
    const u64* prefix = static_cast<const u64*>(scratch);
    const u32* summ = reinterpret_cast<const u32*>(prefix + n_chunks * FX_PFX_WORDS);
    if (fasta)
        hipLaunchKernelGGL(fastx_emit_kernel<true>, dim3((unsigned)n_chunks), dim3(FX_THREADS), 0, st, text, n, prefix, summ, bases, offsets);
    else
        hipLaunchKernelGGL(fastx_emit_kernel<false>, dim3((unsigned)n_chunks), dim3(FX_THREADS), 0, st, text, n, prefix, summ, bases, offsets);
    hipLaunchKernelGGL(fastx_last_offset_kernel, dim3(1), dim3(1), 0, st, totals, offsets);
    return hipGetLastError();
}

}  // namespace kmx
