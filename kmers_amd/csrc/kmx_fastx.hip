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
//   2. fastx_scan_kernel (one block): runs the state through the chunk summaries, exclusive sums of the emitted bytes
//      and records per chunk, totals;
//   3. fastx_emit_kernel: every chunk again, now knowing its entry state and output positions: compacts the bases
//      and writes the offsets.
// Traffic: the text is read twice, the bases written once (HBM-bound byte work: no LDS staging of the text, a lane
// owns 64 consecutive bytes per step and turns them into 64-bit masks -- newlines, line classes, bytes to keep -- a dword
// at a time; only the 16-byte pieces with a line end inside are then copied byte by byte).
#include "kmx_device.h"

namespace kmx {

namespace {

constexpr u32 FX_THREADS = 256;
constexpr u32 FX_LANE = 64;                  // consecutive bytes a lane owns per step (four dwordx4 loads)
constexpr u32 FX_ROW = FX_THREADS * FX_LANE; // bytes a block handles per step (16 KiB)
constexpr u32 FX_ROWS = 8;                   // steps per chunk
constexpr u64 FX_CHUNK = (u64)FX_ROW * FX_ROWS;   // 128 KiB
constexpr u32 FX_SUM_WORDS = 12;             // u32 per chunk summary
constexpr u32 FX_PFX_WORDS = 4;              // u64 per chunk prefix: entry state, first output byte, first record, -

// line types of the FASTA state (0 = "whatever the chunk / row was entered with")
constexpr u32 T_NONE = 0, T_SEQ = 2, T_HDR = 3;

// inclusive scan over the block (blockDim.x a multiple of 64); `tmp` holds blockDim.x/64 elements
template <class T, class Op>
__device__ __forceinline__ T block_scan_incl(T v, Op op, T* tmp, T& total) {
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
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
    for (u32 w = 0; w < nw; ++w) {
        const T x = tmp[w];
        if (w < wv) { pre = have ? op(pre, x) : x; have = true; }
        acc = w == 0 ? x : op(acc, x);
    }
    total = acc;
    return have ? op(pre, v) : v;
}

struct OpAdd32 { __device__ u32 operator()(u32 a, u32 b) const { return a + b; } };
struct OpAdd64 { __device__ u64 operator()(u64 a, u64 b) const { return a + b; } };
struct OpMax32 { __device__ u32 operator()(u32 a, u32 b) const { return a > b ? a : b; } };
struct OpMax64 { __device__ u64 operator()(u64 a, u64 b) const { return a > b ? a : b; } };

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
            const uint4 v = *reinterpret_cast<const uint4*>(text + p + 16u * q);
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
// does any of the 64 bytes (possibly) equal c?  (same false positives)
__device__ __forceinline__ bool maybe_has(const Lane64& d, u32 c) {
    const u32 pat = c * 0x01010101u;
    u32 acc = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u32 x = d.w[q] ^ pat;
        acc |= (x - 0x01010101u) & ~x;
    }
    return (acc & 0x80808080u) != 0u;
}
// '\r' bytes: almost no file has them, so the exact mask is computed only by the waves that may hold one
__device__ __forceinline__ u64 cr_mask(const Lane64& d) {
    return __any(maybe_has(d, '\r')) ? eq_mask(d, '\r') : 0ull;
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
    u64 cls[4];            // cls[j]: bytes on lines number == j (mod 4), the newline that ends a line included
};
__device__ __forceinline__ FqLane fq_analyse(const Lane64& d) {
    FqLane r;
    r.nl = eq_mask(d, '\n');
    r.keep = ~r.nl & ~cr_mask(d) & d.val;
    r.start = r.nl & d.nextval;
    const u64 b0 = prefix_xor_excl(r.nl);            // bit 0 of the number of newlines before a byte
    const u64 b1 = prefix_xor_excl(r.nl & b0);       // bit 1: toggles after a newline that makes the count even again
    r.cls[0] = ~b1 & ~b0; r.cls[1] = ~b1 & b0; r.cls[2] = b1 & ~b0; r.cls[3] = b1 & b0;
    return r;
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
__device__ __forceinline__ FaLane fa_analyse(const Lane64& d) {
    FaLane r;
    r.nl = eq_mask(d, '\n');
    r.keep = ~r.nl & ~cr_mask(d) & d.val;
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
        if (k == 0xFFFFu) {          // the common case: one unaligned 16-byte store
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

// pass 1
template <bool FASTA>
__global__ void __launch_bounds__(FX_THREADS)
fastx_summarise_kernel(const uint8_t* __restrict__ text, u64 n, u32* __restrict__ summ) {
    __shared__ u32 tmp32[FX_THREADS / 64];
    __shared__ u32 wave_last[FX_THREADS / 64];
    __shared__ u32 red[8];
    const u64 chunk = blockIdx.x;
    const u64 c0 = chunk * FX_CHUNK;
    if (threadIdx.x < 8) red[threadIdx.x] = 0;
    __syncthreads();
    u32 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 state = 0;     // FASTQ: newlines so far in the chunk; FASTA: type of the current line (T_NONE: as entered)
    // (the next row's 64 bytes are requested before this row is analysed: one HBM round trip per row hidden)
    Lane64 d_next = c0 < n ? load_lane(text, n, c0 + threadIdx.x * FX_LANE) : Lane64{};
    for (u32 row = 0; row < FX_ROWS; ++row) {
        const u64 p = c0 + (u64)row * FX_ROW + threadIdx.x * FX_LANE;
        if (c0 + (u64)row * FX_ROW >= n) break;
        const Lane64 d = d_next;
        if (row + 1u < FX_ROWS && c0 + (u64)(row + 1u) * FX_ROW < n) d_next = load_lane(text, n, p + FX_ROW);
        if constexpr (!FASTA) {
            const FqLane a = fq_analyse(d);
            const u32 nl = pc64(a.nl);
            u32 tot;
            const u32 incl = block_scan_incl(nl, OpAdd32{}, tmp32, tot);
            const u32 ph = (state + incl - nl) & 3u;            // line number (relative to the chunk) at the lane's first byte
#pragma unroll
            for (u32 j = 0; j < 4; ++j) {
                acc[(ph + j) & 3u] += pc64(a.keep & a.cls[j]);
                acc[4u + ((ph + j + 1u) & 3u)] += pc64(a.start & a.cls[j]);   // the line after a newline on line j
            }
            state += tot;
        } else {
            const FaLane a = fa_analyse(d);
            u32 tot;
            const u32 key = a.def ? (((threadIdx.x + 1u) << 2) | a.def) : 0u;
            const u32 incl = block_scan_incl(key, OpMax32{}, tmp32, tot);
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
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        u32 v = acc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&red[q], v);
    }
    __syncthreads();
    u32* s = summ + chunk * FX_SUM_WORDS;
    if (threadIdx.x < 8) s[threadIdx.x] = red[threadIdx.x];
    if (threadIdx.x == 8) s[8] = state;
}

// pass 2: one block; a thread walks FX_CPT consecutive chunk summaries, the block scans the per-thread aggregates.
// prefix[chunk] = {entry state, first output byte, first record index}; totals[0..1] = reads, bases
constexpr u32 FX_CPT = 16;
template <bool FASTA>
__global__ void __launch_bounds__(1024)
fastx_scan_kernel(const u32* __restrict__ summ, u64 n_chunks, u64* __restrict__ prefix, unsigned long long* __restrict__ totals) {
    __shared__ u64 tmp64[1024 / 64];
    __shared__ u64 keys[1024];
    u64 state = FASTA ? T_HDR : 0;           // the text starts with a header line ('>' checked by the host) / line 0
    u64 out_pos = 0, rec = FASTA ? 1 : 0;    // FASTA: record 0 opens at byte 0 (no newline announces it)
    // what a chunk does to the state / emits when entered in state `in`
    auto next_state = [](u64 in, const u32* s) -> u64 { return FASTA ? (s[8] ? (u64)s[8] : in) : ((in + s[8]) & 3u); };
    auto kept_of = [](u64 in, const u32* s) -> u64 {
        return FASTA ? (u64)s[0] + (in == T_SEQ ? s[1] : 0u) : (u64)s[(1u - (u32)in) & 3u];   // file lines 1 (mod 4) are lines 1 - in of the chunk
    };
    auto recs_of = [](u64 in, const u32* s) -> u64 { return FASTA ? (u64)s[2] : (u64)s[4u + ((1u - (u32)in) & 3u)]; };
    for (u64 base = 0; base < n_chunks; base += 1024ull * FX_CPT) {
        const u64 c_lo = base + (u64)threadIdx.x * FX_CPT;
        const u64 c_hi = c_lo + FX_CPT < n_chunks ? c_lo + FX_CPT : n_chunks;
        // (1) the thread's effect on the state -> its entry state
        u64 in;
        u64 tot;
        if constexpr (!FASTA) {
            u64 nl = 0;
            for (u64 c = c_lo; c < c_hi; ++c) nl += summ[c * FX_SUM_WORDS + 8];
            const u64 incl = block_scan_incl(nl, OpAdd64{}, tmp64, tot);
            in = (state + incl - nl) & 3u;
            state = (state + tot) & 3u;
        } else {
            u64 def = 0;
            for (u64 c = c_lo; c < c_hi; ++c) { const u32 x = summ[c * FX_SUM_WORDS + 8]; def = x ? x : def; }
            const u64 key = def ? (((u64)(threadIdx.x + 1u) << 2) | def) : 0;
            const u64 incl = block_scan_incl(key, OpMax64{}, tmp64, tot);
            __syncthreads();
            keys[threadIdx.x] = incl;
            __syncthreads();
            const u64 prev = threadIdx.x ? keys[threadIdx.x - 1u] : 0;
            in = prev ? (prev & 3u) : state;
            if (tot) state = tot & 3u;
        }
        // (2) what the thread's chunks emit, given that
        u64 kept = 0, recs = 0, st = in;
        for (u64 c = c_lo; c < c_hi; ++c) {
            const u32* s = summ + c * FX_SUM_WORDS;
            kept += kept_of(st, s);
            recs += recs_of(st, s);
            st = next_state(st, s);
        }
        u64 tk, tr;
        const u64 ik = block_scan_incl(kept, OpAdd64{}, tmp64, tk);
        const u64 ir = block_scan_incl(recs, OpAdd64{}, tmp64, tr);
        // (3) per-chunk prefixes
        u64 o = out_pos + ik - kept, r = rec + ir - recs;
        st = in;
        for (u64 c = c_lo; c < c_hi; ++c) {
            const u32* s = summ + c * FX_SUM_WORDS;
            u64* pf = prefix + c * FX_PFX_WORDS;
            pf[0] = st;
            pf[1] = o;
            pf[2] = r;
            o += kept_of(st, s);
            r += recs_of(st, s);
            st = next_state(st, s);
        }
        out_pos += tk;
        rec += tr;
    }
    if (threadIdx.x == 0) {
        totals[0] = rec;
        totals[1] = out_pos;
    }
}

// pass 3
template <bool FASTA>
__global__ void __launch_bounds__(FX_THREADS)
fastx_emit_kernel(const uint8_t* __restrict__ text, u64 n, const u64* __restrict__ prefix, uint8_t* __restrict__ bases,
                  u64* __restrict__ offsets) {
    __shared__ u32 tmp32[FX_THREADS / 64];
    __shared__ u32 wave_last[FX_THREADS / 64];
    const u64 chunk = blockIdx.x;
    const u64 c0 = chunk * FX_CHUNK;
    const u64* pf = prefix + chunk * FX_PFX_WORDS;
    u32 state = (u32)pf[0];
    u64 out_pos = pf[1], rec = pf[2];
    if (chunk == 0 && threadIdx.x == 0 && FASTA) offsets[0] = 0;     // record 0 (see fastx_scan_kernel)
    // (the next row's 64 bytes are requested before this row is analysed: one HBM round trip per row hidden)
    Lane64 d_next = c0 < n ? load_lane(text, n, c0 + threadIdx.x * FX_LANE) : Lane64{};
    for (u32 row = 0; row < FX_ROWS; ++row) {
        const u64 p = c0 + (u64)row * FX_ROW + threadIdx.x * FX_LANE;
        if (c0 + (u64)row * FX_ROW >= n) break;
        const Lane64 d = d_next;
        if (row + 1u < FX_ROWS && c0 + (u64)(row + 1u) * FX_ROW < n) d_next = load_lane(text, n, p + FX_ROW);
        u64 ks, rs;       // bytes to emit; newlines after which a read begins
        u32 tot;
        if constexpr (!FASTA) {
            const FqLane a = fq_analyse(d);
            const u32 nl = pc64(a.nl);
            const u32 incl = block_scan_incl(nl, OpAdd32{}, tmp32, tot);
            const u32 in = (state + incl - nl) & 3u;
            state = (state + tot) & 3u;
            const u32 j = (1u - in) & 3u, jm = (0u - in) & 3u;   // the lane's line numbers that are read lines / header lines
            ks = a.keep & (j == 0 ? a.cls[0] : j == 1 ? a.cls[1] : j == 2 ? a.cls[2] : a.cls[3]);
            rs = a.start & (jm == 0 ? a.cls[0] : jm == 1 ? a.cls[1] : jm == 2 ? a.cls[2] : a.cls[3]);
        } else {
            const FaLane a = fa_analyse(d);
            const u32 key = a.def ? (((threadIdx.x + 1u) << 2) | a.def) : 0u;
            const u32 incl = block_scan_incl(key, OpMax32{}, tmp32, tot);
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
        const u32 ikr = block_scan_incl(kept | (recs << 16), OpAdd32{}, tmp32, tkr);
        const u64 o = out_pos + ((ikr & 0xFFFFu) - kept);
        u64 r = rec + ((ikr >> 16) - recs);
        out_pos += tkr & 0xFFFFu;
        rec += tkr >> 16;
        if (ks) emit_bytes(d, ks, bases + o);
        while (rs) {      // the read that begins after newline q starts at the output position of the bytes kept so far
            const u32 q = (u32)__builtin_ctzll(rs);
            rs &= rs - 1ull;
            offsets[r++] = o + pc64(ks & ((1ull << q) - 1ull));
        }
    }
}

__global__ void fastx_last_offset_kernel(const unsigned long long* __restrict__ totals, u64* __restrict__ offsets) {
    offsets[totals[0]] = totals[1];
}

size_t fastx_scratch_bytes(u64 n_bytes) {
    const u64 n_chunks = (n_bytes + FX_CHUNK - 1u) / FX_CHUNK;
    return (size_t)(n_chunks * (FX_SUM_WORDS * 4u + FX_PFX_WORDS * 8u) + 64u);
}

// counts: passes 1 and 2; totals (device) receives {n_reads, n_bases}
hipError_t launch_fastx_count(const uint8_t* text, u64 n, bool fasta, void* scratch, unsigned long long* totals, hipStream_t st) {
    const u64 n_chunks = (n + FX_CHUNK - 1u) / FX_CHUNK;
    u64* prefix = static_cast<u64*>(scratch);
    u32* summ = reinterpret_cast<u32*>(prefix + n_chunks * FX_PFX_WORDS);
    if (fasta) {
        hipLaunchKernelGGL(fastx_summarise_kernel<true>, dim3((unsigned)n_chunks), dim3(FX_THREADS), 0, st, text, n, summ);
        hipLaunchKernelGGL(fastx_scan_kernel<true>, dim3(1), dim3(1024), 0, st, summ, n_chunks, prefix, totals);
    } else {
        hipLaunchKernelGGL(fastx_summarise_kernel<false>, dim3((unsigned)n_chunks), dim3(FX_THREADS), 0, st, text, n, summ);
        hipLaunchKernelGGL(fastx_scan_kernel<false>, dim3(1), dim3(1024), 0, st, summ, n_chunks, prefix, totals);
    }
    return hipGetLastError();
}

// pass 3 (after launch_fastx_count on the same scratch)
hipError_t launch_fastx_emit(const uint8_t* text, u64 n, bool fasta, const void* scratch, const unsigned long long* totals,
                             uint8_t* bases, u64* offsets, hipStream_t st) {
    const u64 n_chunks = (n + FX_CHUNK - 1u) / FX_CHUNK;
    const u64* prefix = static_cast<const u64*>(scratch);
    if (fasta)
        hipLaunchKernelGGL(fastx_emit_kernel<true>, dim3((unsigned)n_chunks), dim3(FX_THREADS), 0, st, text, n, prefix, bases, offsets);
    else
        hipLaunchKernelGGL(fastx_emit_kernel<false>, dim3((unsigned)n_chunks), dim3(FX_THREADS), 0, st, text, n, prefix, bases, offsets);
    hipLaunchKernelGGL(fastx_last_offset_kernel, dim3(1), dim3(1), 0, st, totals, offsets);
    return hipGetLastError();
}

}  // namespace kmx
