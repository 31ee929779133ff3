// kmx_hist_part.h -- pass 1 of the partitioned bucket histogram: the sink that stages bucket ids in per-partition rings and writes
// them out in 64-byte rows (SinkHistPartT), its parameters, and the dispatch over the scan kernel's instantiations.  Shared by
// kmx_hist.hip (16-bit entries: 2^15..2^22 buckets; the pass between the two levels reuses the sink) and kmx_hist32.hip (32-bit
// entries: the first level of 2^23..2^28 buckets) -- two translation units so that the two sets of scan-kernel instantiations
// compile in parallel (together they took 3.6 minutes, twice the rest of the library).
#pragma once
// (the 64-byte rows of pass 1 are plain stores: with the nt hint they measured -0.2 ... +2 %, inside the noise)
#include "kmx_scan_kernel.h"

namespace kmx {

// the launch hook of launch_one (grid known -> size the segments, fetch the scratch) as ONE type for every caller: a
// trampoline over whatever callable the launcher holds
struct HistPartPre {
    bool (*call)(void*, u64);
    void* obj;
    bool operator()(u64 grid) const { return call(obj, grid); }
};
template <class F>
static HistPartPre make_hist_pre(F& f) {
    return HistPartPre{[](void* o, u64 g) -> bool { return (*static_cast<F*>(o))(g); }, &f};
}

struct HistPartParams {
    u64* counts;
    u32 hasher, hk, log2_buckets;
    void* stream;       // [n_waves][64][cap] ids of E bytes each (E = uint16_t: the low 16 bits of the bucket; u32: the bucket)
    u32* seg_len;       // [n_waves][64]
    u32 cap;            // entries per (wave, partition) segment, multiple of 64
};
// MODE (how the hash of a window comes about, fixed at compile time: three uniform branches per window otherwise):
//   0 LexHasher with hasher_k == k: hash = the 2k-bit complement of the LARGER of fw / rc (kmx_device.h lex_hash: the
//     reversed groups of the canonical word are the complement of the other strand) -- no hash arithmetic at all;
//   1 identity: hash = the smaller of the two;   2 LexHasher with another hasher_k.
// E (round 3): the stream's entry type.  uint16_t: up to 16 low bits per id (2^15..2^22 buckets).  u32: the whole bucket --
// the first level of the TWO-level partition of 2^23..2^28 buckets (hist_repartition_kernel splits every partition's u32 stream
// once more, by the next six bits, into uint16_t streams).  A ring is 128 bytes either way: 64 or 32 entries.
template <int MODE, typename E = uint16_t>
struct SinkHistPartT {
    static_assert(sizeof(E) == 2 || sizeof(E) == 4, "stream entries: uint16_t or u32");
    static constexpr u32 NP = 64, ROW = 128u / (u32)sizeof(E);   // partitions; ring entries per partition; rows of ROW/2 ids (64 bytes) leave together
    static constexpr u32 HALF = ROW / 2u;
    static constexpr u32 ESH = sizeof(E) == 2 ? 1u : 2u;         // log2 of the entry size
    static constexpr u32 EPL = 16u / (u32)sizeof(E);             // entries per 16-byte piece of a row
    // LDS: per wave the {appended|written} words, the segment cursors and the rank -> ring bytes; the rings of the four waves
    // together at the end of the block's LDS, each wave's 8 KB at a multiple of 8 KB: the ring address of an id is then
    // (mix >> 26 | base >> 13 << 6) << 7 -- one v_alignbit_b32 with the wave's base in the high word -- plus the slot bytes
    static constexpr u32 kLdsDwordsPerWave = 2u * NP + NP / 4u;
    static constexpr u32 kBlockLdsAlign = 2048u;   // dwords (8 KB)
    static constexpr bool kRagged = true;
    static constexpr bool kMarksDirty = true;   // (a tile with an invalid byte: fast path + marks, kmx_scan_kernel.h; launch_hist_uniform sweeps behind every scan)   // (ragged reads come window by window through fast(): no batches)
    static constexpr u32 kRingDwords = 4u * NP * 32u;   // 4 waves x 64 rings x 128 bytes
    static u32 block_lds_dwords(const HistPartParams&) { return kRingDwords; }
    HistPartParams p;
    E* ring;           // [NP][ROW]
    u32* word;         // [NP] appended (mod 2^16) << 16 | written out (mod 2^16)
    u32* cur;          // [NP] ids already in this wave's segment of the partition
    E* seg;            // this wave's [NP][cap] segments
    u64 maskk;
    u32 k, lane, lowbits;
    u32 shift_b, ring_hi, word_rel;   // 32 - log2_buckets; LDS byte address of ring[] >> 13; LDS byte address of word[] minus 4 * (ring_hi << 6)
    __device__ SinkHistPartT(const HistPartParams& p_, u32 k_, u32, u32* lds, u32 lane_, u32* block_lds, u32 tid)
        : p(p_), ring(reinterpret_cast<E*>(block_lds + (tid >> 6) * (NP * 32u))), word(lds), cur(lds + NP),
          maskk(mask2k(k_)), k(k_), lane(lane_), lowbits(p_.log2_buckets - 6u) {
        shift_b = 32u - p.log2_buckets;
        typedef u32 __attribute__((address_space(3))) * lds_u32p;
        const u32 ring_lds = (u32)(uintptr_t)(lds_u32p) reinterpret_cast<u32*>(ring);
        if (ring_lds & 8191u) __builtin_trap();   // (the launcher aligns the block region; dynamic LDS starts at 0)
        ring_hi = (u32)__builtin_amdgcn_readfirstlane(ring_lds >> 13);
        word_rel = (u32)(uintptr_t)(lds_u32p)word - ((ring_hi << 6) << 2);
        const u64 wave = (u64)blockIdx.x * 4u + (threadIdx.x >> 6);
        seg = static_cast<E*>(p.stream) + wave * NP * (u64)p.cap;
        word[lane] = 0;
        cur[lane] = 0;
        wave_sync();
    }
    __device__ __forceinline__ void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // MODE 0: the scan kernel hands its windows over COMPLEMENTED (fw ^ mask, rc ^ mask: it builds them from complemented
    // source words, which costs nothing), and the hash -- the complement of the larger strand -- is the smaller of the two as
    // they come.  (The rolled paths -- slow(), tile_slow_emit() -- pass the words themselves.)
    static constexpr bool kComplement = MODE == 0;
    // the 32-bit mix whose top log2_buckets bits are the bucket (bucket_of, kmx_device.h); COMPL: complemented inputs
    template <bool COMPL = false>
    __device__ __forceinline__ u32 mix_of_window(u64 fw, u64 rc) const {
        u64 h;
        if constexpr (MODE == 0 && COMPL) h = fw < rc ? fw : rc;
        else if constexpr (MODE == 0) h = (fw < rc ? rc : fw) ^ maskk;
        else if constexpr (MODE == 1) h = fw < rc ? fw : rc;
        else h = lex_hash(fw < rc ? fw : rc, p.hk);
        return bucket_mix((u32)h, (u32)(h >> 32));
    }
    __device__ __forceinline__ u32 bucket_of_window(u64 fw, u64 rc) const { return mix_of_window(fw, rc) >> shift_b; }
    // the slot of an id in its partition's ring: ONE returning LDS atomic
    __device__ __forceinline__ u32 take_slot(u32 bucket) { return atomicAdd(&word[bucket >> lowbits], 0x10000u); }
    __device__ __forceinline__ void place(u32 bucket, u32 w) {
        const u32 q = bucket >> lowbits;
        const u32 slot = w >> 16;
        if (((slot - w) & 0xFFFFu) < ROW) ring[q * ROW + (slot & (ROW - 1u))] = (E)(bucket & ((1u << lowbits) - 1u));
        else {   // ring full: take the slot back (every slot handed out past the ring is, so the count ends exact) and divert
            atomicSub(&word[q], 0x10000u);
            atomicAdd((unsigned long long*)&p.counts[bucket], 1ull);
        }
    }
    __device__ __forceinline__ void emit(u64 fw, u64 rc) {
        const u32 bucket = bucket_of_window(fw, rc);
        place(bucket, take_slot(bucket));
    }
    // The windows of an unrolled block are consumed NB at a time: their slot requests go out back to back and are
    // waited for once.  One at a time, every window paid the LDS round trip of its atomic before its ring store could be
    // addressed (and a branch on the answer keeps hipcc from overlapping them): the waves of pass 1 sat in s_waitcnt for
    // 47 % of their cycles.
    static constexpr bool kBatch16 = true;
    static constexpr int kWaves = 3;   // (LDS allows three blocks per CU: keep the registers inside 168)
    static constexpr int NB = 8;   // windows whose slot requests are in flight together (divides 16)
    u32 pend[NB];   // the mixes of the windows collected so far
    // The returned word is {appended : 16 | written out : 16} with written out in {0, HALF} and appended < 2 ROW + 64 (flush_rows
    // keeps them small: no 16-bit wrap to mask), so "staged before me" is one sub-dword subtract, the ring byte offset
    // 2 * (appended mod ROW) is the 7-bit field at bit 15, and ONE test per batch (an OR over the staged counts) tells
    // whether any of its ids found its ring full -- then, and only then, the batch takes the id-by-id path with the
    // diversion to the global table.
    __device__ __forceinline__ void fast_slot(int s, u64 fw, u64 rc) { push_mix(s, mix_of_window<kComplement>(fw, rc)); }
    // (also the entry of hist_repartition_kernel: an id whose mix is already known)
    __device__ __forceinline__ void push_mix(int s, u32 mix) {
        static_assert(ROW * sizeof(E) == 128, "ring addressing below: 128 bytes per ring");
        pend[s % NB] = mix;
        if (s % NB == NB - 1) {
            typedef u32 __attribute__((address_space(3))) * lds_u32p;
            typedef E __attribute__((address_space(3))) * lds_u16p;
            u32 w[NB];
            u32 qb[NB];   // ring base >> 7: partition | the wave's 8 KB index << 6
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                qb[j] = __builtin_amdgcn_alignbit(ring_hi, pend[j], 26);
                const u32 a = word_rel + (qb[j] << 2);
                w[j] = __hip_atomic_fetch_add((lds_u32p)(uintptr_t)a, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            u32 over = 0;
#pragma unroll
            for (int j = 0; j < NB; ++j) over |= (w[j] >> 16) - (w[j] & 0xFFFFu);
            if (__builtin_expect(__any((over & ~(ROW - 1u)) != 0u), 0)) {
#pragma unroll
                for (int j = 0; j < NB; ++j) place(pend[j] >> shift_b, w[j]);
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    // byte offset of the slot in its ring, sizeof(E) * (appended mod ROW): the 7-bit field at bit 16 - ESH (the bits
                    // it takes from the written-out half of the word are zero: written out is 0 or HALF)
                    const u32 a = (qb[j] << 7) + __builtin_amdgcn_ubfe(w[j], 16u - ESH, 7);
                    *(lds_u16p)(uintptr_t)a = (E)(pend[j] >> shift_b);   // (the bits above lowbits belong to the partition: the next pass masks them off)
                }
            }
            // u32 entries: a ring holds 32, and a block of 16 windows adds 16 +- 4 to one that may hold 15 already -- every
            // other block a ring overflowed and its ids took the global-atomic path (pass 1 ran 2.3x slower than with 16-bit
            // entries).  Draining after every batch of 8 keeps the staged count under 32.
            if constexpr (sizeof(E) == 4) {
                if (s < 15) flush_rows();
            }
        }
    }
    // ids staged and not yet written out
    static __device__ __forceinline__ u32 staged(u32 w) { return ((w >> 16) - w) & 0xFFFFu; }
    // The whole wave: every ring with a full half row (HALF ids) writes it out.  SIXTEEN rings per round, four lanes
    // (16 bytes each) per ring; which ring a group takes comes from a rank table (ring -> its rank among the rings to
    // flush, by v_mbcnt; rank -> ring through 64 bytes of LDS), not from a scalar walk over the mask: with ~32 of the 64
    // rings due after every block of 16 windows, the first version's rounds of four rings -- eight per block, each with its
    // scalar ctz loop, a quarter-wave busy and a wave_sync -- cost more than the 16 windows they followed
    // (pass 1 at 2^20 buckets: 22 -> see DESIGN 4.3).
    __device__ __forceinline__ void flush_rows() {
        static_assert(HALF * sizeof(E) == 64, "row flush: half a ring = 64 bytes = 4 lanes x 16 bytes");
        wave_sync();
        const bool due = staged(word[lane]) >= HALF;
        const u64 m = __ballot(due);
        if (m == 0) return;
        const u32 n_due = (u32)__builtin_popcountll(m);
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
        uint8_t* order = reinterpret_cast<uint8_t*>(cur + NP);   // [NP] ring of rank r
        if (due) order[rank] = (uint8_t)lane;
        wave_sync();
        const u32 grp = lane >> 2, l4 = lane & 3u;
        for (u32 base = 0; base < n_due; base += 16u) {
            const u32 r = base + grp;
            if (r < n_due) {
                const u32 q = order[r];
                const u32 w = word[q];
                const u32 pos = cur[q];
                const u32 half = w & HALF;   // written-out count is a multiple of HALF: the row starts at ring entry 0 or HALF
                const uint4 v = *reinterpret_cast<const uint4*>(ring + q * ROW + half + EPL * l4);
                if (pos + HALF <= p.cap) {
                    *reinterpret_cast<uint4*>(seg + (u64)q * p.cap + pos + EPL * l4) = v;
                } else {   // segment full: the ids go to the global table
                    const u32 hi = q << lowbits, idm = (1u << lowbits) - 1u;
                    const u32 vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (u32 i = 0; i < 4; ++i) {
                        atomicAdd((unsigned long long*)&p.counts[hi | (vv[i] & idm)], 1ull);
                        if constexpr (sizeof(E) == 2) atomicAdd((unsigned long long*)&p.counts[hi | ((vv[i] >> 16) & idm)], 1ull);
                    }
                }
                if (l4 == 0) {
                    if (pos + HALF <= p.cap) cur[q] = pos + HALF;
                    // written out: 0 -> HALF; HALF -> 0 with a whole ring taken off the appended count (the same slot mod ROW)
                    word[q] = half ? w - HALF - (ROW << 16) : w + HALF;
                }
            }
        }
        wave_sync();
    }
    // (a full unrolled block arrives as 16 fast_slot() calls -- uniform reads only, so every lane has all 16; the partial
    // last block of a read arrives through fast(), one window at a time)
    __device__ __forceinline__ void block_done(u64, u32, u32) { flush_rows(); }
    __device__ __forceinline__ void fast(u32, u64 fw, u64 rc) {   // (from the scan kernel's window(): complemented like fast_slot's)
        const u32 bucket = mix_of_window<kComplement>(fw, rc) >> shift_b;
        place(bucket, take_slot(bucket));
    }
    __device__ __forceinline__ void slow(u32, u64 fw, u64 rc) { emit(fw, rc); }
    __device__ __forceinline__ void begin_read(u64) {}
    __device__ __forceinline__ void slow_block(u32) { flush_rows(); }   // a rolled tile: 16 more windows per read, wave converged: drain the rings
    __device__ __forceinline__ void tile_slow_begin(u64 read) { begin_read(read); }
    __device__ __forceinline__ void tile_slow_emit(u32 pos, u64 fw, u64 rc) { slow(pos, fw, rc); }
    __device__ __forceinline__ void tile_slow_end() { end_read(); }
    // slow path: up to W ids per lane since the last flush (what does not fit the rings went to the global table).
    // The final partial tile calls this with some lanes masked off; flush_rows needs the whole wave, so it waits.
    __device__ __forceinline__ void end_read() {
        if (__ballot(1) == ~0ull) {
            flush_rows();
            flush_rows();   // a ring can hold two full half rows
        }
    }
    __device__ __forceinline__ void tile_fast_done(u32) {}
    __device__ __forceinline__ void finish(const HistPartParams&) {
        flush_rows();
        flush_rows();
        // the tails (< HALF ids per ring), one ring at a time
        for (u32 q = 0; q < NP; ++q) {
            const u32 w = word[q];
            const u32 n = staged(w), pos = cur[q];
            if (lane < n) {
                const u32 e = ring[q * ROW + ((w + lane) & (ROW - 1u))];
                if (pos + n <= p.cap) seg[(u64)q * p.cap + pos + lane] = (E)e;
                else atomicAdd((unsigned long long*)&p.counts[(q << lowbits) | (e & ((1u << lowbits) - 1u))], 1ull);
            }
            wave_sync();
            if (lane == 0 && pos + n <= p.cap) cur[q] = pos + n;
            wave_sync();
        }
        const u64 wave = (u64)blockIdx.x * 4u + (threadIdx.x >> 6);
        p.seg_len[wave * NP + lane] = cur[lane];
    }
};

// pass 2: block (partition q, group g) adds the segments of the waves w == g (mod gridDim.y) into an LDS table
// SUB_BITS = 1 (2^22 buckets: a partition's 2^16-entry table does not fit the LDS): blockIdx.z picks the half of the partition's

template <typename SinkHistPart, typename Pre, bool RAGGED>
static hipError_t dispatch_part_mode(const uint8_t* bases, u64 n_reads, u32 L, u32 k, HistPartParams& p, unsigned long long* queue,
                                     int n_cu, hipStream_t stream, Pre pre, const u64* offsets) {
    const bool big = L > 160 || (RAGGED && L == 0);
#define KMX_PART(NW, V, DW) launch_one<NW, V, DW, SinkHistPart, HistPartParams, Pre, RAGGED>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets)
    if (k <= 16) return big ? KMX_PART(16, 1, 1) : KMX_PART(10, 1, 1);
    if (k == 17) return big ? KMX_PART(16, 1, 2) : KMX_PART(10, 1, 2);
    return big ? KMX_PART(16, 2, 2) : KMX_PART(10, 2, 2);
#undef KMX_PART
}

template <typename Pre, typename E = uint16_t>
static hipError_t dispatch_part(const uint8_t* bases, u64 n_reads, u32 L, u32 k, HistPartParams& p, unsigned long long* queue,
                                int n_cu, hipStream_t stream, Pre pre, const u64* offsets) {
    const int mode = p.hasher != KMX_HASH_LEX ? 1 : p.hk == k ? 0 : 2;
    if constexpr (sizeof(E) == 4) {   // first level of the two-level partition (2^23..2^28 buckets)
        if (offsets) {
            if (mode == 0) return dispatch_part_mode<SinkHistPartT<0, u32>, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
            if (mode == 1) return dispatch_part_mode<SinkHistPartT<1, u32>, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
            return dispatch_part_mode<SinkHistPartT<2, u32>, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
        }
        if (mode == 0) return dispatch_part_mode<SinkHistPartT<0, u32>, Pre, false>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, nullptr);
        if (mode == 1) return dispatch_part_mode<SinkHistPartT<1, u32>, Pre, false>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, nullptr);
        return dispatch_part_mode<SinkHistPartT<2, u32>, Pre, false>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, nullptr);
    }
    if (offsets) {
        if (mode == 0) return dispatch_part_mode<SinkHistPartT<0>, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
        if (mode == 1) return dispatch_part_mode<SinkHistPartT<1>, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
        return dispatch_part_mode<SinkHistPartT<2>, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
    }
    if (mode == 0) return dispatch_part_mode<SinkHistPartT<0>, Pre, false>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, nullptr);
    if (mode == 1) return dispatch_part_mode<SinkHistPartT<1>, Pre, false>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, nullptr);
    return dispatch_part_mode<SinkHistPartT<2>, Pre, false>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, nullptr);
}

// Histogram over uniform or ragged reads.  2^b <= 2^14: block-private LDS tables (SinkHistLds).  2^15..2^22: two passes through
// 64 partitions (SinkHistPart + hist_part_reduce_kernel) in chunks of reads sized to `scratch_budget` bytes of
// caller-provided scratch (`get_scratch(user, bytes)` returns a device buffer of at least `bytes`, or nullptr).
// 2^23..2^28: the same with a second level of 64 partitions in between (hist_repartition_kernel).

// first level of the two-level partition (u32 entries): instantiated in kmx_hist32.hip
hipError_t dispatch_part_u32(const uint8_t* bases, u64 n_reads, u32 L, u32 k, HistPartParams& p, unsigned long long* queue,
                             int n_cu, hipStream_t stream, HistPartPre pre, const u64* offsets);

}  // namespace kmx
