// kmx_hist.hip -- bucket histograms of hash(canonical k-mer) over the word-domain scan kernel (kmx_scan_kernel.h): the sinks
// (device-scope atomics, block-private LDS tables, the two-pass partitioned histogram), pass 2 of the latter, and
// launch_hist_uniform.  BUILD-DEFINED (the reference has no histogram): include/kmx.h kmx_histogram, oracle kmo_histogram.
#include "kmx_scan_kernel.h"

namespace kmx {

struct HistParams {
    u64* counts;
    u32 hasher, hk, log2_buckets;
};
// d_counts[bucket(hash(canonical k-mer))] += 1 with device-scope u64 atomics.  Measured ~24 G atomics/s on MI355X
// independent of the bucket count (2^12..2^26) and of the atomic scope (XCD-private copies updated with
// workgroup-scope atomics ran at the same rate), i.e. bound by the atomic issue rate, not by contention.
struct SinkHist {
    u64* counts;
    u64 maskk;
    u32 hasher, hk, k, b;
    static constexpr u32 kLdsDwordsPerWave = 0;
    __device__ __forceinline__ void block_done(u64, u32, u32) {}
    static constexpr bool kRagged = true;
    static u32 block_lds_dwords(const HistParams&) { return 0; }
    __device__ SinkHist(const HistParams& p, u32 k_, u32, u32*, u32, u32*, u32)
        : counts(p.counts), maskk(mask2k(k_)), hasher(p.hasher), hk(p.hk), k(k_), b(p.log2_buckets) {}
    __device__ __forceinline__ void emit(u64 fw, u64 rc) {
        const u64 canon = fw < rc ? fw : rc;
        u64 h;
        if (hasher == KMX_HASH_LEX) h = (hk == k) ? (maskk ^ fw ^ rc ^ canon) : lex_hash(canon, hk);
        else h = canon;  // identity: write_u64(data), hash.rs:4-8
        atomicAdd((unsigned long long*)&counts[bucket_of(h, b)], 1ull);
    }
    __device__ __forceinline__ void fast(u32, u64 fw, u64 rc) { emit(fw, rc); }
    __device__ __forceinline__ void slow(u32, u64 fw, u64 rc) { emit(fw, rc); }
    __device__ __forceinline__ void begin_read(u64) {}
    __device__ __forceinline__ void slow_block(u32) {}   // a rolled tile: the wave has completed 16 more windows per read
    __device__ __forceinline__ void tile_slow_begin(u64 read) { begin_read(read); }
    __device__ __forceinline__ void tile_slow_emit(u32 pos, u64 fw, u64 rc) { slow(pos, fw, rc); }
    __device__ __forceinline__ void tile_slow_end() { end_read(); }
    __device__ __forceinline__ void end_read() {}
    __device__ __forceinline__ void tile_fast_done(u32) {}
    __device__ __forceinline__ void finish(const HistParams&) {}
};

// Histogram, 2^b <= 2^14 buckets: block-private u32 table in LDS (ds_add_u32, no return), merged into d_counts
// with one u64 atomic per non-empty bucket per block when the block retires.  The global-atomic sink above is bound
// by the atomic rate (24 G/s => 0.5 s per 1e8 reads); LDS atomics are not.
struct SinkHistLds {
    u64* counts;
    u32* tab;
    u64 maskk;
    u32 hasher, hk, k, b, tid;
    static constexpr u32 kLdsDwordsPerWave = 0;
    static constexpr bool kRagged = true;
    static u32 block_lds_dwords(const HistParams& p) { return 1u << p.log2_buckets; }
    __device__ SinkHistLds(const HistParams& p, u32 k_, u32, u32*, u32, u32* block_lds, u32 tid_)
        : counts(p.counts), tab(block_lds), maskk(mask2k(k_)), hasher(p.hasher), hk(p.hk), k(k_), b(p.log2_buckets), tid(tid_) {
        for (u32 j = tid; j < (1u << b); j += 256u) tab[j] = 0;
        __syncthreads();
    }
    __device__ __forceinline__ void block_done(u64, u32, u32) {}
    __device__ __forceinline__ void emit(u64 fw, u64 rc) {
        const u64 canon = fw < rc ? fw : rc;
        u64 h;
        if (hasher == KMX_HASH_LEX) h = (hk == k) ? (maskk ^ fw ^ rc ^ canon) : lex_hash(canon, hk);
        else h = canon;
        atomicAdd(&tab[(u32)bucket_of(h, b)], 1u);
    }
    __device__ __forceinline__ void fast(u32, u64 fw, u64 rc) { emit(fw, rc); }
    __device__ __forceinline__ void slow(u32, u64 fw, u64 rc) { emit(fw, rc); }
    __device__ __forceinline__ void begin_read(u64) {}
    __device__ __forceinline__ void slow_block(u32) {}   // a rolled tile: the wave has completed 16 more windows per read
    __device__ __forceinline__ void tile_slow_begin(u64 read) { begin_read(read); }
    __device__ __forceinline__ void tile_slow_emit(u32 pos, u64 fw, u64 rc) { slow(pos, fw, rc); }
    __device__ __forceinline__ void tile_slow_end() { end_read(); }
    __device__ __forceinline__ void end_read() {}
    __device__ __forceinline__ void tile_fast_done(u32) {}
    __device__ __forceinline__ void finish(const HistParams&) {
        __syncthreads();
        for (u32 j = tid; j < (1u << b); j += 256u) {
            const u32 c = tab[j];
            if (c) atomicAdd((unsigned long long*)&counts[j], (unsigned long long)c);
        }
    }
};

// Histogram, 2^15..2^22 buckets, pass 1 of 2: scatter the bucket ids into 64 partitions (the top 6 bits of the bucket).
// Every wave owns a private segment of every partition's stream, so no global cursor and no global atomic is
// involved.  The low b-6 bits of an id are staged in a ROW-entry ring per partition in the wave's LDS slice: ONE
// ds_add_rtn_u32 on a packed {entries appended : 16 | entries written out : 16} word returns the slot and tells
// whether the ring has room, one ds_write_b16 stores the id.  After every block of 16 windows the rings holding
// >= ROW/2 ids write one half row each to their segment, four partitions at a time (one per quarter-wave).
// Pass 2 (hist_part_reduce_kernel) builds each partition's 2^(b-6)-bucket table in LDS.  Ids that find their ring or
// their segment full (adversarial input: everything in one partition) go straight to the global table, so the result
// is exact for every input.
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
    static constexpr bool kRagged = true;   // (ragged reads come window by window through fast(): no batches)
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
#ifndef KMX_HIST_BATCH
#define KMX_HIST_BATCH 8
#endif
    static constexpr int NB = KMX_HIST_BATCH;   // windows whose slot requests are in flight together (divides 16)
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
// buckets this block counts; both halves read the whole id stream of the partition.
template <int THREADS, int SUB_BITS = 0>
__global__ void __launch_bounds__(THREADS)
hist_part_reduce_kernel(const uint16_t* __restrict__ stream, const u32* __restrict__ seg_len, u32 cap, u32 n_waves,
                        u32 log2_buckets, u64* __restrict__ counts, u64 top_stream_stride, u32 top_len_stride) {
    // blockIdx.z = (top partition T << SUB_BITS) | sub.  One level (2^15..2^22 buckets): T = 0.  Two levels (2^23..2^28): T = the
    // first-level partition whose 64 second-level streams this launch counts; log2_buckets is then the bucket bits BELOW
    // the first level (b - 6), the streams and lengths of T start T strides in, its counters at counts + (T << log2_buckets).
    extern __shared__ __attribute__((aligned(16))) u32 tab[];
    const u32 lowbits = log2_buckets - 6u, idm = (1u << lowbits) - 1u;
    const u32 tb = lowbits - (u32)SUB_BITS, nb = 1u << tb, sub = SUB_BITS ? (blockIdx.z & ((1u << SUB_BITS) - 1u)) : 0u;
    const u32 top = blockIdx.z >> SUB_BITS;
    stream += (u64)top * top_stream_stride;
    seg_len += (u64)top * top_len_stride;
    counts += (u64)top << log2_buckets;
    const u32 q = blockIdx.x;
    for (u32 j = threadIdx.x; j < nb; j += THREADS) tab[j] = 0;
    __syncthreads();
    // every WAVE of the block walks its own segments (w == its index mod the waves of the partition's blocks): a segment is
    // ~25 KB, too short for 512 threads to keep several loads each in flight
    const u32 wv = threadIdx.x >> 6, ln = threadIdx.x & 63u, nwv = THREADS / 64u;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    // (an id carries the low bits of its partition above bit lowbits: the scan writes the bucket's low 16 bits as they are)
    auto count1 = [&](u32 id) {
        id &= idm;
        if constexpr (SUB_BITS == 0) atomicAdd(&tab[id], 1u);
        else if ((id >> tb) == sub) atomicAdd(&tab[id & (nb - 1u)], 1u);
    };
    auto count8 = [&](const u32x4 v) {
        count1(v.x); count1(v.x >> 16); count1(v.y); count1(v.y >> 16);
        count1(v.z); count1(v.z >> 16); count1(v.w); count1(v.w >> 16);
    };
    const u32 w0 = blockIdx.y * nwv + wv, wstep = gridDim.y * nwv;
    u32 len_next = w0 < n_waves ? seg_len[(u64)w0 * 64u + q] : 0u;
    for (u32 w = w0; w < n_waves; w += wstep) {
        const u32 len = len_next;
        if (w + wstep < n_waves) len_next = seg_len[(u64)(w + wstep) * 64u + q];   // (one segment ahead)
        const uint16_t* __restrict__ sp = stream + ((u64)w * 64u + q) * (u64)cap;   // cap is a multiple of 64: 128-byte aligned
        const u32x4* __restrict__ sp8 = reinterpret_cast<const u32x4*>(sp);
        // four 16-byte loads per lane in flight: with one load per thread the pass ran at the latency of its loads (3.8 TB/s
        // of ids, the LDS 39 % busy)
        const u32 n16 = len / 8u;
        u32 i = ln;
        for (; i + 192u < n16; i += 256u) {
            const u32x4 v0 = __builtin_nontemporal_load(sp8 + i);
            const u32x4 v1 = __builtin_nontemporal_load(sp8 + i + 64u);
            const u32x4 v2 = __builtin_nontemporal_load(sp8 + i + 128u);
            const u32x4 v3 = __builtin_nontemporal_load(sp8 + i + 192u);
            count8(v0);
            count8(v1);
            count8(v2);
            count8(v3);
        }
        for (; i < n16; i += 64u) count8(__builtin_nontemporal_load(sp8 + i));
        for (u32 j = (len & ~7u) + ln; j < len; j += 64u) count1(sp[j]);
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < nb; j += THREADS) {
        const u32 c = tab[j];
        if (c) atomicAdd((unsigned long long*)&counts[((u64)q << lowbits) | ((u64)sub << tb) | j], (unsigned long long)c);
    }
}

// 2^23..2^28 buckets, the pass between: the u32 stream of first-level partition T (blockIdx.y; the ids of all waves of pass
// 1) is split by the NEXT six bits of the bucket into 64 uint16_t streams -- through the same rings and rows as pass 1 (the
// sink is reused as it is: an id is pushed as the 32-bit mix whose top bits are the bucket below the first level).
// Every wave owns a segment of each of T's 64 second-level streams: stream2[T][wave][64][cap2], seg_len2[T][wave][64].
__global__ void __launch_bounds__(256)
hist_repartition_kernel(const u32* __restrict__ stream1, const u32* __restrict__ seg_len1, u32 cap1, u32 n_waves1, u32 log2_buckets,
                        u64* __restrict__ counts, uint16_t* __restrict__ stream2, u32* __restrict__ seg_len2, u32 cap2) {
    typedef SinkHistPartT<1, uint16_t> Sink;
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 T = blockIdx.y, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32 n_waves2 = gridDim.x * 4u;
    const u32 low1 = log2_buckets - 6u;                       // bucket bits below the first level (17..22)
    HistPartParams p2{counts + ((u64)T << low1), KMX_HASH_IDENTITY, 0u, low1,
                      stream2 + (u64)T * n_waves2 * 64u * (u64)cap2, seg_len2 + (u64)T * n_waves2 * 64u, cap2};
    // LDS: the four waves' rings first (32 KB, 8 KB-aligned as the sink wants them), then each wave's words
    u32* const rings = lds;
    u32* const mine = lds + Sink::kRingDwords + wv * Sink::kLdsDwordsPerWave;
    Sink sink(p2, 0u, 0u, mine, lane, rings, threadIdx.x);
    const u32 up = 32u - low1;                                // id -> mix: the bucket below the first level in the top bits
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32 w0 = blockIdx.x * 4u + wv;
    for (u32 w = w0; w < n_waves1; w += n_waves2) {
        const u32 len = seg_len1[(u64)w * 64u + T];
        const u32* __restrict__ sp = stream1 + ((u64)w * 64u + T) * (u64)cap1;
        const u32x4* __restrict__ sp4 = reinterpret_cast<const u32x4*>(sp);
        const u32 n16 = len / 4u;                             // 16-byte pieces
        u32 i = 0;
        // 16 ids per lane and round (four loads in flight), then the rings holding a half row write it out -- the rhythm of
        // pass 1 (16 windows per read, then flush_rows)
        for (; i + 256u <= n16; i += 256u) {
            const u32x4 v0 = __builtin_nontemporal_load(sp4 + i + lane);
            const u32x4 v1 = __builtin_nontemporal_load(sp4 + i + 64u + lane);
            const u32x4 v2 = __builtin_nontemporal_load(sp4 + i + 128u + lane);
            const u32x4 v3 = __builtin_nontemporal_load(sp4 + i + 192u + lane);
            sink.push_mix(0, v0.x << up); sink.push_mix(1, v0.y << up); sink.push_mix(2, v0.z << up); sink.push_mix(3, v0.w << up);
            sink.push_mix(4, v1.x << up); sink.push_mix(5, v1.y << up); sink.push_mix(6, v1.z << up); sink.push_mix(7, v1.w << up);
            sink.push_mix(8, v2.x << up); sink.push_mix(9, v2.y << up); sink.push_mix(10, v2.z << up); sink.push_mix(11, v2.w << up);
            sink.push_mix(12, v3.x << up); sink.push_mix(13, v3.y << up); sink.push_mix(14, v3.z << up); sink.push_mix(15, v3.w << up);
            sink.flush_rows();
        }
        // the tail of the segment: id by id, a quarter of a round at a time
        for (u32 j = 4u * i; j < len; j += 256u) {
#pragma unroll
            for (u32 t = 0; t < 4u; ++t) {
                const u32 e = j + 64u * t + lane;
                if (e < len) {
                    const u32 bucket = (sp[e] << up) >> (32u - low1);
                    sink.place(bucket, sink.take_slot(bucket));
                }
            }
            sink.flush_rows();
        }
    }
    sink.finish(p2);
}

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
// Larger tables, or no scratch: device-scope u64 atomics (SinkHist).
hipError_t launch_hist_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 hasher, u32 hk, u32 log2_buckets,
                               u64* counts, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                               void* (*get_scratch)(void*, size_t), void* user, size_t scratch_budget, const u64* offsets) {
    *handled = offsets ? scan_domain_ragged(bases, L, k) : scan_domain(bases, n_reads, L, k);
    if (!*handled) return hipSuccess;
    const HistParams p{counts, hasher, hk, log2_buckets};
    if (log2_buckets <= 14u) return dispatch<SinkHistLds>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets);
    if (log2_buckets >= 23u && log2_buckets <= 28u && get_scratch != nullptr && n_reads >= 4096u) {
        // Two levels of 64 partitions each (round 3; these sizes took device atomics before: 0.5 s per 1e8 reads).  Pass 1 as
        // below but with the whole bucket per id (u32 entries); hist_repartition_kernel splits each partition's stream by the
        // next six bits into uint16_t streams; hist_part_reduce_kernel counts the 64 x 64 streams in LDS tables of 2^(b-12)
        // entries (2^16 at b = 28: two halves).  Scratch per window: 1.5 x (4 + 2) bytes.
        const u32 Lb = offsets ? (L ? L : 256u) : L;
        const u64 W = Lb >= k ? Lb - k + 1u : 1u;
        u64 chunk = scratch_budget / (10u * W);
        if (chunk > n_reads) chunk = n_reads;
        chunk &= ~63ull;
        if (chunk >= 4096u) {
            const u32 low1 = log2_buckets - 6u;
            for (u64 first = 0; first < n_reads; first += chunk) {
                const u64 n = n_reads - first < chunk ? n_reads - first : chunk;
                HistPartParams pp{counts, hasher, hk, log2_buckets, nullptr, nullptr, 0};
                u32 n_waves = 0, n_waves2 = 0, cap2 = 0;
                uint16_t* stream2 = nullptr;
                u32* seg_len2 = nullptr;
                const u32 gx2 = (u32)((n_cu * 4 + 63) / 64 > 0 ? (n_cu * 4 + 63) / 64 : 1);   // blocks per first-level partition in the pass between
                auto pre = [&](u64 grid) -> bool {
                    n_waves = (u32)(grid * 4u);
                    const u64 per_seg = (n * W * 3u / 2u) / ((u64)n_waves * 64u) + 256u;
                    pp.cap = (u32)((per_seg + 63u) & ~63ull);
                    if (pp.cap > (1u << 24)) return false;
                    n_waves2 = gx2 * 4u;
                    const u64 per_seg2 = (n * W * 3u / 2u) / (64ull * n_waves2 * 64u) + 256u;
                    cap2 = (u32)((per_seg2 + 63u) & ~63ull);
                    const size_t s1 = (size_t)n_waves * 64u * pp.cap * 4u, l1 = (size_t)n_waves * 64u * 4u;
                    const size_t s2 = (size_t)64u * n_waves2 * 64u * cap2 * 2u, l2 = (size_t)64u * n_waves2 * 64u * 4u;
                    char* buf = static_cast<char*>(get_scratch(user, s1 + l1 + s2 + l2));
                    if (!buf) return false;
                    pp.stream = buf;
                    pp.seg_len = reinterpret_cast<u32*>(buf + s1);
                    stream2 = reinterpret_cast<uint16_t*>(buf + s1 + l1);
                    seg_len2 = reinterpret_cast<u32*>(buf + s1 + l1 + s2);
                    return true;
                };
                if (first != 0) {
                    hipError_t e = hipMemsetAsync(queue, 0, 32 * 128, stream);
                    if (e != hipSuccess) return e;
                }
                const uint8_t* cb = offsets ? bases : bases + first * (u64)L;
                const u64* co = offsets ? offsets + first : nullptr;
                hipError_t e = dispatch_part<decltype(pre), u32>(cb, n, L, k, pp, queue, n_cu, stream, pre, co);
                if (e == hipErrorOutOfMemory) {   // no scratch: the atomic sink handles the rest
                    (void)hipGetLastError();
                    return dispatch<SinkHist>(cb, n_reads - first, L, k, p, queue, n_cu, stream, NoPre(), co);
                }
                if (e != hipSuccess) return e;
                typedef SinkHistPartT<1, uint16_t> Sink2;
                const HistPartParams dummy{};
                const size_t lds2 = ((size_t)Sink2::block_lds_dwords(dummy) + 4u * Sink2::kLdsDwordsPerWave) * 4u;
                hipLaunchKernelGGL(hist_repartition_kernel, dim3(gx2, 64), dim3(256), lds2, stream, static_cast<const u32*>(pp.stream), pp.seg_len,
                                   pp.cap, n_waves, log2_buckets, counts, stream2, seg_len2, cap2);
                e = hipGetLastError();
                if (e != hipSuccess) return e;
                const bool halves = low1 == 22u;   // 2^16 buckets per second-level partition: two blocks of 2^15 each
                const u32 nb_bytes = 4u << (low1 - 6u - (halves ? 1u : 0u));
                auto red = halves ? hist_part_reduce_kernel<512, 1> : hist_part_reduce_kernel<512, 0>;
                if (nb_bytes > 64u * 1024u) {
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(red), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_bytes);
                    if (e != hipSuccess) return e;
                }
                const u32 groups = n_waves2 < 4u ? n_waves2 : 4u;
                hipLaunchKernelGGL(red, dim3(64, groups, 64u * (halves ? 2u : 1u)), dim3(512), nb_bytes, stream, stream2, seg_len2, cap2, n_waves2,
                                   low1, counts, (u64)n_waves2 * 64u * (u64)cap2, n_waves2 * 64u);
                e = hipGetLastError();
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
    }
    if (log2_buckets <= 22u && get_scratch != nullptr && n_reads >= 4096u) {
        // windows per read the segments are sized for.  Ragged reads: from the caller's bound of the lengths (the frame's 256 if
        // there is none); a read that is longer after all only fills its wave's segments sooner, and what finds a segment full
        // goes to the global table (exact, slow).
        const u32 Lb = offsets ? (L ? L : 256u) : L;
        const u64 W = Lb >= k ? Lb - k + 1u : 1u;
        // scratch per read: 1.5x slack on 2 bytes per window, plus the fixed per-segment pad; chunk the reads to fit
        u64 chunk = scratch_budget / (3u * W);
        if (chunk > n_reads) chunk = n_reads;
        chunk &= ~63ull;
        if (chunk >= 4096u) {
            for (u64 first = 0; first < n_reads; first += chunk) {
                const u64 n = n_reads - first < chunk ? n_reads - first : chunk;
                HistPartParams pp{counts, hasher, hk, log2_buckets, nullptr, nullptr, 0};
                u32 n_waves = 0;
                auto pre = [&](u64 grid) -> bool {
                    n_waves = (u32)(grid * 4u);
                    const u64 per_seg = (n * W * 3u / 2u) / ((u64)n_waves * 64u) + 256u;
                    pp.cap = (u32)((per_seg + 63u) & ~63ull);
                    if (pp.cap > (1u << 24)) return false;   // 64 * cap must stay below 2^31 (SinkHistPart::dest)
                    const size_t stream_bytes = (size_t)n_waves * 64u * pp.cap * 2u;
                    const size_t len_bytes = (size_t)n_waves * 64u * 4u;
                    char* buf = static_cast<char*>(get_scratch(user, stream_bytes + len_bytes));
                    if (!buf) return false;
                    pp.stream = buf;
                    pp.seg_len = reinterpret_cast<u32*>(buf + stream_bytes);
                    return true;
                };
                if (first != 0) {
                    hipError_t e = hipMemsetAsync(queue, 0, 32 * 128, stream);
                    if (e != hipSuccess) return e;
                }
                // the hook fills pp through the reference captured above; dispatch takes its params by value, so hand it
                // a proxy that copies the finished pp at launch time
                // (ragged reads: the offsets are absolute, the chunk is a window into them)
                const uint8_t* cb = offsets ? bases : bases + first * (u64)L;
                const u64* co = offsets ? offsets + first : nullptr;
                hipError_t e = dispatch_part(cb, n, L, k, pp, queue, n_cu, stream, pre, co);
                if (e == hipErrorOutOfMemory) {   // no scratch: the atomic sink handles the rest
                    (void)hipGetLastError();
                    return dispatch<SinkHist>(cb, n_reads - first, L, k, p, queue, n_cu, stream, NoPre(), co);
                }
                if (e != hipSuccess) return e;
                const bool halves = log2_buckets == 22u;   // 2^16 buckets per partition: two blocks of 2^15 each
                const u32 nb_bytes = 4u << (log2_buckets - 6u - (halves ? 1u : 0u));
                auto red = halves ? hist_part_reduce_kernel<512, 1> : hist_part_reduce_kernel<512, 0>;
                if (nb_bytes > 64u * 1024u) {
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(red), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_bytes);
                    if (e != hipSuccess) return e;
                }
                const u32 groups = n_waves < 16u ? n_waves : 16u;
                hipLaunchKernelGGL(red, dim3(64, groups, halves ? 2 : 1), dim3(512), nb_bytes, stream, static_cast<const uint16_t*>(pp.stream), pp.seg_len,
                                   pp.cap, n_waves, log2_buckets, counts, (u64)0, 0u);
                e = hipGetLastError();
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
    }
    return dispatch<SinkHist>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets);
}

}  // namespace kmx
