// kmx_scan.hip -- K1/K2/K4 word-domain scan over uniform-length reads: fused encode + sliding window +
// reverse complement + canonical, feeding a SINK: reduce (summary), bucket histogram, or materialise
// (per-window fw / rc / canonical / flags).  Serves every (k, L) the bit-sliced kernel (kmx_bitslice.hip)
// is not instantiated for, and all histogram / materialise calls on uniform reads.
//
// Replaces, per read, the reference's streaming loop
//   CanonicalKmerIterator::find_next      src/naive_impl/canonical_kmer_iterator.rs:42-70
//   CanonicalKmer::append_base            src/naive_impl/canonical_kmer.rs:90-94
//   Kmer::append_base / prepend_base      src/naive_impl/kmer.rs:91-102
//   CanonicalKmer::get_canonical_word     src/naive_impl/canonical_kmer.rs:113-119
//   encode_binary_u8                      src/naive_impl/mod.rs:40-50
//   LexHasher::write_u64 / Hash for Kmer  src/naive_impl/hash.rs:4-8,60-71
//
// gfx950 design (one wave owns a tile of 64 reads, no block-level barrier anywhere):
//   1. the wave streams its tile (64*L contiguous bytes) from HBM with 16 B/lane
//      global_load_dwordx4 (1 KiB per wave instruction, fully coalesced);
//   2. each 16-byte chunk is packed to 16 two-bit bases (one dword) with v_dot4_u32_u8
//      and checked for non-ACGTacgt bytes with v_perm_b32 (exact); the packed tile
//      (4x smaller) is staged in the wave's private LDS slice;
//   3. each lane pulls ITS read's packed words back from LDS (ds_read_b32) and realigns them with
//      v_alignbit_b32 into a forward word array F and -- via v_bfrev_b32 -- a reverse-complement
//      array G, both in VGPRs;
//   4. every window is then two funnel shifts per strand with compile-time shift amounts
//      (v_alignbit_b32), one 64-bit compare and two v_cndmask.  No per-base rolling, no per-window branches.
//   A tile that contains any invalid byte (or the final partial tile) takes the reference-shaped
//   per-lane rolling path instead (roll_read) -- bit-exact with the iterator's skip semantics.
//   Tiles come from the same interleaved dynamic queue as the bit-sliced kernel.
#include <cstdlib>
#include "kmx_device.h"

namespace kmx {

// ------------------------------------------------------------------------------------------ sinks
// A sink consumes windows.  fast(o, fw, rc): window o of the lane's read on the all-valid fast path;
// slow(pos, fw, rc): a window yielded by roll_read (invalid ones are skipped); begin/end bracket one read
// on the slow path; tile_fast_done(nwin) closes a fast tile.

struct ReduceParams {
    kmx_summary* out;
    u32 want_hash, want_sumfw;
};
template <bool FULL>
struct SinkReduce {
    Acc acc;
    u64 maskk;
    u32 k;
    static constexpr u32 kLdsDwordsPerWave = 0;
    static constexpr bool kRagged = true;
    static u32 block_lds_dwords(const ReduceParams&) { return 0; }
    __device__ SinkReduce(const ReduceParams&, u32 k_, u32, u32*, u32, u32*, u32) : maskk(mask2k(k_)), k(k_) {}
    __device__ __forceinline__ void block_done(u64, u32, u32) {}
    __device__ __forceinline__ void fast(u32, u64 fw, u64 rc) {
        const u64 canon = fw < rc ? fw : rc;  // canonical_kmer.rs:113-119
        acc.sum_canon += canon;
        if (FULL) {
            // LexHasher(k)(canon) = MASK[k] & ~max(fw,rc): the 2-bit-group reversal of x is the
            // complement of revcomp(x) inside 2k bits (hash.rs:60-71 vs kmer.rs:124-136).
            acc.xor_hash ^= maskk ^ fw ^ rc ^ canon;
            acc.sum_fw += fw;
        }
        // Pin the accumulation order: left alone, LLVM reassociates the ~140 adds of a tile into a balanced tree and
        // keeps every window's canonical word live for it (228-256 VGPRs, 1-2 waves/SIMD instead of 4).
        asm volatile("" : "+v"(acc.sum_canon));
        if (FULL) asm volatile("" : "+v"(acc.xor_hash), "+v"(acc.sum_fw));
    }
    __device__ __forceinline__ void slow(u32, u64 fw, u64 rc) {
        const u64 canon = fw < rc ? fw : rc;
        acc.n_valid += 1;
        acc.sum_canon += canon;
        if (FULL) {
            acc.xor_hash ^= lex_hash(canon, k);
            acc.sum_fw += fw;
        }
    }
    __device__ __forceinline__ void begin_read(u64) {}
    __device__ __forceinline__ void slow_block(u32) {}   // a rolled tile: the wave has completed 16 more windows per read
    __device__ __forceinline__ void tile_slow_begin(u64 read) { begin_read(read); }
    __device__ __forceinline__ void tile_slow_emit(u32 pos, u64 fw, u64 rc) { slow(pos, fw, rc); }
    __device__ __forceinline__ void tile_slow_end() { end_read(); }
    __device__ __forceinline__ void end_read() {}
    __device__ __forceinline__ void tile_fast_done(u32 nwin) { acc.n_valid += nwin; }
    __device__ __forceinline__ void finish(const ReduceParams& p) { flush_acc(acc, p.out, FULL && p.want_hash, FULL && p.want_sumfw); }
};

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

// Histogram, 2^15..2^21 buckets, pass 1 of 2: scatter the bucket ids into 64 partitions (the top 6 bits of the bucket).
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
    uint16_t* stream;   // [n_waves][64][cap]
    u32* seg_len;       // [n_waves][64]
    u32 cap;            // entries per (wave, partition) segment, multiple of 64
};
// MODE (how the hash of a window comes about, fixed at compile time: three uniform branches per window otherwise):
//   0 LexHasher with hasher_k == k: hash = the 2k-bit complement of the LARGER of fw / rc (kmx_device.h lex_hash: the
//     reversed groups of the canonical word are the complement of the other strand) -- no hash arithmetic at all;
//   1 identity: hash = the smaller of the two;   2 LexHasher with another hasher_k.
template <int MODE>
struct SinkHistPartT {
#ifndef KMX_HIST_ROW
#define KMX_HIST_ROW 64
#endif
    static constexpr u32 NP = 64, ROW = KMX_HIST_ROW;   // partitions; ring entries per partition (u16); rows of ROW/2 ids leave together
    static constexpr u32 HALF = ROW / 2u, PER_LANE = HALF / 16u;   // ids per lane of the quarter-wave that writes a row (2 or 4)
    // LDS: per wave the {appended|written} words, the segment cursors and the rank -> ring bytes; the rings of the four waves
    // together at the end of the block's LDS, each wave's 8 KB at a multiple of 8 KB: the ring address of an id is then
    // (mix >> 26 | base >> 13 << 6) << 7 -- one v_alignbit_b32 with the wave's base in the high word -- plus the slot bytes
    static constexpr u32 kLdsDwordsPerWave = 2u * NP + NP / 4u;
    static constexpr u32 kBlockLdsAlign = 2048u;   // dwords (8 KB)
    static constexpr bool kRagged = false;
    static u32 block_lds_dwords(const HistPartParams&) { return 4u * NP * ROW / 2u; }
    HistPartParams p;
    uint16_t* ring;    // [NP][ROW]
    u32* word;         // [NP] appended (mod 2^16) << 16 | written out (mod 2^16)
    u32* cur;          // [NP] ids already in this wave's segment of the partition
    uint16_t* seg;     // this wave's [NP][cap] segments
    u64 maskk;
    u32 k, lane, lowbits;
    u32 shift_b, ring_hi, word_rel;   // 32 - log2_buckets; LDS byte address of ring[] >> 13; LDS byte address of word[] minus 4 * (ring_hi << 6)
    __device__ SinkHistPartT(const HistPartParams& p_, u32 k_, u32, u32* lds, u32 lane_, u32* block_lds, u32 tid)
        : p(p_), ring(reinterpret_cast<uint16_t*>(block_lds + (tid >> 6) * (NP * ROW / 2u))), word(lds), cur(lds + NP),
          maskk(mask2k(k_)), k(k_), lane(lane_), lowbits(p_.log2_buckets - 6u) {
        shift_b = 32u - p.log2_buckets;
        typedef u32 __attribute__((address_space(3))) * lds_u32p;
        const u32 ring_lds = (u32)(uintptr_t)(lds_u32p) reinterpret_cast<u32*>(ring);
        if (ring_lds & 8191u) __builtin_trap();   // (the launcher aligns the block region; dynamic LDS starts at 0)
        ring_hi = (u32)__builtin_amdgcn_readfirstlane(ring_lds >> 13);
        word_rel = (u32)(uintptr_t)(lds_u32p)word - ((ring_hi << 6) << 2);
        const u64 wave = (u64)blockIdx.x * 4u + (threadIdx.x >> 6);
        seg = p.stream + wave * NP * (u64)p.cap;
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
        if (((slot - w) & 0xFFFFu) < ROW) ring[q * ROW + (slot & (ROW - 1u))] = (uint16_t)(bucket & ((1u << lowbits) - 1u));
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
    __device__ __forceinline__ void fast_slot(int s, u64 fw, u64 rc) {
        static_assert(ROW == 64, "ring addressing below: 64 entries of 2 bytes");
        pend[s % NB] = mix_of_window<kComplement>(fw, rc);
        if (s % NB == NB - 1) {
            typedef u32 __attribute__((address_space(3))) * lds_u32p;
            typedef uint16_t __attribute__((address_space(3))) * lds_u16p;
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
                    const u32 a = (qb[j] << 7) + __builtin_amdgcn_ubfe(w[j], 15, 7);
                    *(lds_u16p)(uintptr_t)a = (uint16_t)(pend[j] >> shift_b);   // (bits lowbits..15 belong to the partition: pass 2 masks them off)
                }
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
        static_assert(ROW == 64, "row flush: 32 ids = 64 bytes = 4 lanes x 16 bytes");
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
                const uint4 v = *reinterpret_cast<const uint4*>(ring + q * ROW + half + 8u * l4);
                if (pos + HALF <= p.cap) {
                    *reinterpret_cast<uint4*>(seg + (u64)q * p.cap + pos + 8u * l4) = v;
                } else {   // segment full: the ids go to the global table
                    const u32 hi = q << lowbits, idm = (1u << lowbits) - 1u;
                    const u32 vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (u32 i = 0; i < 4; ++i) {
                        atomicAdd((unsigned long long*)&p.counts[hi | (vv[i] & idm)], 1ull);
                        atomicAdd((unsigned long long*)&p.counts[hi | ((vv[i] >> 16) & idm)], 1ull);
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
                const uint16_t e = ring[q * ROW + ((w + lane) & (ROW - 1u))];
                if (pos + n <= p.cap) seg[(u64)q * p.cap + pos + lane] = e;
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
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
hist_part_reduce_kernel(const uint16_t* __restrict__ stream, const u32* __restrict__ seg_len, u32 cap, u32 n_waves,
                        u32 log2_buckets, u64* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) u32 tab[];
    const u32 lowbits = log2_buckets - 6u, nb = 1u << lowbits, idm = nb - 1u;
    const u32 q = blockIdx.x;
    for (u32 j = threadIdx.x; j < nb; j += THREADS) tab[j] = 0;
    __syncthreads();
    // every WAVE of the block walks its own segments (w == its index mod the waves of the partition's blocks): a segment is
    // ~25 KB, too short for 512 threads to keep several loads each in flight
    const u32 wv = threadIdx.x >> 6, ln = threadIdx.x & 63u, nwv = THREADS / 64u;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    // (an id carries the low bits of its partition above bit lowbits: the scan writes the bucket's low 16 bits as they are)
    auto count8 = [&](const u32x4 v) {
        atomicAdd(&tab[v.x & idm], 1u);
        atomicAdd(&tab[(v.x >> 16) & idm], 1u);
        atomicAdd(&tab[v.y & idm], 1u);
        atomicAdd(&tab[(v.y >> 16) & idm], 1u);
        atomicAdd(&tab[v.z & idm], 1u);
        atomicAdd(&tab[(v.z >> 16) & idm], 1u);
        atomicAdd(&tab[v.w & idm], 1u);
        atomicAdd(&tab[(v.w >> 16) & idm], 1u);
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
        for (u32 j = (len & ~7u) + ln; j < len; j += 64u) atomicAdd(&tab[sp[j] & idm], 1u);
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < nb; j += THREADS) {
        const u32 c = tab[j];
        if (c) atomicAdd((unsigned long long*)&counts[((u64)q << lowbits) | j], (unsigned long long)c);
    }
}

struct WindowsParams {
    u64 *fw, *rc, *canon;
    uint8_t* flags;
    const u64* win_offsets;   // ragged reads: slot of window 0 of read r (n_reads+1 entries); nullptr: r*W
};
// dense per-window outputs, slot(read, pos) = read*W + pos.  Write-bound by construction (8 B per array per k-mer).
// Fast path: the 16 windows of a block are staged through LDS (lane-major, pitch 17) and written back transposed,
// so every store instruction covers 128-byte contiguous runs (16 windows of one read) instead of 64 scattered
// 8-byte words at a stride of W*8 bytes.
template <bool ALIGNED>
struct SinkWindowsT {
    static constexpr u32 PITCH = 17;                        // u64 per lane row (16 + 1 pad: conflict-free both ways)
    static constexpr u32 PLANE = 64u * PITCH * 2u;          // dwords of one staged u64 array of a wave
    // staging sized by what the caller asked for (with all three u64 planes a block holds 108 KB = one block per CU and one
    // wave per SIMD; canonical words alone: 9 KB per wave, three blocks per CU)
#ifndef KMX_WIN_WAVES
#define KMX_WIN_WAVES 3
#endif
    // store latency is all this sink waits for: occupancy over registers (the line-aligned variant spills at 168 registers
    // and its 17 KB ring per wave caps a CU at two blocks anyway)
    static constexpr int kWaves = ALIGNED ? 2 : KMX_WIN_WAVES;
    static constexpr u32 kLdsDwordsPerWave = 0;
    // One u64 array and no flags (the usual call: the canonical words): write-back in units of whole, 128-byte ALIGNED
    // lines of the output.  The lines of read r are shifted by a = r*W mod 16 slots against its windows, so the windows
    // of two passes sit in a 32-slot ring per read and pass j writes windows [16j - a, 16j + 16 - a).  (Writing windows
    // [16j, 16j+16) as they come leaves every line half-written until the next pass, and the chip's write rate drops to
    // 2.9 TB/s at W = 120 (64-byte aligned runs) and 2.1 TB/s at W = 130 (16-byte aligned), against 4.6 TB/s at W = 128;
    // line by line: 3.5 and 3.3 TB/s.)
    static constexpr u32 RPITCH = 33;
    static __host__ __device__ bool line_aligned(const WindowsParams& p) {
        return ALIGNED;
    }
    // the caller's side of ALIGNED: one u64 array, no flags (W a multiple of 16: the plain write-back is aligned already)
    static bool wants_aligned(const WindowsParams& p, u32 W) {
        return ((p.fw ? 1u : 0u) + (p.rc ? 1u : 0u) + (p.canon ? 1u : 0u)) == 1u && !p.flags && (W & 15u) != 0u;
    }
    static __host__ __device__ u32 wave_dwords(const WindowsParams& p) {
        if (line_aligned(p)) return 64u * RPITCH * 2u;
        return ((p.fw ? 1u : 0u) + (p.rc ? 1u : 0u) + (p.canon ? 1u : 0u)) * PLANE + (p.flags ? 64u * 16u / 4u : 0u) +
               (p.win_offsets ? 64u * 3u : 0u);   // ragged: first slot and window count of the tile's 64 reads
    }
    static u32 block_lds_dwords(const WindowsParams& p) { return 4u * wave_dwords(p); }
    WindowsParams p;
    u64 *Tfw, *Trc, *Tcn;   // [64][PITCH] staging of the arrays that are wanted
    uint8_t* TF;            // [64][16] flags
    u64* out1;              // line-aligned mode: the one output array (nullptr: staged mode)
    u64* WOL;               // ragged: [64] slot of window 0 of the tile's reads
    u32* NWL;               // ragged: [64] their window counts
    u64 base;      // slot of window 0 of the current read
    u32 W, nwr, next, lane; // W: windows per read (uniform layout); nwr: windows of this lane's read
    static constexpr bool kRagged = !ALIGNED;   // (the line-aligned write-back assumes one window count per read)
    __device__ SinkWindowsT(const WindowsParams& p_, u32, u32 W_, u32*, u32 lane_, u32* block_lds, u32 tid)
        : p(p_), base(0), W(W_), next(0), lane(lane_) {
        u32* mine = block_lds + (tid >> 6) * wave_dwords(p_);
        Tfw = reinterpret_cast<u64*>(mine);
        Trc = Tfw + (p_.fw ? PLANE / 2u : 0u);
        Tcn = Trc + (p_.rc ? PLANE / 2u : 0u);
        TF = reinterpret_cast<uint8_t*>(Tcn + (p_.canon ? PLANE / 2u : 0u));
        WOL = reinterpret_cast<u64*>(TF + (p_.flags ? 64u * 16u : 0u));
        NWL = reinterpret_cast<u32*>(WOL + 64);
        nwr = W_;
        out1 = line_aligned(p_) ? (p_.fw ? p_.fw : p_.rc ? p_.rc : p_.canon) : nullptr;
    }
    __device__ __forceinline__ void store(u64 slot, u64 fw, u64 rc) {
        const bool lt = fw < rc;
        if (p.fw) p.fw[slot] = fw;
        if (p.rc) p.rc[slot] = rc;
        if (p.canon) p.canon[slot] = lt ? fw : rc;
        if (p.flags) p.flags[slot] = (uint8_t)(KMX_WIN_VALID | (lt ? KMX_WIN_FW_CANONICAL : 0u));
    }
    __device__ __forceinline__ void zero_to(u32 end) {
        for (; next < end; ++next) {
            const u64 s = base + next;
            if (p.fw) p.fw[s] = 0;
            if (p.rc) p.rc[s] = 0;
            if (p.canon) p.canon[s] = 0;
            if (p.flags) p.flags[s] = 0;
        }
    }
    __device__ __forceinline__ void fast(u32 o, u64 fw, u64 rc) {
        const bool lt = fw < rc;
        if (out1) {
            Tfw[lane * RPITCH + (o & 31u)] = p.fw ? fw : p.rc ? rc : (lt ? fw : rc);
            return;
        }
        const u32 s = o & 15u, at = lane * PITCH + s;
        if (p.fw) Tfw[at] = fw;
        if (p.rc) Trc[at] = rc;
        if (p.canon) Tcn[at] = lt ? fw : rc;
        if (p.flags) TF[lane * 16u + s] = (uint8_t)(KMX_WIN_VALID | (lt ? KMX_WIN_FW_CANONICAL : 0u));
    }
    // all 64 lanes have staged windows [o0, o0+cnt) of reads [read0, read0+64): write them out coalesced
    __device__ __forceinline__ void block_done(u64 read0, u32 o0, u32 cnt) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (out1) {
            const bool last = o0 + cnt == W;      // the last pass also writes what is left of each read (< 32 windows)
            for (u32 sub = 0; sub < (last ? 2u : 1u); ++sub) {
#pragma unroll 4
                for (u32 it = 0; it < 16u; ++it) {
                    const u32 idx = it * 64u + lane, r = idx >> 4, s = idx & 15u;
                    const u64 read = read0 + r;
                    const u32 a = ((u32)(read & 15u) * (W & 15u)) & 15u;   // read*W mod 16
                    const u32 lo = o0 > a ? o0 - a : 0u;
                    const u32 hi = last ? W : o0 + 16u - a;
                    const u32 o = lo + 16u * sub + s;
                    if (o < hi) out1[read * W + o] = Tfw[r * RPITCH + (o & 31u)];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            return;
        }
#pragma unroll 4
        for (u32 it = 0; it < 16u; ++it) {
            const u32 idx = it * 64u + lane, r = idx >> 4, sw = idx & 15u;
            if (sw < cnt && (!p.win_offsets || o0 + sw < NWL[r])) {
                const u64 slot = (p.win_offsets ? WOL[r] : (read0 + r) * W) + o0 + sw;
                const u32 at = r * PITCH + sw;
                if (p.fw) p.fw[slot] = Tfw[at];
                if (p.rc) p.rc[slot] = Trc[at];
                if (p.canon) p.canon[slot] = Tcn[at];
                if (p.flags) p.flags[slot] = TF[r * 16u + sw];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ void slow(u32 pos, u64 fw, u64 rc) {
        zero_to(pos);
        store(base + pos, fw, rc);
        next = pos + 1u;
    }
    __device__ __forceinline__ void begin_read(u64 read) {
        if (p.win_offsets) {
            base = p.win_offsets[read];
            nwr = (u32)(p.win_offsets[read + 1u] - base);
            if (!ALIGNED) { WOL[lane] = base; NWL[lane] = nwr; }   // for the transposed write-back (block_done starts with a wave barrier)
        } else {
            base = read * W;
        }
        next = 0;
    }
    // A tile rolled per lane (an invalid byte in it) is staged and written back like a fast one -- skipped windows stay the
    // zeros the slots are pre-filled with -- instead of 8-byte stores 1 KB apart: tile_slow_* are the hooks of that path
    // (begin_read / slow / end_read remain the direct-store path of the final partial tile).
    u64 slow_read0;
    bool slow_staged;
    __device__ __forceinline__ void prefill(u32 o0) {     // zero the staging slots of windows [o0, o0+16)
#pragma unroll 4
        for (u32 sI = 0; sI < 16u; ++sI) {
            const u32 o = o0 + sI;
            if (out1) Tfw[lane * RPITCH + (o & 31u)] = 0;
            else {
                const u32 at = lane * PITCH + (o & 15u);
                if (p.fw) Tfw[at] = 0;
                if (p.rc) Trc[at] = 0;
                if (p.canon) Tcn[at] = 0;
                if (p.flags) TF[lane * 16u + (o & 15u)] = 0;
            }
        }
    }
    __device__ __forceinline__ void tile_slow_begin(u64 read) {
        begin_read(read);
        slow_read0 = read - lane;
        prefill(0);
    }
    __device__ __forceinline__ void tile_slow_emit(u32 pos, u64 fw, u64 rc) { fast(pos, fw, rc); }
    __device__ __forceinline__ void slow_block(u32 wb) {
        const u32 o0 = 16u * wb;
        if (p.win_offsets == nullptr && o0 >= W) return;            // (uniform layout: past the last window block)
        const u32 cnt = p.win_offsets ? 16u : (W - o0 < 16u ? W - o0 : 16u);
        block_done(slow_read0, o0, cnt);
        prefill(o0 + 16u);
    }
    __device__ __forceinline__ void tile_slow_end() {}
    __device__ __forceinline__ void end_read() { zero_to(nwr); }
    __device__ __forceinline__ void tile_fast_done(u32) {}
    __device__ __forceinline__ void finish(const WindowsParams&) {}
};

// ----------------------------------------------------------------------------------------- kernel
// NW  = packed dwords per read = ceil(L/16) rounded up to an instantiated size (L <= 16*NW)
// V   = 1: k in [2,17]   2: k in [18,32]   (fixes the static register index of the rc window)
// DW  = dwords per k-mer (1: k<=16, 2: k>=17)
// RAGGED: reads of different lengths, read r = bases[offsets[r], offsets[r+1]).  A tile is still 64 consecutive reads =
// one contiguous byte span, streamed from its 16-byte-aligned start; lanes carry their own start and window count,
// windows past a lane's read are masked.  A tile whose span or longest read does not fit the NW-word frame, or that
// would load past the end of the buffer, takes the per-lane rolling path.
#ifndef KMX_SCAN_RAGGED_2COPY
#define KMX_SCAN_RAGGED_2COPY 0
#endif
#ifndef KMX_SCAN_DEV_NOGUARD
#define KMX_SCAN_DEV_NOGUARD 0   // dev: drop the per-window length mask of the ragged kernel (wrong for unequal lengths; timing only)
#endif
#ifndef KMX_SCAN_WAVES
#define KMX_SCAN_WAVES 1   // waves per SIMD the register allocation is sized for (hipcc otherwise spends up to 256 VGPRs on hoisting)
#endif
// a sink may ask for a register budget of its own (static constexpr int kWaves)
template <typename S, typename = void> struct SinkWaves { static constexpr int value = KMX_SCAN_WAVES; };
template <typename S> struct SinkWaves<S, decltype((void)S::kWaves)> { static constexpr int value = S::kWaves; };
// a sink may take the 16 windows of an unrolled block together (static constexpr bool kBatch16 = true; fast_slot())
template <typename S, typename = void> struct SinkBatch16 { static constexpr bool value = false; };
template <typename S> struct SinkBatch16<S, decltype((void)S::kBatch16)> { static constexpr bool value = S::kBatch16; };
// a sink may want its block-level LDS region to start at a multiple of kBlockLdsAlign dwords (static constexpr u32)
template <typename S, typename = void> struct SinkBlockAlign { static constexpr u32 value = 1u; };
template <typename S> struct SinkBlockAlign<S, decltype((void)S::kBlockLdsAlign)> { static constexpr u32 value = S::kBlockLdsAlign; };
// a sink may ask for complemented windows (fw ^ mask, rc ^ mask) from the kernel's fast path (static constexpr bool kComplement)
template <typename S, typename = void> struct SinkComplement { static constexpr bool value = false; };
template <typename S> struct SinkComplement<S, decltype((void)S::kComplement)> { static constexpr bool value = S::kComplement; };
template <typename S, int NW> constexpr int sink_waves() { return NW <= 10 ? SinkWaves<S>::value : KMX_SCAN_WAVES; }   // (the 16-word frame would spill 1 KB)
template <int NW, int V, int DW, typename Sink, typename Params, bool RAGGED = false>
__global__ void __launch_bounds__(256, (sink_waves<Sink, NW>()))
scan_uniform_kernel(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 k, Params params,
                    unsigned long long* __restrict__ queue, const u64* __restrict__ offsets, u32 lead) {
    // `lead` (uniform reads whose first byte is not 16-byte aligned): `bases` is the aligned address below it, read r
    // starts at byte lead + r*L and a tile spans one more chunk (as a ragged tile streamed from its aligned start does)
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 lane = threadIdx.x & 63u;
    const u32 wib = threadIdx.x >> 6;
    const u32 chunks_u = RAGGED ? 64u * NW : 4u * L + (lead != 0u ? 1u : 0u);  // 16-byte chunks per 64-read tile (ragged: the most a tile may span)
    const u32 ldsw = (chunks_u + 1u + 6u + 3u) & ~3u; // front pad 1, tail pad >= 6
    u32* P = lds + wib * (ldsw + Sink::kLdsDwordsPerWave);

    const u64 n_full = n_reads >> 6;
    const u64 wave_id = (u64)blockIdx.x * 4u + wib;
    const u64 total_bytes = RAGGED ? offsets[n_reads] : 0;

    // per-lane alignment of this lane's read inside the packed tile (LDS index 1+c holds bases [16c,16c+16))
    u32 posF = lane * L + lead + 16u;
    u32 qF = posF >> 4, aF = 2u * (posF & 15u);
    const u32 delta = (1u - k) & 15u;  // rc stream pre-offset so that rc sub-shift == 30-2s
    u32 posR = posF - delta;
    u32 qR = posR >> 4, aR = 2u * (posR & 15u);

    u32 omax = L - k;  // last window start (uniform: of every read; ragged: the longest read of the tile)
    u32 imax = omax >> 4, smax = omax & 15u;
    const u64 maskk = mask2k(k);
    const u32 mlo = (u32)maskk;
    const u32 mhi = (u32)(maskk >> 32);
    u32 nwin = omax + 1u;       // windows of this lane's read
    u32 chunks = chunks_u;

    constexpr u32 BAL = SinkBlockAlign<Sink>::value;
    Sink sink(params, k, nwin, P + ldsw, lane, lds + (4u * (ldsw + Sink::kLdsDwordsPerWave) + BAL - 1u) / BAL * BAL, threadIdx.x);

    [[maybe_unused]] u32 nwin_min = nwin;   // ragged: the shortest read of the tile (blocks of windows below it need no per-lane mask)
    // `slot`: position of the window inside a fully unrolled block of 16 (a compile-time value there), -1 elsewhere; a
    // sink with kBatch16 collects the 16 windows of such a block and consumes them together in block_done()
    auto window = [&](u32 o, u32 f0, u32 f1, u32 f2, u32 g0, u32 g1, u32 g2, u32 sf, u32 sr, bool guard = RAGGED, int slot = -1) {
        if (guard && o >= nwin) return;   // past the end of this lane's (shorter) read
        u64 fw, rc;
        if (DW == 2) {
            const u32 fw_lo = alignbit(f1, f0, sf);
            const u32 fw_hi = alignbit(f2, f1, sf) & mhi;
            const u32 rc_lo = alignbit(g1, g0, sr);
            const u32 rc_hi = alignbit(g2, g1, sr) & mhi;
            fw = ((u64)fw_hi << 32) | fw_lo;
            rc = ((u64)rc_hi << 32) | rc_lo;
        } else {
            fw = (u64)(alignbit(f1, f0, sf) & mlo);
            rc = (u64)(alignbit(g1, g0, sr) & mlo);
        }
        if constexpr (SinkBatch16<Sink>::value) {
            if (slot >= 0) {
                sink.fast_slot(slot, fw, rc);
                return;
            }
        }
        sink.fast(o, fw, rc);
    };

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
    // ragged: this lane's [start, end) of the NEXT tile, fetched one tile ahead so that the tile's byte loads never wait
    // behind a dependent offsets load
    u64 nx_off = 0, nx_end = 0;
    if (RAGGED && next_tile < n_full) {
        nx_off = offsets[next_tile * 64u + lane];
        nx_end = offsets[next_tile * 64u + lane + 1u];
    }
    for (u64 tile = next_tile; tile < n_full; tile = next_tile) {
        next_tile = dequeue();
        const u64 read = tile * 64u + lane;
        const uint4* __restrict__ tb = reinterpret_cast<const uint4*>(bases + tile * 64u * (u64)L);
        u64 my_off = 0;
        u32 my_len = 0;
        bool tile_fits = true;
        if constexpr (RAGGED) {
            my_off = nx_off;
            my_len = (u32)(nx_end - nx_off);
            // (the builtins return int: through u32 first, or offsets >= 2^31 get sign-extended into the high word)
            const u32 t0l = __builtin_amdgcn_readfirstlane((u32)nx_off), t0h = __builtin_amdgcn_readfirstlane((u32)(nx_off >> 32));
            const u32 t1l = __builtin_amdgcn_readlane((u32)nx_end, 63), t1h = __builtin_amdgcn_readlane((u32)(nx_end >> 32), 63);
            const u64 t0 = ((u64)t0h << 32) | t0l, t1 = ((u64)t1h << 32) | t1l;
            if (next_tile < n_full) {
                nx_off = offsets[next_tile * 64u + lane];
                nx_end = offsets[next_tile * 64u + lane + 1u];
            }
            const u64 base_al = t0 & ~15ull;
            const u64 n_ch = (t1 - base_al + 15u) >> 4;
            const u32 max_len = (u32)wave_max_u32(my_len);
            tile_fits = n_ch <= 64u * NW && max_len <= 16u * NW && base_al + 16u * n_ch <= total_bytes;
            chunks = (u32)n_ch;
            tb = reinterpret_cast<const uint4*>(bases + base_al);
            posF = (u32)(my_off - base_al) + 16u;
            qF = posF >> 4;
            aF = 2u * (posF & 15u);
            posR = posF - delta;
            qR = posR >> 4;
            aR = 2u * (posR & 15u);
            nwin = my_len >= k ? my_len - k + 1u : 0u;
            nwin_min = ~wave_max_u32(~nwin);
            omax = max_len >= k ? max_len - k : 0u;
            imax = omax >> 4;
            smax = omax & 15u;
            if (max_len < k) {   // nothing to emit in this tile
                sink.tile_fast_done(0);
                continue;
            }
        }
        // ---- 1. stream the tile: all loads in flight before the first use
        uint4 w[NW];
        if (tile_fits) {
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const u32 c = it * 64u + lane;
                if (c < chunks) w[it] = tb[c];
            }
        }
        // ---- 2. pack + validate, stage packed words in LDS
        u32 bad = 0;
        if (tile_fits) {
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const u32 c = it * 64u + lane;
                if (c < chunks) P[1u + c] = encode16(w[it], bad);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (!tile_fits || __any(chunk_has_invalid(bad))) {
            // ---- rare: a non-ACGTacgt byte somewhere in this tile (or a ragged tile outside the frame) -> exact iterator semantics
            sink.tile_slow_begin(read);
            u32 roll_max = L;
            if constexpr (RAGGED) roll_max = (u32)wave_max_u32(my_len);
            roll_read_stepped(RAGGED ? bases + my_off : bases + lead + read * (u64)L, RAGGED ? my_len : L, roll_max, k,
                              [&](u32 pos, u64 fw, u64 rc) { sink.tile_slow_emit(pos, fw, rc); }, [&](u32 wb) { sink.slow_block(wb); });
            sink.tile_slow_end();
            continue;
        }

        // ---- 3. this lane's read: forward words F, reverse-complement words G
        u32 F[NW + 2], G[NW + 2];
        {
            u32 R[NW + 1];
#pragma unroll
            for (int j = 0; j <= NW; ++j) R[j] = P[qF + j];
#pragma unroll
            for (int i = 0; i < NW; ++i) F[i] = SinkComplement<Sink>::value ? ~alignbit(R[i + 1], R[i], aF) : alignbit(R[i + 1], R[i], aF);
            F[NW] = 0;
            F[NW + 1] = 0;
            u32 Rr[NW + 2];
#pragma unroll
            for (int j = 0; j <= NW + 1; ++j) Rr[j] = P[qR + j];
#pragma unroll
            for (int m = 0; m <= NW; ++m)
                G[m] = SinkComplement<Sink>::value ? revgroups32(alignbit(Rr[NW - m + 1], Rr[NW - m], aR)) : revgroups32(~alignbit(Rr[NW - m + 1], Rr[NW - m], aR));
            G[NW + 1] = 0;
        }

        // ---- 4. windows: o = 16*i + s;  fw from F[i..i+2] >> 2s;  rc from G[M..M+2] >> (30-2s), M = NW-V-i
        sink.begin_read(read);
#pragma unroll
        for (int i = 0; i <= NW - V; ++i) {
            const int M = NW - V - i;
            if ((u32)i < imax) {
                // Opaque copies of the six source words, made INSIDE the block: LLVM's speculative execution otherwise
                // hoists the (cheap, side-effect-free) funnel shifts of every block above the chain of uniform branches
                // and keeps them all live -- 228-256 VGPRs, 1-2 waves per SIMD instead of 4.
                u32 f0 = F[i], f1 = F[i + 1], f2 = F[i + 2], g0 = G[M], g1 = G[M + 1], g2 = G[M + 2];
                asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(g0), "+v"(g1), "+v"(g2));
#if KMX_SCAN_RAGGED_2COPY
                if (!RAGGED || 16u * i + 16u <= nwin_min) {   // every lane owns all 16 windows: straight-line code
#pragma unroll
                    for (int s = 0; s < 16; ++s) window(16 * i + s, f0, f1, f2, g0, g1, g2, 2 * s, 30 - 2 * s, false);
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s) window(16 * i + s, f0, f1, f2, g0, g1, g2, 2 * s, 30 - 2 * s, true);
                }
#else
                // (a second, unmasked copy of the block for tiles of equal-length reads doubles the code past the
                //  instruction cache and costs more than the per-window mask it saves)
#pragma unroll
                for (int s = 0; s < 16; ++s) window(16 * i + s, f0, f1, f2, g0, g1, g2, 2 * s, 30 - 2 * s, RAGGED && !KMX_SCAN_DEV_NOGUARD, RAGGED ? -1 : s);
#endif
                sink.block_done(tile * 64u, 16u * i, 16u);
            } else if ((u32)i == imax) {
                for (u32 s = 0; s <= smax; ++s) window(16u * i + s, F[i], F[i + 1], F[i + 2], G[M], G[M + 1], G[M + 2], 2u * s, 30u - 2u * s);
                sink.block_done(tile * 64u, 16u * i, smax + 1u);
            }
        }
        sink.tile_fast_done(nwin);
    }

    // ---- final partial tile (n_reads % 64 reads): per-lane rolling
    const u32 rem = (u32)(n_reads & 63u);
    if (rem != 0u && wave_id == 0 && lane < rem) {
        const u64 read = n_full * 64u + lane;
        sink.begin_read(read);
        if constexpr (RAGGED) {
            const u64 o0 = offsets[read];
            roll_read(bases + o0, (u32)(offsets[read + 1u] - o0), k, [&](u32 pos, u64 fw, u64 rc) { sink.slow(pos, fw, rc); });
        } else {
            roll_read(bases + lead + read * (u64)L, L, k, [&](u32 pos, u64 fw, u64 rc) { sink.slow(pos, fw, rc); });
        }
        sink.end_read();
    }
    sink.finish(params);
}

// ------------------------------------------------------------------ launchers

// `pre(grid)` runs once the grid size is known and may finish filling `params` (the partitioned histogram sizes its
// per-wave segments from it); it returns false to abandon the launch.
struct NoPre {
    bool operator()(u64) const { return true; }
};
template <int NW, int V, int DW, typename Sink, typename Params, typename Pre = NoPre, bool RAGGED = false>
static hipError_t launch_one(const uint8_t* bases, u64 n_reads, u32 L, u32 k, Params& params,
                             unsigned long long* queue, int n_cu, hipStream_t stream, Pre pre = Pre(), const u64* offsets = nullptr) {
    auto kern = scan_uniform_kernel<NW, V, DW, Sink, Params, RAGGED>;
    u32 lead = 0;   // uniform reads from a base that is not 16-byte aligned: streamed from the aligned address below it
    if constexpr (!RAGGED) {
        lead = (u32)(reinterpret_cast<uintptr_t>(bases) & 15u);
        bases -= lead;
        if (lead != 0u && 4u * L + 1u > 64u * (u32)NW) return hipErrorInvalidValue;   // (scan_domain checks: the extra chunk must fit the frame)
    }
    const u32 chunks = RAGGED ? 64u * NW : 4u * L + (lead != 0u ? 1u : 0u);
    const u32 ldsw = (chunks + 1u + 6u + 3u) & ~3u;
    constexpr u32 BAL = SinkBlockAlign<Sink>::value;
    const size_t lds_bytes = (size_t)((4u * (ldsw + Sink::kLdsDwordsPerWave) + BAL - 1u) / BAL * BAL) * 4u + (size_t)Sink::block_lds_dwords(params) * 4u;
    // blocks per CU, cached per host thread and device (the ABI's model is one thread per context / GPU: a plain static
    // would be written by all of them at once, and the function attribute below is a per-device setting)
    static thread_local int bpc = 0, bpc_dev = -1;
    static thread_local size_t bpc_lds = 0;
    int dev_now = -1;
    (void)hipGetDevice(&dev_now);
    if (bpc == 0 || bpc_lds != lds_bytes || bpc_dev != dev_now) {
        bpc_dev = dev_now;
        if (lds_bytes > 64u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        int b = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kern, 256, lds_bytes);
        if (e != hipSuccess) return e;
        bpc = b > 0 ? b : 1;
        bpc_lds = lds_bytes;
    }
    const u64 n_tiles = (n_reads + 63u) >> 6;
    u64 grid = (u64)n_cu * (u64)bpc;
    if (const char* ov = getenv("KMX_DEV_BPC")) grid = (u64)n_cu * (u64)atoi(ov);   // (dev) blocks per CU
    const u64 need = (n_tiles + 3u) / 4u;
    if (grid > need) grid = need;
    if (grid == 0) grid = 1;
    if (!pre(grid)) return hipErrorOutOfMemory;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds_bytes, stream, bases, n_reads, L, k, params, queue, offsets, lead);
    return hipGetLastError();
}

static bool scan_domain(const uint8_t* bases, u64 n_reads, u32 L, u32 k) {
    if (k < 2 || k > 31 || L < k || L > 256) return false;
    if ((reinterpret_cast<uintptr_t>(bases) & 15u) && (L == 160 || L == 256)) return false;   // the extra chunk of an unaligned start must fit the frame
    return n_reads * (u64)L < (1ull << 62);
}
// ragged reads: L is an optional upper bound of the lengths (0 = unknown); reads longer than the frame fall back per tile
static bool scan_domain_ragged(const uint8_t* bases, u32 L, u32 k) {
    return !(k < 2 || k > 31 || L > 256 || (reinterpret_cast<uintptr_t>(bases) & 15u));
}

// offsets != nullptr: ragged reads; L is then only an upper bound of the read lengths (0 = unknown) that selects the frame
template <typename SinkT, typename Params, typename Pre = NoPre>
static hipError_t dispatch(const uint8_t* bases, u64 n_reads, u32 L, u32 k, Params p, unsigned long long* queue,
                           int n_cu, hipStream_t stream, Pre pre = Pre(), const u64* offsets = nullptr) {
    const bool big = L > 160 || (offsets && L == 0);
    if constexpr (!SinkT::kRagged) {
        if (offsets) return hipErrorInvalidValue;
    } else if (offsets) {
        if (k <= 16) {
            if (big) return launch_one<16, 1, 1, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
            return launch_one<10, 1, 1, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
        }
        if (k == 17) {   // the one k whose rc window sits at the V = 1 register index while the k-mer needs two dwords
            if (big) return launch_one<16, 1, 2, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
            return launch_one<10, 1, 2, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
        }
        if (big) return launch_one<16, 2, 2, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
        return launch_one<10, 2, 2, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
    }
    if (k <= 16) {
        if (big) return launch_one<16, 1, 1, SinkT, Params, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
        return launch_one<10, 1, 1, SinkT, Params, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
    }
    if (k == 17) {
        if (big) return launch_one<16, 1, 2, SinkT, Params, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
        return launch_one<10, 1, 2, SinkT, Params, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
    }
    if (big) return launch_one<16, 2, 2, SinkT, Params, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
    return launch_one<10, 2, 2, SinkT, Params, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
}

template <typename SinkHistPart, typename Pre>
static hipError_t dispatch_part_mode(const uint8_t* bases, u64 n_reads, u32 L, u32 k, HistPartParams& p, unsigned long long* queue,
                                     int n_cu, hipStream_t stream, Pre pre) {
    const bool big = L > 160;
    if (k <= 16) {
        if (big) return launch_one<16, 1, 1, SinkHistPart, HistPartParams, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
        return launch_one<10, 1, 1, SinkHistPart, HistPartParams, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
    }
    if (k == 17) {
        if (big) return launch_one<16, 1, 2, SinkHistPart, HistPartParams, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
        return launch_one<10, 1, 2, SinkHistPart, HistPartParams, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
    }
    if (big) return launch_one<16, 2, 2, SinkHistPart, HistPartParams, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
    return launch_one<10, 2, 2, SinkHistPart, HistPartParams, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
}

template <typename Pre>
static hipError_t dispatch_part(const uint8_t* bases, u64 n_reads, u32 L, u32 k, HistPartParams& p, unsigned long long* queue,
                                int n_cu, hipStream_t stream, Pre pre) {
    if (p.hasher != KMX_HASH_LEX) return dispatch_part_mode<SinkHistPartT<1>, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
    if (p.hk == k) return dispatch_part_mode<SinkHistPartT<0>, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
    return dispatch_part_mode<SinkHistPartT<2>, Pre>(bases, n_reads, L, k, p, queue, n_cu, stream, pre);
}

// Each returns hipSuccess and sets *handled=false when (L,k) is outside the fast kernel's domain.
// `queue`: 32 zeroed u64 heads, 128 B apart, owned by the caller for the duration of the launch.
hipError_t launch_scan_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw,
                               kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                               const u64* offsets) {
    *handled = offsets ? scan_domain_ragged(bases, L, k) : scan_domain(bases, n_reads, L, k);
    if (!*handled) return hipSuccess;
    const ReduceParams p{out, want_hash ? 1u : 0u, want_sumfw ? 1u : 0u};
    if (want_hash || want_sumfw) return dispatch<SinkReduce<true>>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets);
    return dispatch<SinkReduce<false>>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets);
}

// Histogram over uniform reads.  2^b <= 2^14: block-private LDS tables (SinkHistLds).  2^15..2^21: two passes through
// 64 partitions (SinkHistPart + hist_part_reduce_kernel) in chunks of reads sized to `scratch_budget` bytes of
// caller-provided scratch (`get_scratch(user, bytes)` returns a device buffer of at least `bytes`, or nullptr).
// Larger tables, or no scratch: device-scope u64 atomics (SinkHist).
hipError_t launch_hist_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u32 hasher, u32 hk, u32 log2_buckets,
                               u64* counts, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                               void* (*get_scratch)(void*, size_t), void* user, size_t scratch_budget, const u64* offsets) {
    *handled = offsets ? scan_domain_ragged(bases, L, k) : scan_domain(bases, n_reads, L, k);
    if (!*handled) return hipSuccess;
    const HistParams p{counts, hasher, hk, log2_buckets};
    if (offsets) {   // ragged reads: LDS tables up to 2^14 buckets, device atomics above (the partitioned path sizes its segments from a uniform L)
        if (log2_buckets <= 14u) return dispatch<SinkHistLds>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets);
        return dispatch<SinkHist>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets);
    }
    if (log2_buckets <= 14u) return dispatch<SinkHistLds>(bases, n_reads, L, k, p, queue, n_cu, stream);
    if (log2_buckets <= 21u && get_scratch != nullptr && n_reads >= 4096u) {
        const u64 W = L - k + 1u;
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
                    pp.stream = reinterpret_cast<uint16_t*>(buf);
                    pp.seg_len = reinterpret_cast<u32*>(buf + stream_bytes);
                    return true;
                };
                if (first != 0) {
                    hipError_t e = hipMemsetAsync(queue, 0, 32 * 128, stream);
                    if (e != hipSuccess) return e;
                }
                // the hook fills pp through the reference captured above; dispatch takes its params by value, so hand it
                // a proxy that copies the finished pp at launch time
                hipError_t e = dispatch_part(bases + first * (u64)L, n, L, k, pp, queue, n_cu, stream, pre);
                if (e == hipErrorOutOfMemory) {   // no scratch: the atomic sink handles the rest
                    (void)hipGetLastError();
                    return dispatch<SinkHist>(bases + first * (u64)L, n_reads - first, L, k, p, queue, n_cu, stream);
                }
                if (e != hipSuccess) return e;
                const u32 nb_bytes = 4u << (log2_buckets - 6u);
                auto red = hist_part_reduce_kernel<512>;
                if (nb_bytes > 64u * 1024u) {
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(red), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb_bytes);
                    if (e != hipSuccess) return e;
                }
                const u32 groups = n_waves < 16u ? n_waves : 16u;
                hipLaunchKernelGGL(red, dim3(64, groups), dim3(512), nb_bytes, stream, pp.stream, pp.seg_len, pp.cap, n_waves,
                                   log2_buckets, counts);
                e = hipGetLastError();
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
    }
    return dispatch<SinkHist>(bases, n_reads, L, k, p, queue, n_cu, stream);
}

hipError_t launch_windows_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u64* fw, u64* rc, u64* canon,
                                  uint8_t* flags, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled) {
    *handled = scan_domain(bases, n_reads, L, k);
    if (!*handled) return hipSuccess;
    const WindowsParams p{fw, rc, canon, flags, nullptr};
    if (SinkWindowsT<true>::wants_aligned(p, L - k + 1u)) return dispatch<SinkWindowsT<true>>(bases, n_reads, L, k, p, queue, n_cu, stream);
    return dispatch<SinkWindowsT<false>>(bases, n_reads, L, k, p, queue, n_cu, stream);
}

// ragged reads: win_offsets[r] = slot of window 0 of read r (n_reads+1 entries); L = optional bound of the read lengths
hipError_t launch_windows_ragged(const uint8_t* bases, const u64* offsets, const u64* win_offsets, u64 n_reads, u32 L, u32 k,
                                 u64* fw, u64* rc, u64* canon, uint8_t* flags, unsigned long long* queue, int n_cu,
                                 hipStream_t stream, bool* handled) {
    *handled = offsets && win_offsets && scan_domain_ragged(bases, L, k);
    if (!*handled) return hipSuccess;
    const WindowsParams p{fw, rc, canon, flags, win_offsets};
    return dispatch<SinkWindowsT<false>>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets);
}

}  // namespace kmx
