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
#include <type_traits>
#include "kmx_scan_kernel.h"

namespace kmx {

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

struct WindowsParams {
    u64 *fw, *rc, *canon;
    uint8_t* flags;
    const u64* win_offsets;   // ragged reads: slot of window 0 of read r (n_reads+1 entries); nullptr: r*W
};
// dense per-window outputs, slot(read, pos) = read*W + pos.  Write-bound by construction (8 B per array per k-mer).
// Fast path: the 16 windows of a block are staged through LDS (lane-major, pitch 17) and written back transposed,
// so every store instruction covers 128-byte contiguous runs (16 windows of one read) instead of 64 scattered
// 8-byte words at a stride of W*8 bytes.
// RG (round 4): the line-aligned ring for RAGGED reads too (one array asked for): a read's line shift comes from its first output
// slot (win_offsets[r] mod 16) instead of r*W, its window count from the offsets; a read's last pass writes what is left of
// it.  No shared-line merge, no prefetch of the next tile (the tile's geometry is not known that early).
template <bool ALIGNED, bool RG = false>
struct SinkWindowsT {
    static_assert(!RG || ALIGNED, "the ragged ring is a mode of the line-aligned sink");
    // (round 6) a tile with an invalid byte: fast path + marks (kmx_scan_kernel.h, SinkMarksDirty) -- launch_windows_* ends with the
    // sweep that writes the spoiled windows' slots as the iterator leaves them, zeros (kmx_sweep.hip, ZERO); rolled per lane such a
    // tile cost 6 tiles' worth, and 2 % dirty reads the materialise 37 % (profiles/r06_windows_dirty.txt)
    static constexpr bool kMarksDirty = true;
    static constexpr bool kMarksCoarse = true;     // (all reads of a dirty tile: kmx_scan_kernel.h, SinkMarksCoarse)
    static constexpr u32 PITCH = 17;                        // u64 per lane row (16 + 1 pad: conflict-free both ways)
    static constexpr u32 PLANE = 64u * PITCH * 2u;          // dwords of one staged u64 array of a wave
    // staging sized by what the caller asked for (with all three u64 planes a block holds 108 KB = one block per CU and one
    // wave per SIMD; canonical words alone: 9 KB per wave, three blocks per CU)
// (the stores of the line-aligned write-back, and of the staged one when W is a multiple of 8 (whole or half lines), carry the nt hint: canon-only k = 31 4.9 -> 4.3 ms per 2e7 reads (smaller pieces must NOT: the L2 merges those))
// (the line-aligned write-back loop stays rolled (unrolled by 4: 253-256 registers and up to 276 bytes of spills in the window loop; rolled: 182, none -- canon-only materialise at k = 31 5.4 -> 4.9 ms per 2e7 reads))
// (... and so does the staged (several arrays / flags) write-back loop (fw + rc + canon + flags: 18.3-19.0 -> 17.1 ms per 2e7 reads))
    // store latency is all this sink waits for: occupancy over registers (the line-aligned variant spills at 168 registers
    // and its 17 KB ring per wave caps a CU at two blocks anyway)
    static constexpr int kWaves = ALIGNED ? 2 : 3;
    // (the 16-word frame: two waves as well -- without the prefetch it needs 230 registers -- and, with the packed tile
    // living in the ring, two blocks per CU up to 256 bases: 2.5 -> 4.5 TB/s)
    static constexpr int kWavesBig = ALIGNED ? 2 : 1;
    static constexpr bool kAliasPacked = ALIGNED;
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
        if (line_aligned(p)) return 64u * RPITCH * 2u + (RG ? 64u * 3u : 0u);   // (RG: first slot and window count of the tile's 64 reads, behind the ring)
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
    static constexpr bool kRagged = !ALIGNED || RG;
    __device__ SinkWindowsT(const WindowsParams& p_, u32, u32 W_, u32*, u32 lane_, u32* block_lds, u32 tid)
        : p(p_), base(0), W(W_), next(0), lane(lane_) {
        u32* mine = block_lds + (tid >> 6) * wave_dwords(p_);
        Tfw = reinterpret_cast<u64*>(mine);
        Trc = Tfw + (p_.fw ? PLANE / 2u : 0u);
        Tcn = Trc + (p_.rc ? PLANE / 2u : 0u);
        TF = reinterpret_cast<uint8_t*>(Tcn + (p_.canon ? PLANE / 2u : 0u));
        WOL = reinterpret_cast<u64*>(TF + (p_.flags ? 64u * 16u : 0u));
        if constexpr (RG) WOL = Tfw + 64u * RPITCH;
        NWL = reinterpret_cast<u32*>(WOL + 64);
        nwr = W_;
        out1 = line_aligned(p_) ? (p_.fw ? p_.fw : p_.rc ? p_.rc : p_.canon) : nullptr;
        want_fw = p_.fw != nullptr;
        want_rc = !want_fw && p_.rc != nullptr;
    }
    // line-aligned mode: the one array's word of a window -- two wave-uniform masks decide, no branch per window
    bool want_fw, want_rc;
    __device__ __forceinline__ u64 pick(bool lt, u64 fw, u64 rc) const { return ((lt || want_fw) && !want_rc) ? fw : rc; }
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
        if constexpr (ALIGNED) {
            Tfw[lane * RPITCH + (o & 31u)] = pick(lt, fw, rc);
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
        if constexpr (ALIGNED) {
            // 16 lanes per read and line, four reads per step.  The tile's first read is a multiple of 64, so read*W mod 16 comes
            // from the read's index in the tile, and every address is a 32-bit offset from a wave-uniform base.  The loop is
            // bound by the LDS round trip of each step (two waves per SIMD): the ring read of step i + 1 is issued before
            // the store of step i.
            typedef typename std::conditional<RG, u64, u32>::type goff_t;     // byte offset of a slot from the tile's first
            bool last = o0 + cnt == W;      // the last pass also writes what is left of each read (< 32 windows)
            if constexpr (RG) last = __any(nwr > o0 && nwr <= o0 + cnt);    // (ragged: some read of the tile ends in this pass)
            const u32 g = lane >> 4, s = lane & 15u, Wm = W & 15u;
            uint8_t* const gbase = reinterpret_cast<uint8_t*>(RG ? out1 + WOL[0] : out1 + read0 * W);
            for (u32 sub = 0; sub < (last ? 2u : 1u); ++sub) {
                auto prep = [&](u32 it, u32& at, goff_t& goff) -> bool {
                    const u32 r = 4u * it + g;
                    if constexpr (RG) {
                        // the read's own line shift and window count; its slots relative to the tile's first, in 64 bits (a rolled
                        // tile may hold reads of any length)
                        const u64 s0 = WOL[r];
                        const u32 Wr = NWL[r], a = (u32)s0 & 15u;
                        const u32 lo = o0 > a ? o0 - a : 0u;
                        const u32 hi = o0 + cnt >= Wr ? Wr : o0 + 16u - a;
                        const u32 o = lo + 16u * sub + s;
                        at = r * RPITCH + (o & 31u);
                        goff = (s0 - WOL[0] + o) * 8u;
                        return o < hi;
                    }
                    const u32 a = (r * Wm) & 15u;          // read*W mod 16
                    const u32 lo = o0 > a ? o0 - a : 0u;
                    u32 hi = last ? W : o0 + 16u - a;
                    if (merge) {
                        // the line that read r shares with read r - 1 (its head: windows [0, 16 - a)) and the one it shares with
                        // read r + 1 (its tail: the last (a + W) mod 16 windows) are written whole by heads_done(); the two at the
                        // ends of the tile are shared with other tiles and go out in pieces here
                        if (o0 == 0u && a != 0u && r != 0u) hi = 0u;
                        if (last && r != 63u) hi = W - ((a + Wm) & 15u);
                    }
                    const u32 o = lo + 16u * sub + s;
                    at = r * RPITCH + (o & 31u);
                    goff = (r * W + o) * 8u;
                    return o < hi;
                };
                // (merged: a read's part of the last pass is one line or none -- only the tile's last read, whose tail goes out
                // in pieces, has windows left for the second round)
                const u32 it_first = (merge && sub == 1u) ? 15u : 0u;
                u32 at0;
                goff_t go0;
                bool c0 = prep(it_first, at0, go0);
                u64 v0 = Tfw[at0];
#pragma unroll 1
                for (u32 it = it_first; it < 16u; ++it) {
                    u32 at1 = 0;
                    goff_t go1 = 0;
                    bool c1 = false;
                    u64 v1 = 0;
                    c1 = prep((it + 1u) & 15u, at1, go1);    // (the 17th: loaded, never stored)
                    v1 = Tfw[at1];
                    // (every store of this loop, the pieces at the ends of a tile too, carries the nt hint: nt only on the whole lines
                    // measured like no nt at all, 4.93 against 4.14-4.30 ms)
                    if (c0) __builtin_nontemporal_store(v0, reinterpret_cast<u64*>(gbase + go0));
                    c0 = c1;
                    v0 = v1;
                    go0 = go1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            return;
        }
        if (!p.win_offsets && (W & 7u) == 0u && cnt == 16u) {   // uniform reads, W a multiple of 8: every 16-window run is a whole line or two half lines (W = 120: 17.5 -> 15.7 ms for all four arrays; at W = 130, 16-byte aligned runs, nt costs 40 %)
#pragma unroll 1
            for (u32 it = 0; it < 16u; ++it) {
                const u32 idx = it * 64u + lane, r = idx >> 4, sw = idx & 15u;
                const u64 slot = (read0 + r) * W + o0 + sw;
                const u32 at = r * PITCH + sw;
                if (p.fw) __builtin_nontemporal_store(Tfw[at], &p.fw[slot]);
                if (p.rc) __builtin_nontemporal_store(Trc[at], &p.rc[slot]);
                if (p.canon) __builtin_nontemporal_store(Tcn[at], &p.canon[slot]);
                if (p.flags) p.flags[slot] = TF[r * 16u + sw];
            }
        } else {
#pragma unroll 1
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
            if (!ALIGNED || RG) { WOL[lane] = base; NWL[lane] = nwr; }   // for the transposed write-back (block_done starts with a wave barrier)
        } else {
            base = read * W;
        }
        next = 0;
        merge = ALIGNED && !RG && W >= 32u;
    }
    // ---- line-aligned mode: the output line two neighbouring reads of a tile share.  Written in two pieces (the tail of read
    // r - 1 in the tile's last pass, the head of read r in its first), 16-byte multiples at W = 130, such a line costs ~3.4x a
    // whole one: 3.5 TB/s at W = 110 / 130 / 138 against 4.7 at W = 120 (pieces of 64 bytes) and W = 128 (none).  So the kernel
    // hands the first 16 windows of every read over once more after the last block (kRedoHead: ~10 instructions per window,
    // this sink waits for its stores) -- staged in the ring slots next to the tail's, (W + o) mod 32 -- and the lines go out whole.
    static constexpr bool kRedoHead = ALIGNED;
    static constexpr bool kPrefetch = ALIGNED;   // (the staged variant is laid out for three waves: fifty more registers spill)
    bool merge;
    __device__ __forceinline__ bool wants_heads() const { return merge; }
    __device__ __forceinline__ void head(u32 o, u64 fw, u64 rc) {
        Tfw[lane * RPITCH + ((W + o) & 31u)] = pick(fw < rc, fw, rc);
    }
    __device__ __forceinline__ void heads_done(u64 read0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const u32 g = lane >> 4, s = lane & 15u, Wm = W & 15u;
        uint8_t* const gbase = reinterpret_cast<uint8_t*>(out1 + read0 * W);
        auto prep = [&](u32 it, u32& at, u32& goff) -> bool {
            const u32 r = 4u * it + g + 1u;
            const u32 a = (r * Wm) & 15u;
            // slot s of the line: below a the tail of read r - 1 (window W - a + s), from a on the head of read r (window s - a)
            at = s < a ? (r - 1u) * RPITCH + ((W - a + s) & 31u) : r * RPITCH + ((W + s - a) & 31u);
            goff = (r * W - a + s) * 8u;
            return r < 64u && a != 0u;
        };
        u32 at0, go0;
        bool c0 = prep(0u, at0, go0);
        u64 v0 = Tfw[at0];
#pragma unroll 1
        for (u32 it = 0; it < 16u; ++it) {
            u32 at1 = 0, go1 = 0;
            bool c1 = false;
            u64 v1 = 0;
            if (it + 1u < 16u) {
                c1 = prep(it + 1u, at1, go1);
                v1 = c1 ? Tfw[at1] : 0;      // (r = 64: past the ring)
            }
            if (c0) __builtin_nontemporal_store(v0, reinterpret_cast<u64*>(gbase + go0));
            c0 = c1;
            v0 = v1;
            go0 = go1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
            if constexpr (ALIGNED) Tfw[lane * RPITCH + (o & 31u)] = 0;
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
        merge = false;           // (a rolled tile writes its pieces where they fall)
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

// The flags array alone (uniform reads, W >= 16): one byte per window, VALID | FW_CANONICAL -- two BITS per window.  A lane
// keeps the "fw < rc" bit of every window of its read in registers (three instructions per window), the tile's bits meet in
// LDS (64 x 8 dwords per mask) and go out as the tile's 64 W contiguous bytes, 16 per lane and store: which read a chunk of
// 16 bytes belongs to is one multiply-high, its bits one funnel shift (two when the chunk straddles two reads), a nibble
// becomes four flag bytes by one multiply.  A rolled tile (an invalid byte in it) sets its bits in LDS as the windows come and
// is written the same way.  Beside one launch of the line-aligned kernel per u64 array this replaces the staged transposed
// store of all four arrays where W is not a multiple of 16 (16-byte runs of the flags, 128-byte runs straddling two lines of
// the words: 2.1 TB/s at W = 130).
struct FlagsParams {
    uint8_t* flags;
    u32 magic;       // floor(2^32 / W) + 1: b / W = umulhi(b, magic) for b < 2^16
};
struct SinkFlags {
    static constexpr bool kMarksDirty = true;   // (as SinkWindowsT: the sweep behind the passes zeroes the spoiled windows' flags)
    static constexpr bool kMarksCoarse = true;
    static constexpr u32 NB = 8, BP = 9;        // mask dwords per read (W <= 256); LDS pitch (the ninth stays zero)
    static constexpr u32 kLdsDwordsPerWave = 0;
    static constexpr bool kRagged = false;
    static constexpr int kWaves = 3;            // (four: 40-60 bytes of spills)
    static u32 block_lds_dwords(const FlagsParams&) { return 4u * 2u * 64u * BP; }
    FlagsParams p;
    u32* VB;      // [64][BP] valid bits of the tile's reads
    u32* LB;      // [64][BP] fw < rc
    u32 lb[NB];
    u64 base;
    u32 W, next, lane;
    __device__ SinkFlags(const FlagsParams& p_, u32, u32 W_, u32*, u32 lane_, u32* block_lds, u32 tid) : p(p_), base(0), W(W_), next(0), lane(lane_) {
        VB = block_lds + (tid >> 6) * (2u * 64u * BP);
        LB = VB + 64u * BP;
        VB[lane * BP + NB] = 0;
        LB[lane * BP + NB] = 0;
#pragma unroll
        for (u32 j = 0; j < NB; ++j) lb[j] = 0;
    }
    __device__ __forceinline__ void fast(u32 o, u64 fw, u64 rc) {
        const u32 bit = (fw < rc ? 1u : 0u) << (o & 31u);
        // (o is a constant in the unrolled blocks -- one OR; the read's last, partial block comes with a run-time o: no indexed
        // register array, eight selects)
#pragma unroll
        for (u32 j = 0; j < NB; ++j) lb[j] |= (o >> 5) == j ? bit : 0u;
    }
    __device__ __forceinline__ void block_done(u64, u32, u32) {}
    // the tile's masks are in LDS: 4 W chunks of 16 flag bytes
    __device__ __forceinline__ void write_tile(u64 read0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        v4u* const out = reinterpret_cast<v4u*>(p.flags + read0 * W);
        auto take16 = [&](const u32* M, u32 r0, u32 o0, u32 n0) -> u32 {
            const u32 j = o0 >> 5;
            u32 x = alignbit(M[r0 * BP + j + 1u], M[r0 * BP + j], o0 & 31u);
            if (n0 < 16u) x = (x & ((1u << n0) - 1u)) | (M[(r0 + 1u) * BP] << n0);   // the chunk runs into the next read
            return x & 0xFFFFu;
        };
        for (u32 c = lane; c < 4u * W; c += 64u) {
            const u32 b = 16u * c, r0 = __umulhi(b, p.magic), o0 = b - r0 * W, n0 = W - o0;
            const u32 v = take16(VB, r0, o0, n0), l = take16(LB, r0, o0, n0) & v;
            auto bytes = [&](u32 sh) -> u32 {      // four windows -> four flag bytes
                const u32 vn = (v >> sh) & 15u, ln = (l >> sh) & 15u;
                return ((vn * 0x00204081u) & 0x01010101u) * KMX_WIN_VALID | ((ln * 0x00204081u) & 0x01010101u) * KMX_WIN_FW_CANONICAL;
            };
            __builtin_nontemporal_store(v4u{bytes(0), bytes(4), bytes(8), bytes(12)}, &out[c]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ void tile_fast_done(u32) {
#pragma unroll
        for (u32 j = 0; j < NB; ++j) {
            const u32 lo = 32u * j;
            VB[lane * BP + j] = W >= lo + 32u ? ~0u : W > lo ? (1u << (W - lo)) - 1u : 0u;    // every window of the read is valid
            LB[lane * BP + j] = lb[j];
            lb[j] = 0;
        }
        write_tile(tile_read0);
    }
    u64 tile_read0;
    __device__ __forceinline__ void begin_read(u64 read) {
        base = read * W;
        next = 0;
        tile_read0 = read - lane;
    }
    __device__ __forceinline__ void tile_slow_begin(u64 read) {
        begin_read(read);
#pragma unroll
        for (u32 j = 0; j < NB; ++j) {
            VB[lane * BP + j] = 0;
            LB[lane * BP + j] = 0;
        }
    }
    __device__ __forceinline__ void tile_slow_emit(u32 pos, u64 fw, u64 rc) {     // (a lane's own row: no atomics)
        VB[lane * BP + (pos >> 5)] |= 1u << (pos & 31u);
        if (fw < rc) LB[lane * BP + (pos >> 5)] |= 1u << (pos & 31u);
    }
    __device__ __forceinline__ void slow_block(u32) {}
    __device__ __forceinline__ void tile_slow_end() { write_tile(tile_read0); }
    // the final partial tile: byte stores, window by window
    __device__ __forceinline__ void slow(u32 pos, u64 fw, u64 rc) {
        for (; next < pos; ++next) p.flags[base + next] = 0;
        p.flags[base + pos] = (uint8_t)(KMX_WIN_VALID | (fw < rc ? KMX_WIN_FW_CANONICAL : 0u));
        next = pos + 1u;
    }
    __device__ __forceinline__ void end_read() {
        for (; next < W; ++next) p.flags[base + next] = 0;
    }
    __device__ __forceinline__ void finish(const FlagsParams&) {}
};

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

hipError_t launch_sweep_windows(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u64* fw, u64* rc, u64* canon, uint8_t* flags,
                                const u64* win_offsets, unsigned long long* queue, int n_cu, hipStream_t stream, const u64* offsets,
                                const u64* ends, bool two_words);

static hipError_t windows_uniform_passes(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u64* fw, u64* rc, u64* canon,
                                         uint8_t* flags, unsigned long long* queue, int n_cu, hipStream_t stream);
// (the passes, then the sweep over the reads they marked: every pass marks the same reads, the masks are consumed once)
hipError_t launch_windows_uniform(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u64* fw, u64* rc, u64* canon,
                                  uint8_t* flags, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled) {
    *handled = scan_domain(bases, n_reads, L, k);
    if (!*handled) return hipSuccess;
    if (hipError_t e = windows_uniform_passes(bases, n_reads, L, k, fw, rc, canon, flags, queue, n_cu, stream)) return e;
    if (k < 2u) return hipSuccess;     // (k = 1: the sinks' scan does not mark)
    return launch_sweep_windows(bases, n_reads, L, k, fw, rc, canon, flags, nullptr, queue, n_cu, stream, nullptr, nullptr, false);
}
static hipError_t windows_uniform_passes(const uint8_t* bases, u64 n_reads, u32 L, u32 k, u64* fw, u64* rc, u64* canon,
                                         uint8_t* flags, unsigned long long* queue, int n_cu, hipStream_t stream) {
    const WindowsParams p{fw, rc, canon, flags, nullptr};
    const u32 W = L - k + 1u;
    if (SinkWindowsT<true>::wants_aligned(p, W)) return dispatch<SinkWindowsT<true>>(bases, n_reads, L, k, p, queue, n_cu, stream);
    // several arrays, W not a multiple of 16 (the staged store's 128-byte runs would straddle two lines each): one pass of the
    // line-aligned kernel per u64 array, one of the flags kernel -- the reads are a seventh of what one array takes to write
    if ((W & 15u) != 0u && W >= 32u && (!flags || (reinterpret_cast<uintptr_t>(flags) & 15u) == 0u)) {
        bool first = true;
        auto again = [&]() -> hipError_t {      // (every pass owns the tile queue from zeroed heads)
            if (first) {
                first = false;
                return hipSuccess;
            }
            return hipMemsetAsync(queue, 0, 32 * 128, stream);
        };
        u64* const arr[3] = {fw, rc, canon};
        for (int a = 0; a < 3; ++a) {
            if (!arr[a]) continue;
            if (hipError_t e = again()) return e;
            const WindowsParams one{a == 0 ? fw : nullptr, a == 1 ? rc : nullptr, a == 2 ? canon : nullptr, nullptr, nullptr};
            if (hipError_t e = dispatch<SinkWindowsT<true>>(bases, n_reads, L, k, one, queue, n_cu, stream)) return e;
        }
        if (flags) {
            if (hipError_t e = again()) return e;
            const FlagsParams fp{flags, (u32)(0x100000000ull / W) + 1u};
            if (hipError_t e = dispatch<SinkFlags>(bases, n_reads, L, k, fp, queue, n_cu, stream)) return e;
        }
        return hipSuccess;
    }
    return dispatch<SinkWindowsT<false>>(bases, n_reads, L, k, p, queue, n_cu, stream);
}

// ragged reads: win_offsets[r] = slot of window 0 of read r (n_reads+1 entries); L = optional bound of the read lengths
static hipError_t windows_ragged_passes(const uint8_t* bases, const u64* offsets, const u64* win_offsets, u64 n_reads, u32 L, u32 k,
                                        u64* fw, u64* rc, u64* canon, uint8_t* flags, unsigned long long* queue, int n_cu,
                                        hipStream_t stream, const u64* ends);
hipError_t launch_windows_ragged(const uint8_t* bases, const u64* offsets, const u64* win_offsets, u64 n_reads, u32 L, u32 k,
                                 u64* fw, u64* rc, u64* canon, uint8_t* flags, unsigned long long* queue, int n_cu,
                                 hipStream_t stream, bool* handled, const u64* ends) {
    *handled = offsets && win_offsets && scan_domain_ragged(bases, L, k);
    if (!*handled) return hipSuccess;
    if (hipError_t e = windows_ragged_passes(bases, offsets, win_offsets, n_reads, L, k, fw, rc, canon, flags, queue, n_cu, stream, ends)) return e;
    if (k < 2u) return hipSuccess;
    return launch_sweep_windows(bases, n_reads, L, k, fw, rc, canon, flags, win_offsets, queue, n_cu, stream, offsets, ends, false);
}
static hipError_t windows_ragged_passes(const uint8_t* bases, const u64* offsets, const u64* win_offsets, u64 n_reads, u32 L, u32 k,
                                        u64* fw, u64* rc, u64* canon, uint8_t* flags, unsigned long long* queue, int n_cu,
                                        hipStream_t stream, const u64* ends) {
    const WindowsParams p{fw, rc, canon, flags, win_offsets};
    // one u64 array, no flags (the usual call: the canonical words): whole lines through the ring, each read shifted by its own first slot
    const int n_arr = (fw ? 1 : 0) + (rc ? 1 : 0) + (canon ? 1 : 0);
    if (n_arr == 1 && !flags) return dispatch<SinkWindowsT<true, true>, WindowsParams, NoPre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets, ends);
    // several arrays: one pass of the ring per array of words, the flags through the staged write-back alone (all four arrays of 2e7
    // trimmed 150-base reads 23.9 -> 20.6 ms, of 1.5e6 x 1 000 bases 16.4 -> 12.6 ms)
    if (n_arr >= 2) {
        bool first = true;
        u64* const arr[3] = {fw, rc, canon};
        for (int a = 0; a < 3; ++a) {
            if (!arr[a]) continue;
            if (!first) {
                if (hipError_t e = hipMemsetAsync(queue, 0, 32 * 128, stream)) return e;
            }
            first = false;
            const WindowsParams one{a == 0 ? fw : nullptr, a == 1 ? rc : nullptr, a == 2 ? canon : nullptr, nullptr, win_offsets};
            if (hipError_t e = dispatch<SinkWindowsT<true, true>, WindowsParams, NoPre, true>(bases, n_reads, L, k, one, queue, n_cu, stream, NoPre(), offsets, ends)) return e;
        }
        if (flags) {
            if (hipError_t e = hipMemsetAsync(queue, 0, 32 * 128, stream)) return e;
            const WindowsParams fo{nullptr, nullptr, nullptr, flags, win_offsets};
            if (hipError_t e = dispatch<SinkWindowsT<false>>(bases, n_reads, L, k, fo, queue, n_cu, stream, NoPre(), offsets, ends)) return e;
        }
        return hipSuccess;
    }
    return dispatch<SinkWindowsT<false>>(bases, n_reads, L, k, p, queue, n_cu, stream, NoPre(), offsets, ends);
}

}  // namespace kmx
