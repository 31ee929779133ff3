// kmx_scan_kernel.h -- the word-domain scan kernel (scan_uniform_kernel) and its launch plumbing, shared by the translation
// units that instantiate it with their sinks: kmx_scan.hip (reduce, materialise) and kmx_hist.hip (bucket histograms).
// See kmx_scan.hip for the design.  Everything here is a template or static: one copy per translation unit.
#pragma once
#include <cstdlib>
#include "kmx_device.h"

namespace kmx {

// ------------------------------------------------------------------------------------------ sinks
// A sink consumes windows.  fast(o, fw, rc): window o of the lane's read on the all-valid fast path;
// slow(pos, fw, rc): a window yielded by roll_read (invalid ones are skipped); begin/end bracket one read
// on the slow path; tile_fast_done(nwin) closes a fast tile.

// ----------------------------------------------------------------------------------------- kernel
// NW  = packed dwords per read = ceil(L/16) rounded up to an instantiated size (L <= 16*NW)
// V   = 1: k in [2,17]   2: k in [18,32]   (fixes the static register index of the rc window)
// DW  = dwords per k-mer (1: k<=16, 2: k>=17)
// RAGGED: reads of different lengths, read r = bases[offsets[r], offsets[r+1]).  A tile is still 64 consecutive reads =
// one contiguous byte span, streamed from its 16-byte-aligned start; lanes carry their own start and window count,
// windows past a lane's read are masked.  A tile whose span or longest read does not fit the NW-word frame, or that
// would load past the end of the buffer, takes the per-lane rolling path.
// SinkWaves / SinkWavesBig: waves per SIMD the register allocation of a sink is sized for (default 1: hipcc otherwise spends up to 256
// VGPRs on hoisting)
// a sink may ask for a register budget of its own (static constexpr int kWaves)
template <typename S, typename = void> struct SinkWaves { static constexpr int value = 1; };
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
// a sink may ask for the first 16 windows of every read once more after the last block (static constexpr bool kRedoHead;
// wants_heads(), head(o, fw, rc), heads_done(read0)): the materialise sink writes the output line that two neighbouring reads
// share in one piece then
template <typename S, typename = void> struct SinkRedoHead { static constexpr bool value = false; };
template <typename S> struct SinkRedoHead<S, decltype((void)S::kRedoHead)> { static constexpr bool value = S::kRedoHead; };
// a sink that stores to global memory in bulk may ask for the NEXT tile's bytes early (static constexpr bool kPrefetch).  On
// gfx9 loads and stores share one in-order counter: a tile's loads, issued behind the 150 stores of the tile before, return
// when the last of those has been acknowledged -- every tile began with a drain of the wave's store queue (materialise, one
// array: 2.04 ms per 1e7 reads against 1.73 with the loads taken out).  With kPrefetch the loads of tile t + 1 are issued once
// tile t has built its F / G words -- behind the stores of tile t - 1 only, which have had a whole load-and-encode phase to
// drain -- and are encoded into ten registers right after the first window block, before that block's stores.
template <typename S, typename = void> struct SinkPrefetch { static constexpr bool value = false; };
template <typename S> struct SinkPrefetch<S, decltype((void)S::kPrefetch)> { static constexpr bool value = S::kPrefetch; };
// ... and for the 16-word frame (static constexpr int kWavesBig; without it: no cap -- most sinks would spill 1 KB there)
template <typename S, typename = void> struct SinkWavesBig { static constexpr int value = 1; };
template <typename S> struct SinkWavesBig<S, decltype((void)S::kWavesBig)> { static constexpr int value = S::kWavesBig; };
// a sink whose block-level LDS is one region per wave, empty between two tiles (static constexpr bool kAliasPacked;
// wave_dwords(params) >= the packed tile), lends it to the tile's packed words: they are dead once the lanes hold their F / G
// words, and a wave's LDS operations complete in order
template <typename S, typename = void> struct SinkAliasPacked { static constexpr bool value = false; };
template <typename S> struct SinkAliasPacked<S, decltype((void)S::kAliasPacked)> { static constexpr bool value = S::kAliasPacked; };
// a sink whose windows a later pass can take back (static constexpr bool kMarksDirty; round 6): a tile that holds an invalid byte is
// then NOT rolled per lane -- it takes the fast path as it is, an invalid byte counting as the base its bits (b >> 1) & 3 spell, and
// the reads that touch a bad chunk are marked in the array behind queue[515] exactly as the bit-sliced scan marks them; what the
// windows with an invalid byte added is subtracted by sweep_flagged_kernel (kmx_sweep.hip).  The bucket histograms: a rolled tile
// cost 2-3 tiles, and with an N in 2 % of the reads 73 % of the tiles rolled (+62 %: profiles/r05_dirty_bench.txt).
// The reads of a tile that touch a chunk with an invalid byte (a sink that marks, below): the tile's chunks once more (they are in the L2),
// one ballot per row; every lane keeps the two rows' ballots its read's chunks lie in (a chunk shared by two reads marks both: the sweep
// looks at the bytes).  NOT inlined: the line-aligned window sinks run at 249-255 registers, and inline this block's temporaries were 40
// registers spilled in their window loop (round 6) -- as a call it costs the tiles that take it, and nobody else.
template <int NW>
__device__ __attribute__((noinline)) u64 mark_dirty_rows(const uint4* __restrict__ tb, u32 chunks, u32 rd_off, u32 rd_len) {
    const u32 lane = threadIdx.x & 63u;
    const u32 c0 = rd_off >> 4, c1 = rd_len ? (rd_off + rd_len - 1u) >> 4 : c0, q0 = c0 >> 6, b0 = c0 & 63u;
    u64 lo = 0, hi = 0;
#pragma unroll 1
    for (int it = 0; it < NW; ++it) {
        const u32 c = it * 64u + lane;
        uint4 wv = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
        if (c < chunks) wv = tb[c];
        u32 rb = 0;
        (void)encode16(wv, rb);
        const u64 row = __ballot(chunk_has_invalid(rb));
        lo = q0 == (u32)it ? row : lo;
        hi = q0 + 1u == (u32)it ? row : hi;
    }
    const u64 bits = b0 ? ((lo >> b0) | (hi << (64u - b0))) : lo;
    return __ballot(rd_len != 0u && (bits & ((1ull << (c1 - c0 + 1u)) - 1ull)) != 0ull);
}

template <typename S, typename = void> struct SinkMarksDirty { static constexpr bool value = false; };
template <typename S> struct SinkMarksDirty<S, decltype((void)S::kMarksDirty)> { static constexpr bool value = S::kMarksDirty; };
// ... and a sink that marks COARSELY (static constexpr bool kMarksCoarse): every read of a dirty tile -- one store of a constant, nothing
// computed -- and leaves it to the sweep to look at the bytes.  The line-aligned window sinks: at 249-255 registers they have none to
// spare for finding the reads (inline or as a call, the block cost the CLEAN materialise 15-18 %: profiles/r06_windows_dirty.txt).
template <typename S, typename = void> struct SinkMarksCoarse { static constexpr bool value = false; };
template <typename S> struct SinkMarksCoarse<S, decltype((void)S::kMarksCoarse)> { static constexpr bool value = S::kMarksCoarse; };
template <typename S, int NW> constexpr int sink_waves() { return NW <= 10 ? SinkWaves<S>::value : SinkWavesBig<S>::value; }
template <int NW, int V, int DW, typename Sink, typename Params, bool RAGGED = false>
__global__ void __launch_bounds__(256, (sink_waves<Sink, NW>()))
scan_uniform_kernel(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 k, Params params,
                    unsigned long long* __restrict__ queue, const u64* __restrict__ offsets, u32 lead, const u64* __restrict__ ends_arg) {
    // RAGGED: read r = bases[offsets[r], ends[r]); ends_arg == nullptr: the reads lie back to back (ends = offsets + 1).  A
    // separate ends array serves reads that OVERLAP in memory: the segments a long read is cut into (materialise, round 4).
    [[maybe_unused]] const u64* __restrict__ const ends = RAGGED ? (ends_arg ? ends_arg : offsets + 1) : nullptr;
    // `lead` (uniform reads whose first byte is not 16-byte aligned): `bases` is the aligned address below it, read r
    // starts at byte lead + r*L and a tile spans one more chunk (as a ragged tile streamed from its aligned start does)
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 lane = threadIdx.x & 63u;
    const u32 wib = threadIdx.x >> 6;
    const u32 chunks_u = RAGGED ? 64u * NW : 4u * L + (lead != 0u ? 1u : 0u);  // 16-byte chunks per 64-read tile (ragged: the most a tile may span)
    const u32 ldsw = (chunks_u + 1u + 6u + 3u) & ~3u; // front pad 1, tail pad >= 6
    constexpr bool ALIAS = SinkAliasPacked<Sink>::value;
    u32* P;
    if constexpr (ALIAS) P = lds + wib * Sink::wave_dwords(params);
    else P = lds + wib * (ldsw + Sink::kLdsDwordsPerWave);

    const u64 n_full = n_reads >> 6;
    const u64 wave_id = (u64)blockIdx.x * 4u + wib;
    const u64 total_bytes = RAGGED ? ends[n_reads - 1u] : 0;

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
    Sink sink(params, k, nwin, P + ldsw, lane, ALIAS ? lds : lds + (4u * (ldsw + Sink::kLdsDwordsPerWave) + BAL - 1u) / BAL * BAL, threadIdx.x);

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
    // (a grid of fewer than 256 blocks -- a small batch -- spreads over all 32 heads too: crowded on gridDim / 8 of them, most waves found their
    // head drained at once and walked the others in step, one round trip per head: 1e4 reads took longer than 1e5)
    u32 qid = ((blockIdx.x & 255u) * NQ) / (gridDim.x < 256u ? gridDim.x : 256u);
    u32 heads_left = NQ;   // non-zero: some head may still hold a ticket
    bool rot = true;   // every ticket from the next head until the first head is seen exhausted (kmx_bitslice_kernel.h: the heads keep pace, the tiles in flight stay close together)
    // (round 5) a head seen drained: the 32 counters at a glance -- lane i reads head i, coherently -- and the next ticket from the nearest
    // head that still holds one, instead of a sweep of the heads with one synchronous device atomic each (~40 us at the end of every
    // launch: kmx_bitslice_kernel.h, profiles/r05_small_batches.txt)
    auto dequeue = [&]() -> u64 {
        while (heads_left != 0u) {
            unsigned long long v = 0;
            if (lane == 0) v = atomicAdd(queue + qid * 16u, 1ull);
            const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
            const u64 t = (((u64)hi << 32) | lo) * NQ + qid;
            if (t < n_full) {
                if (rot) qid = (qid + 1u) & (NQ - 1u);
                return t;
            }
            rot = false;
            // (lanes 32..63 look at the heads again -- same answer, no branch)
            const u32 ln = lane & (NQ - 1u);
            const u64 c = __hip_atomic_load(queue + ln * 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 live = (u32)__ballot(c < (1ull << 58) && c * NQ + ln < n_full);
            if (live == 0u) {
                heads_left = 0u;
                break;
            }
            // (the nearest live head counted from a place that differs from wave to wave: the waves that fail together do not all fall on one head)
            const u32 at = (qid + (u32)wave_id) & (NQ - 1u);
            const u32 from = (live >> at) | (at ? live << (NQ - at) : 0u);             // bit i: head at + i
            qid = (at + (u32)__builtin_ctz(from)) & (NQ - 1u);
        }
        return ~0ull;
    };
    // (round 6) Uniform reads, no prefetching sink: the ticket of the NEXT tile is requested at the top of an iteration and looked at when
    // the iteration ends -- dequeue() reads the atomic's return at once, a device round trip at the top of every tile that the other
    // waves of the SIMD have to cover.  (Ragged reads and the prefetching materialise sink want the next tile's number early: its
    // offsets / its bytes are requested under this tile's windows.)  BOTH words of the return stay live until then: with the high word
    // dead hipcc hands its register out behind an s_waitcnt vmcnt(0) (kmx_bitslice_kernel.h, "the atomic's return").
    constexpr bool ASYNC_TICKET = !RAGGED && !(SinkPrefetch<Sink>::value && NW <= 10);
    unsigned long long pend = 0;
    u32 pend_qid = 0;
    bool pend_any = false;
    auto ticket_issue = [&]() {
        pend_any = heads_left != 0u;
        pend_qid = qid;
        if (pend_any && lane == 0) {
            unsigned long long one = 1ull;
            u32 zero = 0;        // (the head's address through a VGPR the compiler cannot see through: a wave-uniform address lets hipcc's atomic optimizer rewrite the add and wait for its result at once)
            asm volatile("" : "+v"(one), "+v"(zero));
            pend = atomicAdd(queue + qid * 16u + zero, one);
        }
        if (rot) qid = (qid + 1u) & (NQ - 1u);
    };
    auto ticket_take = [&]() -> u64 {
        if (!pend_any) return ~0ull;
        const u32 lo = __builtin_amdgcn_readfirstlane((u32)pend), hi = __builtin_amdgcn_readfirstlane((u32)(pend >> 32));
        const u64 t = (((u64)hi << 32) | lo) * NQ + pend_qid;
        if (t < n_full) return t;
        rot = false;
        qid = (pend_qid + 1u) & (NQ - 1u);   // that head is drained: on, synchronously (rare)
        return dequeue();
    };
    constexpr bool MARK = SinkMarksDirty<Sink>::value;
    [[maybe_unused]] u64* const dirty_masks = (MARK && !SinkMarksCoarse<Sink>::value) ? reinterpret_cast<u64*>(queue[515]) : nullptr;
    [[maybe_unused]] u32 n_marked = 0;
    u64 next_tile = dequeue();
    constexpr bool PF = SinkPrefetch<Sink>::value && !RAGGED && NW <= 10;   // (the 16-word frame: 80 more registers do not fit two waves)
    [[maybe_unused]] u32 E[NW];          // PF: the next tile, encoded (chunk it * 64 + lane)
    [[maybe_unused]] u32 bad_pf = 0;
    [[maybe_unused]] uint4 wpf[NW];
    auto pf_issue = [&](u64 t) {
        const uint4* __restrict__ nb = reinterpret_cast<const uint4*>(bases + t * 64u * (u64)L);
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const u32 c = it * 64u + lane;
            if (c < chunks_u) wpf[it] = nb[c];
        }
    };
    auto pf_take = [&]() {
        bad_pf = 0;
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const u32 c = it * 64u + lane;
            if (c < chunks_u) E[it] = encode16(wpf[it], bad_pf);
        }
    };
    if constexpr (PF) {
        if (next_tile < n_full) {
            pf_issue(next_tile);
            pf_take();
        }
    }
    // ragged: this lane's [start, end) of the NEXT tile, fetched one tile ahead so that the tile's byte loads never wait
    // behind a dependent offsets load
    u64 nx_off = 0, nx_end = 0;
    if (RAGGED && next_tile < n_full) {
        nx_off = offsets[next_tile * 64u + lane];
        nx_end = ends[next_tile * 64u + lane];
    }
    for (u64 tile = next_tile; tile < n_full; tile = ASYNC_TICKET ? (next_tile = ticket_take()) : next_tile) {
        if constexpr (ASYNC_TICKET) ticket_issue();
        else next_tile = dequeue();
        const u64 read = tile * 64u + lane;
        const uint4* __restrict__ tb = reinterpret_cast<const uint4*>(bases + tile * 64u * (u64)L);
        u64 my_off = 0;
        u32 my_len = 0;
        bool tile_fits = true;
        if constexpr (RAGGED) {
            my_off = nx_off;
            my_len = (u32)(nx_end - nx_off);
            if (read_too_long(nx_end - nx_off, queue + KMX_TOOLONG_FROM_QUEUE)) my_len = 0u;   // (not scanned; kmx_ctx_synchronize reports it)
            // (the builtins return int: through u32 first, or offsets >= 2^31 get sign-extended into the high word)
            const u32 t0l = __builtin_amdgcn_readfirstlane((u32)nx_off), t0h = __builtin_amdgcn_readfirstlane((u32)(nx_off >> 32));
            const u32 t1l = __builtin_amdgcn_readlane((u32)nx_end, 63), t1h = __builtin_amdgcn_readlane((u32)(nx_end >> 32), 63);
            const u64 t0 = ((u64)t0h << 32) | t0l, t1 = ((u64)t1h << 32) | t1l;
            if (next_tile < n_full) {
                nx_off = offsets[next_tile * 64u + lane];
                nx_end = ends[next_tile * 64u + lane];
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
        u32 bad = 0;
        if constexpr (PF) {
            // ---- 1 + 2. the tile came in while the tile before was being written out
            bad = bad_pf;
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const u32 c = it * 64u + lane;
                if (c < chunks) P[1u + c] = E[it];
            }
        } else {
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
        if (tile_fits) {
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const u32 c = it * 64u + lane;
                if (c < chunks) P[1u + c] = encode16(w[it], bad);
            }
        }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        bool roll_tile = !tile_fits || __any(chunk_has_invalid(bad));
        if constexpr (MARK && SinkMarksCoarse<Sink>::value) {
            // (nothing of this is carried across the tile loop -- no pointer, no counter: the mask array's address is read here, and
            // what the sweep is told is "many", by a plain store that every marking wave agrees on; it fields all its waves then)
            if (roll_tile && tile_fits) {
                const unsigned long long dq = queue[515];
                u64* const dmk = reinterpret_cast<u64*>(((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(dq >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((u32)dq));
                if (dmk != nullptr) {
                    if (lane == 0) {      // (all 64: a read without a window costs the sweep nothing)
                        dmk[tile] = ~0ull;
                        queue[512] = 1ull << 40;
                    }
                    roll_tile = false;
                }
            }
        } else if constexpr (MARK) {
            if (roll_tile && tile_fits && dirty_masks != nullptr) {
                // which reads touch a chunk with an invalid byte?  The tile's chunks once more (they are in the L2), one ballot per row;
                // every lane keeps the two rows' ballots its read's chunks lie in (a chunk shared by two reads marks both: the sweep looks
                // at the bytes)
                const u32 rd_off = posF - 16u, rd_len = RAGGED ? my_len : L;     // the read's bytes, relative to the tile's aligned start
                const u64 dm = mark_dirty_rows<NW>(tb, chunks, rd_off, rd_len);
                if (lane == 0 && dm != 0ull) dirty_masks[tile] = dm;
                n_marked += (u32)__builtin_popcountll(dm);
                roll_tile = false;
            }
        }
        if (roll_tile) {
            // ---- rare: a non-ACGTacgt byte somewhere in this tile (or a ragged tile outside the frame) -> exact iterator semantics
            sink.tile_slow_begin(read);
            u32 roll_max = L;
            if constexpr (RAGGED) roll_max = (u32)wave_max_u32(my_len);
            roll_read_stepped(RAGGED ? bases + my_off : bases + lead + read * (u64)L, RAGGED ? my_len : L, roll_max, k,
                              [&](u32 pos, u64 fw, u64 rc) { sink.tile_slow_emit(pos, fw, rc); }, [&](u32 wb) { sink.slow_block(wb); });
            sink.tile_slow_end();
            if constexpr (PF) {
                if (next_tile < n_full) {
                    pf_issue(next_tile);
                    pf_take();
                }
            }
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
        if constexpr (PF) {
            if (next_tile < n_full) pf_issue(next_tile);
        }
#pragma unroll
        for (int i = 0; i <= NW - V; ++i) {
            const int M = NW - V - i;
            if ((u32)i < imax) {
                // Opaque copies of the six source words, made INSIDE the block: LLVM's speculative execution otherwise
                // hoists the (cheap, side-effect-free) funnel shifts of every block above the chain of uniform branches
                // and keeps them all live -- 228-256 VGPRs, 1-2 waves per SIMD instead of 4.
                u32 f0 = F[i], f1 = F[i + 1], f2 = F[i + 2], g0 = G[M], g1 = G[M + 1], g2 = G[M + 2];
                asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(g0), "+v"(g1), "+v"(g2));
                if constexpr (RAGGED && SinkBatch16<Sink>::value) {
                    // a sink that takes the windows of a block in batches (the partitioned histogram: its LDS round trips, paid
                    // once per batch instead of once per window, are what the ragged scan otherwise runs at): a second,
                    // unmasked copy of the block for the blocks that every read of the tile owns in full
                    if (16u * i + 16u <= nwin_min) {
#pragma unroll
                        for (int s = 0; s < 16; ++s) window(16 * i + s, f0, f1, f2, g0, g1, g2, 2 * s, 30 - 2 * s, false, s);
                    } else {
#pragma unroll
                        for (int s = 0; s < 16; ++s) window(16 * i + s, f0, f1, f2, g0, g1, g2, 2 * s, 30 - 2 * s, true, -1);
                    }
                } else {
                // (a second, unmasked copy of the block for tiles of equal-length reads doubles the code past the
                //  instruction cache and costs more than the per-window mask it saves)
#pragma unroll
                for (int s = 0; s < 16; ++s) window(16 * i + s, f0, f1, f2, g0, g1, g2, 2 * s, 30 - 2 * s, RAGGED, RAGGED ? -1 : s);
                }
                if constexpr (PF) {
                    if (i == 0 && next_tile < n_full) pf_take();    // (before the block's stores: nothing younger than the loads to wait for)
                }
                sink.block_done(tile * 64u, 16u * i, 16u);
            } else if ((u32)i == imax) {
                for (u32 s = 0; s <= smax; ++s) window(16u * i + s, F[i], F[i + 1], F[i + 2], G[M], G[M + 1], G[M + 2], 2u * s, 30u - 2u * s);
                if constexpr (PF) {
                    if (i == 0 && next_tile < n_full) pf_take();
                }
                sink.block_done(tile * 64u, 16u * i, smax + 1u);
            }
        }
        if constexpr (SinkRedoHead<Sink>::value && !RAGGED) {
            if (sink.wants_heads()) {
                constexpr int M0 = NW - V;
                u32 f0 = F[0], f1 = F[1], f2 = F[2], g0 = G[M0], g1 = G[M0 + 1], g2 = G[M0 + 2];
                asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(g0), "+v"(g1), "+v"(g2));
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    u64 fw, rc;
                    if (DW == 2) {
                        fw = ((u64)(alignbit(f2, f1, 2 * s) & mhi) << 32) | alignbit(f1, f0, 2 * s);
                        rc = ((u64)(alignbit(g2, g1, 30 - 2 * s) & mhi) << 32) | alignbit(g1, g0, 30 - 2 * s);
                    } else {
                        fw = (u64)(alignbit(f1, f0, 2 * s) & mlo);
                        rc = (u64)(alignbit(g1, g0, 30 - 2 * s) & mlo);
                    }
                    sink.head((u32)s, fw, rc);
                }
                sink.heads_done(tile * 64u);
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
            const u64 o0 = offsets[read], len64 = ends[read] - o0;
            if (!read_too_long(len64, queue + KMX_TOOLONG_FROM_QUEUE))
                roll_read(bases + o0, (u32)len64, k, [&](u32 pos, u64 fw, u64 rc) { sink.slow(pos, fw, rc); });
        } else {
            roll_read(bases + lead + read * (u64)L, L, k, [&](u32 pos, u64 fw, u64 rc) { sink.slow(pos, fw, rc); });
        }
        sink.end_read();
    }
    if constexpr (MARK) {
        if (n_marked != 0u && lane == 0) atomicAdd(queue + 512, (unsigned long long)n_marked);   // (how many waves the sweep fields)
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
                             unsigned long long* queue, int n_cu, hipStream_t stream, Pre pre = Pre(), const u64* offsets = nullptr,
                             const u64* ends = nullptr) {
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
    size_t lds_bytes = (size_t)((4u * (ldsw + Sink::kLdsDwordsPerWave) + BAL - 1u) / BAL * BAL) * 4u + (size_t)Sink::block_lds_dwords(params) * 4u;
    if constexpr (SinkAliasPacked<Sink>::value) {
        if (Sink::block_lds_dwords(params) < 4u * ldsw) return hipErrorInvalidValue;
        lds_bytes = (size_t)Sink::block_lds_dwords(params) * 4u;
    }
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
    const u64 need = (n_tiles + 3u) / 4u;
    if (grid > need) grid = need;
    if (grid == 0) grid = 1;
    if (!pre(grid)) return hipErrorOutOfMemory;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds_bytes, stream, bases, n_reads, L, k, params, queue, offsets, lead, ends);
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
template <typename SinkT, typename Params, typename Pre = NoPre, bool ONLY_RAGGED = false>
static hipError_t dispatch(const uint8_t* bases, u64 n_reads, u32 L, u32 k, Params p, unsigned long long* queue,
                           int n_cu, hipStream_t stream, Pre pre = Pre(), const u64* offsets = nullptr, const u64* ends = nullptr) {
    const bool big = L > 160 || (offsets && L == 0);
    if constexpr (!SinkT::kRagged) {
        if (offsets) return hipErrorInvalidValue;
    } else if (offsets) {
        if (k <= 16) {
            if (big) return launch_one<16, 1, 1, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets, ends);
            return launch_one<10, 1, 1, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets, ends);
        }
        if (k == 17) {   // the one k whose rc window sits at the V = 1 register index while the k-mer needs two dwords
            if (big) return launch_one<16, 1, 2, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets, ends);
            return launch_one<10, 1, 2, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets, ends);
        }
        if (big) return launch_one<16, 2, 2, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets, ends);
        return launch_one<10, 2, 2, SinkT, Params, Pre, true>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets, ends);
    }
    if constexpr (ONLY_RAGGED) return hipErrorInvalidValue;   // (a sink instantiated for reads behind an offsets array only)
    else {
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
}

}  // namespace kmx
