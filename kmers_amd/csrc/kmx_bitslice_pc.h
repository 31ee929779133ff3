// kmx_bitslice_pc.h -- K1c: the bit-sliced canonical k-mer scan with PRODUCER and CONSUMER waves (round 3).
//
// Same arithmetic as scan_bitsliced_kernel (kmx_bitslice_kernel.h: phases A-D, the closed form of the epilogue; the loop it
// replaces is CanonicalKmerIterator::find_next, /root/reference/src/naive_impl/canonical_kmer_iterator.rs:42-70), for uniform
// ASCII reads and single-word k.  What changes is who does what:
//   * a PRODUCER wave streams tiles from HBM and runs phases A-C (encode + validate, realign, 32x32 bit transposes); the
//     planes of a tile go to one of the block's plane buffers in LDS;
//   * a CONSUMER wave takes a full buffer and runs phase D (the fw<rc ripples and the masked popcounts) into its counters.
// Neither kind holds the other's registers -- the 40 prefetch registers and the transposes live in the producers, the 33
// counters in the consumers -- so the kernel fits 80 (96) registers and runs 6 (5) waves per SIMD where the one-role kernel
// runs 4: more waves per SIMD is what the VALU co-issue rule of gfx950 rewards (a half-rate instruction only pairs with a
// FULL-rate instruction of ANOTHER wave).
//
// Hand-off, all in LDS, no block barrier after start-up:
//   state[b]      0 = buffer b is empty (its producer may store planes), 1 = full
//   ready[32]     ticketed ring: a producer that filled b takes tail++ and writes b+1 into its slot; a consumer takes head++
//                 and waits for its slot to become non-zero.  One slot = one consumer, so slots are cleared with a plain store.
//   valid[b]      the tile's 64-bit mask of reads that are NOT blanked (reads with an invalid byte; see the one-role kernel)
// Each producer owns BPP buffers and fills them in turn; when all producers are out of tiles the last one posts one POISON
// per consumer.  The results are sums and xors, so every wave adds its own part of the closed form to the output:
// consumers the counter terms, producers the per-plane totals and the k-mer count.
#pragma once
#include "kmx_bitslice_kernel.h"

#ifndef KMX_PC_SLEEP
#define KMX_PC_SLEEP 1   // s_sleep argument of the hand-off polls (x64 cycles)
#endif
#ifndef KMX_PC_LATE
#define KMX_PC_LATE 3    // producer: this many of the NW prefetch rows are requested after the planes are stored instead of right after phase A (0: 91 registers, spills at 80)
#endif
#ifndef KMX_PC_ABLATE
#define KMX_PC_ABLATE 0  // dev: 1 = consumers skip phase D, 2 = producers skip A-C (results become wrong)
#endif

namespace kmx {

constexpr u32 PC_RING = 32u;     // ready-ring slots (> buffers + consumers)
constexpr u32 PC_CTRL = 128u;    // dwords of control words in front of the block's LDS
constexpr u32 PC_POISON = 0xFFFFu;

// LDS geometry of one block (dwords); the kernel and the launcher both derive it from (L, lead)
template <int K, int NW, int NP, int NC, int BPP>
struct PcLayout {
    static constexpr u32 NB = (u32)(NP * BPP);
    static constexpr u32 PLANES = (u32)bs_plane_dwords(NW);
    static constexpr u32 BSZ = 2u * PLANES + 64u;                 // one plane buffer: two sets + the set-stride slack
    static constexpr u32 CSZ = 128u;                              // consumer scratch (epilogue)
    u32 ldsw, psz, prod0, cons0, buf0, total;
    __host__ __device__ explicit PcLayout(u32 chunks) {
        ldsw = (chunks + 1u + 6u + 3u) & ~3u;
        psz = ldsw + 64u * NW + 64u;                              // packed words, TOT[NW][64], the bitmap scratch of dirty tiles
        prod0 = PC_CTRL;
        cons0 = prod0 + (u32)NP * psz;
        buf0 = cons0 + (u32)NC * CSZ;
        total = buf0 + NB * BSZ;
    }
};

typedef volatile u32 __attribute__((address_space(3))) * pc_vu32p;
__device__ __forceinline__ u32 pc_ld(u32* a) { return __builtin_amdgcn_readfirstlane(*(pc_vu32p)a); }
__device__ __forceinline__ void pc_st(u32* a, u32 v) { *(pc_vu32p)a = v; }
__device__ __forceinline__ void pc_lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int K, int NW, int WPL, int NP, int NC, int WPS, int BPP, int LATE = KMX_PC_LATE>
__global__ void __launch_bounds__(64 * (NP + NC), WPS)
scan_bitsliced_pc_kernel(const uint8_t* __restrict__ bases, u64 n_reads, u32 L, u32 want_hash, u32 want_sumfw,
                         kmx_summary* __restrict__ out, unsigned long long* __restrict__ queue, u32 lead) {
    static_assert(K >= 2 && K <= 32 && WPL <= 8, "single-word k-mers");
    typedef PcLayout<K, NW, NP, NC, BPP> Lay;
    constexpr u32 NB = Lay::NB;
    static_assert(NB + NC < PC_RING && NB <= 16, "ring / control words");
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    constexpr bool ROT = (WPL == 2 || WPL == 4 || WPL == 8);
    constexpr int RW = ROT ? WPL : 4;
    constexpr int S2 = (16 * NW) / RW + 1;
    constexpr u32 PLANES = Lay::PLANES;
    constexpr int NT = (K + 1) / 2;
    const u32 lane = threadIdx.x & 63u;
    const u32 half = lane >> 5, p = lane & 31u;
    const u32 wib = (u32)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32 chunks = 4u * L + (lead != 0u ? 1u : 0u);
    const Lay lay(chunks);
    const u32 W = L - (u32)K + 1u;
    const u32 NG = (W + WPL - 1u) / WPL;
    constexpr u32 LS = ROT ? 2u : 2u * (u32)WPL;
    const u32 SP = PLANES + ((LS * NG - PLANES) & 63u);        // set stride, see KMX_BS_BANKFIX
    const u64 n_full = n_reads >> 6;
    const u64 wave_id = (u64)blockIdx.x * (u32)(NP + NC) + wib;

    u32* const c_head = lds + 0;
    u32* const c_tail = lds + 1;
    u32* const c_done = lds + 2;
    u32* const c_state = lds + 16;
    u32* const c_ready = lds + 32;
    u32* const c_valid = lds + 64;   // [NB] u64
    if (threadIdx.x < PC_CTRL) lds[threadIdx.x] = 0u;
    __syncthreads();

    auto emit_sums = [&](u64 n, u64 r0, u64 h0, u64 f) {
        if (lane == 0) {
            if (n) atomicAdd((unsigned long long*)&out->n_valid, (unsigned long long)n);
            atomicAdd((unsigned long long*)&out->sum_canon, (unsigned long long)r0);
            if (want_hash) atomicXor((unsigned long long*)&out->xor_hash, (unsigned long long)h0);
            if (want_sumfw && f) atomicAdd((unsigned long long*)&out->sum_fw, (unsigned long long)f);
        }
    };
    auto lds_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    if (wib < (u32)NP) {
        // ============================================================================ PRODUCER
        u32* const P = lds + lay.prod0 + wib * lay.psz;
        u32* const TOT = P + lay.ldsw;
        u64* const BM = reinterpret_cast<u64*>(TOT + 64u * NW);
#pragma unroll
        for (int g = 0; g < NW; ++g) TOT[64u * g + lane] = 0;
        const u32 posF = lane * L + lead + 16u;
        const u32 qF = posF >> 4, aF = 2u * (posF & 15u);
        u32 tr_sh[5], tr_keep[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const u32 d = 16u >> s;
            const u32 md = 0xFFFFFFFFu / ((1u << d) + 1u);
            tr_sh[s] = (p & d) ? d : 32u - d;
            tr_keep[s] = (p & d) ? ~md : md;
        }
        const u32 tr_sel16 = (p & 16u) ? 0x03020706u : 0x05040100u;
        const u32 tr_sel8 = (p & 8u) ? 0x03070105u : 0x06020400u;
        u32 n_bs_tiles = 0, n_blanked = 0;
        u64 valid_reads = ~0ull;

        // ---- tile queue (as in the one-role kernel: 32 heads, a ticket requested one tile ahead)
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
        u32 pend = 0, pend_qid = 0;
        auto ticket_issue = [&]() {
            pend_qid = qid;
            if (heads_left != 0u && lane == 0) {
                unsigned long long one = 1ull;
                asm volatile("" : "+v"(one));
                pend = (u32)atomicAdd(queue + qid * 16u, one);
            }
        };
        auto ticket_take = [&]() -> u64 {
            if (heads_left == 0u) return ~0ull;
            const u32 lo = __builtin_amdgcn_readfirstlane(pend);
            const u64 t = (u64)lo * NQ + pend_qid;
            if (t < n_full) return t;
            qid = (pend_qid + 1u) & (NQ - 1u);
            heads_left -= 1u;
            return dequeue();
        };

        // ---- loads: one raw buffer descriptor per tile, lanes past the tile's end read zeros
        constexpr int NLD = NW;
        uint4 w[NLD];
        const u32 lane16 = lane * 16u;
        auto issue_loads = [&](u64 t, int row0, int row1) {
            const uint8_t* tb = bases + t * 64u * (u64)L;
            u32 l16 = lane16;
            asm volatile("" : "+v"(l16));
            uint8_t* const tbu = reinterpret_cast<uint8_t*>(uniform_u64(reinterpret_cast<u64>(tb)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tbu, 0, (int)__builtin_amdgcn_readfirstlane(chunks * 16u), 0x00020000);
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                if (it < row0 || it >= row1) continue;
                typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, l16 + (u32)it * 1024u, 0, KMX_BS_LOAD_AUX);
                w[it] = make_uint4(v.x, v.y, v.z, v.w);
            }
        };
        auto prefetch = [&](u64 t, u64 fallback_t, int row0, int row1) {
            const u64 nxt = t < n_full ? t : fallback_t;
            __builtin_amdgcn_sched_barrier(0);
            issue_loads(nxt, row0, row1);
            __builtin_amdgcn_sched_barrier(0);
        };
        const u32 k55 = 0x55555555u;
        auto encode_prio = [&](const uint4& wv, u32& bad) -> u32 {
            constexpr u32 TBL_LO = 0x00430041u, TBL_HI = 0x00470054u, W4 = 0x40100401u;
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
            bad = __builtin_amdgcn_bitop3_b32(bad, e0, wv.x, 0xF6);
            bad = __builtin_amdgcn_bitop3_b32(bad, e1, wv.y, 0xF6);
            bad = __builtin_amdgcn_bitop3_b32(bad, e2, wv.z, 0xF6);
            bad = __builtin_amdgcn_bitop3_b32(bad, e3, wv.w, 0xF6);
            asm volatile("s_setprio 3\n\t"
                         "v_dot4_u32_u8 %0, %0, %4, 0\n\t"
                         "v_dot4_u32_u8 %1, %1, %4, 0\n\t"
                         "v_dot4_u32_u8 %2, %2, %4, 0\n\t"
                         "v_dot4_u32_u8 %3, %3, %4, 0\n\t"
                         "s_nop 0\n\t"
                         "v_lshl_or_b32 %0, %1, 8, %0\n\t"
                         "v_lshl_or_b32 %0, %2, 16, %0\n\t"
                         "v_lshrrev_b32 %0, 1, %0\n\t"
                         "v_lshl_or_b32 %0, %3, 23, %0\n\t"
                         "s_setprio 0"
                         : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)
                         : "s"(W4));
            return __builtin_amdgcn_bitop3_b32(t0 >> 1, t0, k55, 0x6c);
        };

        u64 tile = uniform_u64(dequeue());
        u64 next_tile = uniform_u64(dequeue());
        ticket_issue();
        if (tile < n_full) issue_loads(tile, 0, NLD);
        u32 bsel = 0;                              // which of this producer's BPP buffers comes next
        while (tile < n_full) {
            // ---- A. pack + validate the tile sitting in w[]
            u32 bad = 0;
            if (!(KMX_PC_ABLATE & 2)) {
                if (chunks >= 64u * (NW - 1)) {
#pragma unroll
                    for (int it = 0; it < NW - 1; ++it) P[1u + it * 64u + lane] = encode_prio(w[it], bad);
                    const u32 c = (NW - 1) * 64u + lane;
                    if (c < chunks) P[1u + c] = encode_prio(w[NW - 1], bad);
                } else {
#pragma unroll
                    for (int it = 0; it < NW; ++it) {
                        const u32 c = it * 64u + lane;
                        if (c < chunks) P[1u + c] = encode16(w[it], bad);
                    }
                }
            }
            const bool bad_tile = __any(chunk_has_invalid(bad));
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every row has been used; the registers are free on every path
            valid_reads = ~0ull;
            if (bad_tile) {
                // which reads hold the invalid bytes: one ballot per row, every lane looks up the chunks of its read
                u64* const masks = reinterpret_cast<u64*>(queue[515]);
                if (masks == nullptr) __builtin_trap();   // (the host side always provides the array)
#pragma unroll
                for (int it = 0; it < NW; ++it) {
                    const u32 c = it * 64u + lane;
                    u32 rb = 0;
                    (void)encode16(w[it], rb);
                    const u64 row = __ballot(c < chunks && chunk_has_invalid(rb));
                    if (lane == 0) BM[it] = row;
                }
                if (lane == 0) { BM[NW] = 0; BM[NW + 1] = 0; }
                lds_fence();
                const u32 rd_off = lane * L + lead;
                const u32 c0 = rd_off >> 4, c1 = (rd_off + L - 1u) >> 4;
                const u32 q0 = c0 >> 6, b0 = c0 & 63u;
                const u64 lo = BM[q0], hi = BM[q0 + 1u];
                const u64 bits = b0 ? ((lo >> b0) | (hi << (64u - b0))) : lo;
                const bool dirty = (bits & ((1ull << (c1 - c0 + 1u)) - 1ull)) != 0ull;
                const u64 dm = uniform_u64(__ballot(dirty));
                if (lane == 0) {
                    u32 one = 1u;
                    asm volatile("" : "+v"(one));
                    masks[tile] = dm;
                    queue[512] = one;
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);
                valid_reads = ~dm;
                n_blanked += (u32)__builtin_popcountll(dm);
                lds_fence();
            }
            prefetch(next_tile, tile, 0, NLD - LATE);
            lds_fence();
            // ---- B. this lane's read, realigned
            u32 F[NW];
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
            if (valid_reads != ~0ull) {
                u32 ln_o = lane;
                asm volatile("" : "+v"(ln_o));
                const bool blank = ((valid_reads >> ln_o) & 1ull) == 0ull;
#pragma unroll
                for (int g = 0; g < NW; ++g) F[g] = blank ? 0u : F[g];
            }
            // ---- C. 32x32 bit transposes, stage-major (see the one-role kernel)
            if (!(KMX_PC_ABLATE & 2)) {
                u32 Y[NW];
#define KMX_HRUN_BEGIN __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(3);
#define KMX_HRUN_END __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int g = 0; g < NW; ++g) Y[g] = (u32)__builtin_amdgcn_ds_swizzle((int)F[g], (16 << 10) | 0x1f);
                KMX_HRUN_BEGIN
#pragma unroll
                for (int g = 0; g < NW; ++g) F[g] = __builtin_amdgcn_perm(Y[g], F[g], tr_sel16);
                KMX_HRUN_END
#pragma unroll
                for (int g = 0; g < NW; ++g) Y[g] = (u32)__builtin_amdgcn_ds_swizzle((int)F[g], (8 << 10) | 0x1f);
                KMX_HRUN_BEGIN
#pragma unroll
                for (int g = 0; g < NW; ++g) F[g] = __builtin_amdgcn_perm(Y[g], F[g], tr_sel8);
                KMX_HRUN_END
#pragma unroll
                for (int g = 0; g < NW; ++g) Y[g] = (u32)__builtin_amdgcn_ds_swizzle((int)F[g], (4 << 10) | 0x1f);
                KMX_HRUN_BEGIN
#pragma unroll
                for (int g = 0; g < NW; ++g) Y[g] = alignbit(Y[g], Y[g], tr_sh[2]);
                KMX_HRUN_END
#pragma unroll
                for (int g = 0; g < NW; ++g) F[g] = bitsel(F[g], Y[g], tr_keep[2]);
#pragma unroll
                for (int st = 3; st < 5; ++st) {
                    KMX_HRUN_BEGIN
#pragma unroll
                    for (int g = 0; g < NW; ++g)
                        Y[g] = st == 3 ? (u32)__builtin_amdgcn_update_dpp(0, (int)F[g], 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true)
                                       : (u32)__builtin_amdgcn_update_dpp(0, (int)F[g], 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
#pragma unroll
                    for (int g = 0; g < NW; ++g) Y[g] = alignbit(Y[g], Y[g], st == 3 ? tr_sh[3] : tr_sh[4]);
                    KMX_HRUN_END
#pragma unroll
                    for (int g = 0; g < NW; ++g) F[g] = bitsel(F[g], Y[g], st == 3 ? tr_keep[3] : tr_keep[4]);
                }
                // per-plane popcount totals (this lane's slots only)
                KMX_HRUN_BEGIN
#pragma unroll
                for (int g = 0; g < NW; ++g) Y[g] = (u32)__builtin_popcount(F[g]);
                KMX_HRUN_END
                u32* const tot_l = TOT + lane;
#pragma unroll
                for (int g = 0; g < NW; ++g) atomicAdd(tot_l + 64 * g, Y[g]);
#undef KMX_HRUN_BEGIN
#undef KMX_HRUN_END
            }
            // ---- hand the planes over: wait for this producer's next buffer to be empty, store, publish
            const u32 b = wib * (u32)BPP + bsel;
            bsel = bsel + 1u == (u32)BPP ? 0u : bsel + 1u;
            while (pc_ld(c_state + b) != 0u) __builtin_amdgcn_s_sleep(KMX_PC_SLEEP);
            asm volatile("" ::: "memory");
            {
                u32* const PLb = lds + lay.buf0 + b * Lay::BSZ;
                const u32 b0 = p >> 1;
                const u32 slot0 = ROT ? (b0 % (u32)RW) * S2 + (b0 / (u32)RW) : b0;
                u32* const pst = PLb + (half * SP + 2u * slot0 + (p & 1u));
#pragma unroll
                for (int g = 0; g < NW; ++g) pst[ROT ? (32 / RW) * g : 32 * g] = F[g];
                if (lane == 0) {
                    pc_st(c_valid + 2u * b, (u32)valid_reads);
                    pc_st(c_valid + 2u * b + 1u, (u32)(valid_reads >> 32));
                    pc_st(c_state + b, 1u);
                }
                pc_lds_drain();                    // the planes are in LDS before the ticket says so
                if (lane == 0) {
                    const u32 slot = atomicAdd(c_tail, 1u);
                    pc_st(c_ready + (slot & (PC_RING - 1u)), b + 1u);
                }
            }
            if constexpr (LATE > 0) prefetch(next_tile, tile, NLD - LATE, NLD);
            n_bs_tiles += 1;
            tile = next_tile;
            next_tile = uniform_u64(ticket_take());
            ticket_issue();
        }
        // ---- out of tiles: the last producer to get here posts one POISON per consumer
        if (lane == 0) {
            const u32 d = atomicAdd(c_done, 1u);
            if (d == (u32)NP - 1u) {
                for (u32 i = 0; i < (u32)NC; ++i) {
                    const u32 slot = atomicAdd(c_tail, 1u);
                    pc_st(c_ready + (slot & (PC_RING - 1u)), PC_POISON);
                }
            }
        }
        // ---- final partial tile: per-lane rolling (one wave of the grid)
        const u32 rem = (u32)(n_reads & 63u);
        if (rem != 0u && wave_id == 0) {
            u64 fn = 0, fs = 0, fx = 0, ff = 0;
            if (lane < rem) {
                roll_read(bases + lead + (n_full * 64u + lane) * (u64)L, L, (u32)K, [&](u32, u64 fw, u64 rc) {
                    const u64 canon = fw < rc ? fw : rc;
                    fn += 1;
                    fs += canon;
                    fx ^= lex_hash(canon, (u32)K);
                    ff += fw;
                });
            }
            emit_sums(wave_sum(fn), wave_sum(fs), wave_xor(fx), wave_sum(ff));
        }
        // ---- this producer's part of the closed form: cnt(t,b) += nk - Tq[t][b]; sum of all fw words; the k-mer count
        if (n_bs_tiles != 0u) {
            const u64 nk = ((u64)n_bs_tiles * 64u - n_blanked) * (u64)W;
            u64 fwall = 0;
            u32* const PLt = P;                      // the packed region is free now: PLt[2*base + bit] = popcount total of that plane
            lds_fence();
#pragma unroll
            for (int g = 0; g < NW; ++g) {
                const u32 qidx = 32u * g + p;
                const u32 pcq = TOT[64u * g + lane];
                u64 wf, wr;
                plane_weights(qidx >> 1, L, (u32)K, wf, wr);
                fwall += (u64)pcq * (wf << (qidx & 1u));
                const u32 both = pcq + __shfl_xor(pcq, 32, WAVE);
                if (half == 0) PLt[32u * g + p] = both;
            }
            const u64 bs_fw = wave_sum(fwall);
            lds_fence();
            u64 s0 = 0, x0 = 0;
            for (u32 pid = lane; pid < 2u * (u32)K; pid += 64u) {
                const u32 t = pid >> 1, bb = pid & 1u, t2 = (u32)K - 1u - t;
                u64 tq = 0;
                for (u32 i = t2; i <= L - 1u - t; ++i) tq += PLt[2u * i + bb];
                const u64 cnt = nk - tq;
                s0 += cnt << (2u * t + bb);
                if (want_hash && (cnt & 1ull)) x0 ^= 1ull << (2u * t2 + bb);
            }
            emit_sums(nk, wave_sum(s0), wave_xor(x0), bs_fw);
        }
    } else {
        // ============================================================================ CONSUMER
        const u32 cidx = wib - (u32)NP;
        u32 D[2 * NT];
#pragma unroll
        for (int q = 0; q < 2 * NT; ++q) D[q] = 0;
        u32 mcnt = 0;
        u32 n_tiles = 0;
        const u32 gidx = lane;
        const bool active = gidx < 2u * NG;
        const u32 set = gidx >= NG ? 1u : 0u;                       // (idle lanes continue set 1's sequence: KMX_BS_BANKFIX)
        const u32 o = (u32)WPL * (gidx - set * NG);
        const u32 nwin = active ? (W - o < (u32)WPL ? W - o : (u32)WPL) : 0u;
        const u32 lane_off = set * SP + 2u * (ROT ? (o / (u32)RW) : o);   // dwords from the buffer's base to this lane's first plane
        typedef const volatile u64 __attribute__((address_space(3))) * lds_cvu64p;
#define KMX_PLANE(i) src[ROT ? (((i) % RW) * S2 + ((i) / RW)) : (i)]
        for (;;) {
            u32 slot = 0;
            if (lane == 0) slot = atomicAdd(c_head, 1u);
            slot = (u32)__builtin_amdgcn_readfirstlane(slot) & (PC_RING - 1u);
            u32 v;
            while ((v = pc_ld(c_ready + slot)) == 0u) __builtin_amdgcn_s_sleep(KMX_PC_SLEEP);
            if (lane == 0) pc_st(c_ready + slot, 0u);
            if (v == PC_POISON) break;
            asm volatile("" ::: "memory");
            const u32 b = v - 1u;
            const u32 vlo = pc_ld(c_valid + 2u * b), vhi = pc_ld(c_valid + 2u * b + 1u);
            const lds_cvu64p src = (lds_cvu64p)(reinterpret_cast<const u64*>(lds + lay.buf0 + b * Lay::BSZ + lane_off));
            n_tiles += 1;
            if (!(KMX_PC_ABLATE & 1)) {
                // ---- D, pass 1: the fw<rc ripples of this lane's WPL windows, least significant deciding pair first
                u32 lt[WPL];
#pragma unroll
                for (int w = 0; w < WPL; ++w) lt[w] = 0u;
                {
                    u64 Pv[K + WPL - 1];
                    constexpr int J0 = (K + 1) / 2 - 1;
#pragma unroll
                    for (int i = J0; i <= K - 1 - J0 + WPL - 1; ++i) Pv[i] = KMX_PLANE(i);
#pragma unroll
                    for (int d = 1; d < KMX_BS_P1D; ++d) {
                        if (J0 - d >= 0) {
                            Pv[J0 - d] = KMX_PLANE(J0 - d);
                            Pv[K - 1 - (J0 - d) + WPL - 1] = KMX_PLANE(K - 1 - (J0 - d) + WPL - 1);
                        }
                    }
#pragma unroll
                    for (int j = J0; j >= 0; --j) {
                        if (j - KMX_BS_P1D >= 0) {
                            asm volatile("" : : "v"(lt[0]) : "memory");
                            Pv[j - KMX_BS_P1D] = KMX_PLANE(j - KMX_BS_P1D);
                            Pv[K - 1 - (j - KMX_BS_P1D) + WPL - 1] = KMX_PLANE(K - 1 - (j - KMX_BS_P1D) + WPL - 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int w = 0; w < WPL; ++w) {
                            const int ia = K - 1 - j + w, iq = j + w;
                            const u32 a0 = (u32)Pv[ia], a1 = (u32)(Pv[ia] >> 32);
                            const u32 q0 = (u32)Pv[iq], q1 = (u32)(Pv[iq] >> 32);
                            lt[w] = ripple(lt[w], a0, q0);
                            lt[w] = ripple(lt[w], a1, q1);
                            // (pinned here: hipcc otherwise sinks the ripples of windows 1.. under the lane-dependent
                            // `w < nwin` of the masks below -- behind the loads of all K + WPL - 1 planes, 68 registers)
                            asm volatile("" : "+v"(lt[w]));
                        }
                    }
                }
                u32 m[WPL];
#pragma unroll
                for (int w = 0; w < WPL; ++w) m[w] = ((u32)w < nwin) ? lt[w] : 0u;
                if ((vlo & vhi) != ~0u) {     // (wave-uniform: a tile with blanked reads)
                    const u32 vm = set ? vhi : vlo;
#pragma unroll
                    for (int w = 0; w < WPL; ++w) m[w] &= vm;
                }
#pragma unroll
                for (int w = 0; w < WPL; ++w) pc_acc(mcnt, m[w]);
                asm volatile("" ::: "memory");
                // ---- D, pass 2: R planes per (v_and run, v_bcnt run at raised priority) pair
                constexpr int NPL = K + WPL - 1;
                constexpr int R = (WPL <= 4) ? KMX_BS_RUNR : 1;
                u64 vcur[R];
#pragma unroll
                for (int h = 0; h < R; ++h) vcur[h] = KMX_PLANE(h < NPL ? h : 0);
#pragma unroll
                for (int i = 0; i < NPL; i += R) {
                    u64 vnext[R];
#pragma unroll
                    for (int h = 0; h < R; ++h) vnext[h] = KMX_PLANE(i + R + h < NPL ? i + R + h : i);
                    __builtin_amdgcn_sched_barrier(0);
                    u32 x[2 * R * WPL];
#pragma unroll
                    for (int h = 0; h < R; ++h) {
#pragma unroll
                        for (int w = 0; w < WPL; ++w) {
                            x[2 * WPL * h + 2 * w] = m[w] & (u32)vcur[h];
                            x[2 * WPL * h + 2 * w + 1] = m[w] & (u32)(vcur[h] >> 32);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(3);
#pragma unroll
                    for (int h = 0; h < R; ++h) {
                        if (i + h >= NPL) continue;
#pragma unroll
                        for (int w = 0; w < WPL; ++w) {
                            const int t = i + h - w;
                            if (t < 0 || t > K - 1) continue;
                            const int tc = t < K - 1 - t ? t : K - 1 - t;
                            pc_acc(D[2 * tc], x[2 * WPL * h + 2 * w]);
                            pc_acc(D[2 * tc + 1], x[2 * WPL * h + 2 * w + 1]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(0);
#pragma unroll
                    for (int h = 0; h < R; ++h) vcur[h] = vnext[h];
                }
            }
            // ---- the buffer goes back to its producer
            pc_lds_drain();
            if (lane == 0) pc_st(c_state + b, 0u);
        }
#undef KMX_PLANE
        // ---- this consumer's part of the closed form: cnt(t,b) += C[t][b] + C[K-1-t][b] - sum popcount(m)
        if (n_tiles != 0u) {
            u32 mcnt_c;
            asm volatile("v_mov_b32 %0, %1" : "=&v"(mcnt_c) : "v"(mcnt));
            const u64 mc = wave_sum((u64)mcnt_c);
            u64* const CS = reinterpret_cast<u64*>(lds + lay.cons0 + cidx * Lay::CSZ);
#pragma unroll
            for (int q = 0; q < 2 * NT; ++q) {
                u32 dq;
                asm volatile("v_mov_b32 %0, %1" : "=&v"(dq) : "v"(D[q]));
                const u64 v = wave_sum((u64)dq);
                if (lane == 0) CS[q] = v;
            }
            lds_fence();
            u64 s0 = 0, x0 = 0;
            for (u32 pid = lane; pid < 2u * (u32)K; pid += 64u) {
                const u32 t = pid >> 1, bb = pid & 1u, t2 = (u32)K - 1u - t;
                const u32 tc = t < t2 ? t : t2;
                u64 cc = CS[2u * tc + bb];
                if (t == t2) cc += cc;
                const u64 cnt = cc - mc;
                s0 += cnt << (2u * t + bb);
                if (want_hash && (cnt & 1ull)) x0 ^= 1ull << (2u * t2 + bb);
            }
            emit_sums(0, wave_sum(s0), wave_xor(x0), 0);
        }
    }
}

// ------------------------------------------------------------------ launcher
template <int K, int NW, int WPL, int NP, int NC, int WPS, int BPP, int LATE = KMX_PC_LATE>
static hipError_t launch_bs_pc(const uint8_t* bases, u64 n_reads, u32 L, u32 want_hash, u32 want_sumfw, kmx_summary* out,
                               unsigned long long* queue, int n_cu, hipStream_t stream) {
    auto kern = scan_bitsliced_pc_kernel<K, NW, WPL, NP, NC, WPS, BPP, LATE>;
    u32 lead = (u32)(reinterpret_cast<uintptr_t>(bases) & 15u);
    bases -= lead;
    if (lead != 0u && 4u * L + 1u > 64u * (u32)NW) return hipErrorInvalidValue;
    const u32 chunks = 4u * L + (lead != 0u ? 1u : 0u);
    const PcLayout<K, NW, NP, NC, BPP> lay(chunks);
    const size_t lds_bytes = (size_t)lay.total * 4u;
    static thread_local int bpc = 0, bpc_dev = -1;
    static thread_local size_t bpc_lds = 0;
    int dev_now = -1;
    (void)hipGetDevice(&dev_now);
    if (bpc == 0 || bpc_lds != lds_bytes || bpc_dev != dev_now) {
        if (lds_bytes > 64u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        int b = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kern, 64 * (NP + NC), lds_bytes);
        if (e != hipSuccess) return e;
        bpc = b > 0 ? b : 1;
        bpc_lds = lds_bytes;
        bpc_dev = dev_now;
        if (getenv("KMX_BS_PRINT_BPC"))
            fprintf(stderr, "kmx: producer/consumer K=%d NW=%d WPL=%d %dP+%dC: %d blocks per CU, %zu B of LDS each\n", K, NW, WPL, NP, NC, bpc, lds_bytes);
    }
    if (L < (u32)K || 2u * ((L - (u32)K + 1u + (u32)WPL - 1u) / (u32)WPL) > 64u) return hipErrorInvalidValue;
    if ((n_reads >> 6) >= (1ull << 36)) return hipErrorInvalidValue;
    const u64 n_tiles = (n_reads + 63u) >> 6;
    u64 grid = (u64)n_cu * (u64)bpc;
    const u64 need = (n_tiles + (u64)NP - 1u) / (u64)NP;
    if (grid > need) grid = need;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * (NP + NC)), lds_bytes, stream, bases, n_reads, L, want_hash, want_sumfw, out, queue, lead);
    // the reads the producers blanked out (none on clean input: the waves return at once)
    u64 grid1 = (u64)n_cu * 4u;
    const u64 need1 = ((n_reads >> 6) + 255u) / 256u;
    if (grid1 > need1) grid1 = need1;
    hipLaunchKernelGGL((roll_flagged_kernel<K, false>), dim3((unsigned)(grid1 ? grid1 : 1)), dim3(256), 0, stream, bases, n_reads, L, want_hash,
                       want_sumfw, out, queue, nullptr, lead);
    return hipGetLastError();
}

}  // namespace kmx
