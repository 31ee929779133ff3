// dev probe (round 4): what does the HBM read stream of the scan's shape (9600-byte tiles, 16 B per lane, one tile per wave at a
// time, the next one requested while this one is consumed) deliver as a function of waves per SIMD and of how tiles are handed out?
//   static  : wave w takes tiles w, w + n_waves, ...                     (the calibration kernel of bench.py)
//   queue32 : 32 ticket heads, head q owns the tiles == q (mod 32)      (the scan kernels)
//   queue1k : tickets taken in blocks of 1024 consecutive tiles per head (a narrower window per head)
// busy: N dependent VALU instructions per tile between the loads' arrival and the next request (0 = pure stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32; typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int BUSY, int NT>
__global__ void __launch_bounds__(256) stream_kernel(const uint8_t* __restrict__ buf, u64 n_tiles, u64* __restrict__ queue, u64* __restrict__ out) {
    constexpr int TILE16 = 600, IT = 10;
    const u32x4* __restrict__ p = reinterpret_cast<const u32x4*>(buf);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = (u64)gridDim.x * 4u;
    u32 acc = 0;
    u32x4 w[IT];
    auto issue = [&](u64 t) {
        const u32x4* tb = p + t * TILE16;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            u32 c = it * 64u + lane;
            c = c < TILE16 ? c : TILE16 - 1;
            if (NT) w[it] = __builtin_nontemporal_load(tb + c); else w[it] = tb[c];
        }
    };
    u32 qid = (blockIdx.x & 255u) >> 3;
    u64 pos = 0, run_left = 0;
    auto next = [&](u64 cur) -> u64 {
        if (MODE == 0) return cur + n_waves;
        if (MODE == 1) {
            u64 v = 0;
            if (lane == 0) v = atomicAdd(queue + qid * 16u, 1ull);
            v = __shfl(v, 0);
            return v * 32u + qid;
        }
        // MODE 2: runs of 4 consecutive tiles from ONE of 8 heads (head q owns the blocks of 4 tiles == q mod 8)
        if (run_left == 0) {
            u64 v = 0;
            if (lane == 0) v = atomicAdd(queue + (qid & 7u) * 16u, 1ull);
            v = __shfl(v, 0);
            pos = (v * 8u + (qid & 7u)) * 4u;
            run_left = 4;
        }
        run_left -= 1;
        return pos++;
    };
    u64 t = MODE == 0 ? wave : next(0);
    if (t < n_tiles) issue(t);
    while (t < n_tiles) {
        u32 a = 0;
#pragma unroll
        for (int it = 0; it < IT; ++it) a ^= w[it].x ^ w[it].y ^ w[it].z ^ w[it].w;
        const u64 tn = next(t);
        issue(tn < n_tiles ? tn : t);
#pragma unroll 8
        for (int i = 0; i < BUSY; ++i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(acc));
        acc ^= a;
        t = tn;
    }
    if (acc == 0x12345u) out[0] = acc;
}
template <int MODE, int BUSY, int NT>
static void run(const char* name, const uint8_t* buf, u64 n_tiles, u64* queue, u64* out, int bpc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0; int n = 0;
    for (int rep = 0; rep < 12; ++rep) {
        hipMemsetAsync(queue, 0, 8192, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<MODE, BUSY, NT>), dim3(256 * bpc), dim3(256), 0, 0, buf, n_tiles, queue, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 4) { best = ms < best ? ms : best; sum += ms; ++n; }
    }
    printf("%-10s busy %4d nt %d  waves/SIMD %d: avg %.3f ms min %.3f ms -> %.0f GB/s\n", name, BUSY, NT, bpc, sum / n, best, n_tiles * 9600.0 / (sum / n) / 1e6);
}
int main() {
    const u64 n_tiles = 1562500;   // 15 GB
    uint8_t* buf; u64 *queue, *out;
    hipMalloc(&buf, n_tiles * 9600 + 4096); hipMemset(buf, 1, n_tiles * 9600); hipMalloc(&queue, 8192); hipMalloc(&out, 64);
    for (int bpc : {2, 3, 4, 6, 8}) {
        run<0, 0, 1>("static", buf, n_tiles, queue, out, bpc);
        run<1, 0, 1>("queue32", buf, n_tiles, queue, out, bpc);
        run<2, 0, 1>("runs4x8", buf, n_tiles, queue, out, bpc);
    }
    for (int bpc : {2, 3, 4}) {
        run<1, 0, 0>("queue32", buf, n_tiles, queue, out, bpc);
        run<1, 300, 1>("queue32", buf, n_tiles, queue, out, bpc);
        run<1, 600, 1>("queue32", buf, n_tiles, queue, out, bpc);
        run<0, 600, 1>("static", buf, n_tiles, queue, out, bpc);
    }
    return 0;
}
