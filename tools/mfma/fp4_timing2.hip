// dev probe 2: does an FP4 MFMA overlap with other waves' VALU work on the same SIMD?  shader cycles (s_memtime) + wall time
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32;
typedef unsigned long long u64;

// MODE bit0: VALU = 1 full-rate v_and only, 2 half-rate v_bcnt only, 3 mix (runs of 16 with s_setprio)
// NM: MFMAs per 512 VALU; SHAPE 0: 32x32x64, 1: 16x16x128
// ROLE: 0 all waves do both; 1 = blocks with (blockIdx & 3) == 0 do MFMA only, the others VALU only
template <int VMODE, int NM, int SHAPE, int ROLE>
__global__ void __launch_bounds__(256, 2) body(u32* out, u64* cyc, int iters) {
    u32 x[8], d[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 2654435761u + i; d[i] = 0; }
    v8i a = {1, 2, 3, 4, 0, 0, 0, 0}, b = {5, 6, 7, 8, 0, 0, 0, 0};
    a[0] = (int)(threadIdx.x & 0x22222222); b[1] = (int)(threadIdx.x & 0x11111111);
    v16f c[4];
    v4f c4[4];
    for (int q = 0; q < 4; ++q) { for (int j = 0; j < 16; ++j) c[q][j] = 0.f; for (int j = 0; j < 4; ++j) c4[q][j] = 0.f; }
    const bool do_valu = ROLE == 0 || (blockIdx.x & 3) != 0;
    const bool do_mfma = ROLE == 0 || (blockIdx.x & 3) == 0;
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (VMODE != 0 && do_valu) {
                if constexpr (VMODE == 1) {
#pragma unroll
                    for (int r = 0; r < 32; ++r) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[r & 7]) : "v"(x[(r + 3) & 7]));
                } else if constexpr (VMODE == 2) {
#pragma unroll
                    for (int r = 0; r < 32; ++r) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[r & 7]) : "v"(x[r & 7]));
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[r & 7]) : "v"(x[(r + 3) & 7]));
                    __builtin_amdgcn_s_setprio(3);
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[r & 7]) : "v"(x[r & 7]));
                    __builtin_amdgcn_s_setprio(0);
                }
            }
            if constexpr (NM > 0) {
                if (do_mfma && s < NM) {
                    if constexpr (SHAPE == 0)
                        c[s & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[s & 3], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                    else
                        c4[s & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c4[s & 3], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            }
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    u32 acc = 0;
    for (int i = 0; i < 8; ++i) acc += d[i] + x[i];
    float f = 0;
    for (int q = 0; q < 4; ++q) { for (int j = 0; j < 16; ++j) f += c[q][j]; for (int j = 0; j < 4; ++j) f += c4[q][j]; }
    out[blockIdx.x * 256 + threadIdx.x] = acc + (u32)f;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int VMODE, int NM, int SHAPE, int ROLE>
static void run(const char* name, u32* out, u64* cyc, int bpc) {
    const int blocks = 256 * bpc;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 300;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((body<VMODE, NM, SHAPE, ROLE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((body<VMODE, NM, SHAPE, ROLE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<u64> h(blocks * 4);
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / iters, mx = (double)h.back() / iters;
    printf("%-44s w/SIMD %d: %.3f ms/launch  cycles per iteration of a wave: median %.0f max %.0f  -> per SIMD-iteration %.0f\n", name, bpc, ms / 5, med, mx, mx / bpc);
}
int main() {
    u32* out; u64* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 4 * 8);
    for (int bpc : {1, 2, 4}) {
        run<1, 0, 0, 0>("512 v_and", out, cyc, bpc);
        run<1, 16, 0, 0>("512 v_and + 16 mfma32x32x64", out, cyc, bpc);
        run<2, 0, 0, 0>("512 v_bcnt", out, cyc, bpc);
        run<2, 16, 0, 0>("512 v_bcnt + 16 mfma32x32x64", out, cyc, bpc);
        run<3, 0, 0, 0>("256 and + 256 bcnt (prio)", out, cyc, bpc);
        run<3, 16, 0, 0>("256 and + 256 bcnt (prio) + 16 mfma32", out, cyc, bpc);
        run<3, 16, 1, 0>("256 and + 256 bcnt (prio) + 16 mfma16x16x128", out, cyc, bpc);
        run<0, 16, 0, 0>("16 mfma32x32x64 only", out, cyc, bpc);
        run<0, 16, 1, 0>("16 mfma16x16x128 only", out, cyc, bpc);
    }
    // roles: 4 waves per SIMD, one of them MFMA-only (16 per iteration), three VALU-only (512 per iteration)
    run<3, 16, 0, 1>("roles: 3 x (mix 512) | 1 x 16 mfma32", out, cyc, 4);
    run<3, 0, 0, 1>("roles: 3 x (mix 512) | 1 x idle", out, cyc, 4);
    run<2, 16, 0, 1>("roles: 3 x (bcnt 512) | 1 x 16 mfma32", out, cyc, 4);
    run<2, 0, 0, 1>("roles: 3 x (bcnt 512) | 1 x idle", out, cyc, 4);
    run<1, 16, 0, 1>("roles: 3 x (and 512) | 1 x 16 mfma32", out, cyc, 4);
    run<1, 0, 0, 1>("roles: 3 x (and 512) | 1 x idle", out, cyc, 4);
    return 0;
}
