// dev probe: what do 16 FP4 MFMAs (32x32x64) per "tile" cost next to a VALU stream like the scan's?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u32;

template <int NV, int NM>
__global__ void __launch_bounds__(256, 3) body(u32* out, int iters) {
    u32 x[8], d[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 2654435761u + i; d[i] = 0; }
    v8i a = {1, 2, 3, 4, 0, 0, 0, 0}, b = {5, 6, 7, 8, 0, 0, 0, 0};
    a[0] = (int)(threadIdx.x & 0x22222222); b[1] = (int)(threadIdx.x & 0x11111111);
    v16f c[4];
    for (int q = 0; q < 4; ++q) for (int j = 0; j < 16; ++j) c[q][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // NV/16 VALU instructions per step: half v_and (full rate), half v_bcnt (half rate), in runs of 8 with s_setprio
            if constexpr (NV > 0) {
                constexpr int R = NV / 32;
#pragma unroll
                for (int r = 0; r < R; ++r) { x[r & 7] = x[(r + 1) & 7] & (x[(r + 3) & 7] | 0x01010101u); }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(3);
#pragma unroll
                for (int r = 0; r < R; ++r) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[r & 7]) : "v"(x[r & 7]));
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(0);
            }
            if constexpr (NM > 0) {
                if (s < NM) {
                    a[1] ^= (int)x[0];
                    c[s & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[s & 3], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            }
        }
    }
    u32 acc = 0;
    for (int i = 0; i < 8; ++i) acc += d[i] + x[i];
    float f = 0;
    for (int q = 0; q < 4; ++q) for (int j = 0; j < 16; ++j) f += c[q][j];
    out[blockIdx.x * 256 + threadIdx.x] = acc + (u32)f;
}
template <int NV, int NM>
static void run(const char* name, u32* out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((body<NV, NM>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((body<NV, NM>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: (blocks / 256 CUs) waves, each iters "tiles"
    const double tiles_per_simd = (double)blocks / 256.0 * iters;
    printf("%-28s blocks/CU %d: %.3f ms per launch, %.1f ns per tile per SIMD\n", name, blocks / 256, ms / 5, ms / 5 * 1e6 / tiles_per_simd);
}
int main() {
    u32* out; hipMalloc(&out, 4096 * 256 * 4);
    for (int bpc = 1; bpc <= 4; ++bpc) {
        if (bpc == 3) continue;
        const int blocks = 256 * bpc;
        run<512, 0>("512 VALU", out, blocks);
        run<512, 16>("512 VALU + 16 MFMA", out, blocks);
        run<0, 16>("16 MFMA", out, blocks);
        run<1024, 0>("1024 VALU", out, blocks);
        run<1024, 16>("1024 VALU + 16 MFMA", out, blocks);
    }
    return 0;
}
