#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u32;

// A: mbits[lane] = 32-bit word: row (lane&31), K-half (lane>>5), bit rho = k-slot rho
// B: pbits[lane] = col (lane&31), K-half (lane>>5)
__device__ __forceinline__ v8i expandB(u32 P) {
    v8i r = {0,0,0,0,0,0,0,0};
    r[0] = P & 0x44444444u;          // 2.0 * P[4n+2]
    r[1] = P & 0x22222222u;          // 1.0 * P[4n+1]
    r[2] = P & 0x11111111u;          // 0.5 * P[4n]
    r[3] = (P >> 1) & 0x44444444u;   // 2.0 * P[4n+3]
    return r;
}
__device__ __forceinline__ v8i expandA(u32 M) {
    v8i r = {0,0,0,0,0,0,0,0};
    r[0] = __builtin_amdgcn_alignbit(M, M, 2) & 0x11111111u;   // 0.5 * M[4n+2]
    r[1] = M & 0x22222222u;                                     // 1.0 * M[4n+1]
    r[2] = __builtin_amdgcn_alignbit(M, M, 30) & 0x44444444u;  // 2.0 * M[4n]
    r[3] = __builtin_amdgcn_alignbit(M, M, 3) & 0x11111111u;   // 0.5 * M[4n+3]
    return r;
}
__global__ void probe(const u32* mb, const u32* pb, float* out, int iters) {
    const u32 lane = threadIdx.x;
    v8i a = expandA(mb[lane]), b = expandB(pb[lane]);
    v16f c;
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
    for (int it = 0; it < iters; ++it)
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int j = 0; j < 16; ++j) out[j * 64 + lane] = c[j];
}
int main() {
    std::vector<u32> mb(64), pb(64);
    srand(1);
    int fails = 0;
    for (int trial = 0; trial < 6; ++trial) {
        for (int i = 0; i < 64; ++i) {
            mb[i] = trial == 5 ? 0xFFFFFFFFu : ((u32)rand() << 16) ^ (u32)rand();
            pb[i] = trial == 5 ? 0xFFFFFFFFu : ((u32)rand() << 16) ^ (u32)rand();
            if (trial == 1) { mb[i] = 1u << (i & 31); }            // single bits
            if (trial == 2) { pb[i] = 0x80000000u >> (i & 31); }
        }
        u32 *dm, *dp; float* dout;
        hipMalloc(&dm, 256); hipMalloc(&dp, 256); hipMalloc(&dout, 16 * 64 * 4);
        hipMemcpy(dm, mb.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dp, pb.data(), 256, hipMemcpyHostToDevice);
        int iters = trial == 5 ? (1 << 18) : (trial == 4 ? 1000 : 1);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dm, dp, dout, iters);
        std::vector<float> out(16 * 64);
        hipMemcpy(out.data(), dout, 16 * 64 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int j = 0; j < 16; ++j) for (int l = 0; l < 64; ++l) {
            int row = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5), col = l & 31;
            long long exp = 0;
            for (int h = 0; h < 2; ++h) exp += __builtin_popcount(mb[row + 32 * h] & pb[col + 32 * h]);
            exp *= iters;
            if ((double)out[j * 64 + l] != (double)exp) { if (bad < 5) printf("trial %d j %d lane %d: got %.3f want %lld\n", trial, j, l, out[j*64+l], exp); ++bad; }
        }
        printf("trial %d iters %d: %d mismatches (sample out[0]=%.1f)\n", trial, iters, bad, out[0]);
        fails += bad != 0;
    }
    printf(fails ? "PROBE FAIL\n" : "PROBE OK\n");
    return fails;
}
