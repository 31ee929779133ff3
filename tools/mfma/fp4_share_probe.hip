// dev probe: two matrix products share ONE accumulator block -- rows 0..15 of the first and rows 16..31 of the second count, the
// other rows are switched off through the A operand's per-lane E8M0 scale (0 = 2^-127: their products vanish next to integers)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u32;
__device__ __forceinline__ v8i expandB(u32 P) { v8i r = {0,0,0,0,0,0,0,0}; r[0] = P & 0x44444444u; r[1] = P & 0x22222222u; r[2] = P & 0x11111111u; r[3] = (P >> 1) & 0x44444444u; return r; }
__device__ __forceinline__ v8i expandA(u32 M) {
    v8i r = {0,0,0,0,0,0,0,0};
    r[0] = __builtin_amdgcn_alignbit(M, M, 2) & 0x11111111u; r[1] = M & 0x22222222u;
    r[2] = __builtin_amdgcn_alignbit(M, M, 30) & 0x44444444u; r[3] = __builtin_amdgcn_alignbit(M, M, 3) & 0x11111111u; return r;
}
__global__ void probe(const u32* m1, const u32* p1, const u32* m2, const u32* p2, float* out, int iters) {
    const u32 lane = threadIdx.x;
    const v8i a1 = expandA(m1[lane]), b1 = expandB(p1[lane]), a2 = expandA(m2[lane]), b2 = expandB(p2[lane]);
    const int lo = (lane & 31u) < 16u ? 0x7F7F7F7F : 0, hi = (lane & 31u) < 16u ? 0 : 0x7F7F7F7F, unit = 0x7F7F7F7F;
    v16f c;
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
    for (int it = 0; it < iters; ++it) {
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b1, c, 4, 4, 0, lo, 0, unit);
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a2, b2, c, 4, 4, 0, hi, 0, unit);
    }
    for (int j = 0; j < 16; ++j) out[j * 64 + lane] = c[j];
}
int main() {
    std::vector<u32> m1(64), p1(64), m2(64), p2(64);
    srand(7);
    int fails = 0;
    for (int trial = 0; trial < 5; ++trial) {
        for (int i = 0; i < 64; ++i) {
            auto r = [&]() { return trial == 4 ? 0xFFFFFFFFu : (((u32)rand() << 16) ^ (u32)rand()) & (trial == 3 ? 0x00010001u : 0xFFFFFFFFu); };
            m1[i] = r(); p1[i] = r(); m2[i] = r(); p2[i] = r();
        }
        u32 *d[4]; float* dout;
        std::vector<u32>* h[4] = {&m1, &p1, &m2, &p2};
        for (int x = 0; x < 4; ++x) { (void)hipMalloc(&d[x], 256); (void)hipMemcpy(d[x], h[x]->data(), 256, hipMemcpyHostToDevice); }
        (void)hipMalloc(&dout, 16 * 64 * 4);
        const int iters = trial == 4 ? (1 << 17) : (trial == 2 ? 3000 : 1);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d[0], d[1], d[2], d[3], dout, iters);
        std::vector<float> out(16 * 64);
        (void)hipMemcpy(out.data(), dout, 16 * 64 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int j = 0; j < 16; ++j) for (int l = 0; l < 64; ++l) {
            const int row = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5), col = l & 31;
            const std::vector<u32>&mm = row < 16 ? m1 : m2, &pp = row < 16 ? p1 : p2;
            long long exp = 0;
            for (int h2 = 0; h2 < 2; ++h2) exp += __builtin_popcount(mm[row + 32 * h2] & pp[col + 32 * h2]);
            exp *= iters;
            if ((long long)out[j * 64 + l] != exp || out[j * 64 + l] < 0) { if (bad < 5) printf("trial %d row %d col %d: got %.9g want %lld\n", trial, row, col, out[j * 64 + l], exp); ++bad; }
        }
        printf("trial %d iters %d: %d mismatches after truncation to integers (sample %.9g %.9g)\n", trial, iters, bad, out[0], out[8 * 64]);
        fails += bad != 0;
    }
    printf(fails ? "SHARE PROBE FAIL\n" : "SHARE PROBE OK\n");
    return fails;
}
