// ubench10.hip -- (development tool) what does a MIXED full-rate / half-rate VALU stream cost on gfx950, in shader cycles?
// One block per CU, W waves per SIMD, every wave runs the same 64-instruction body ITER times; reports cycles per body per
// SIMD (s_memtime, max over the block's waves) next to the wall-clock figure.  Bodies: F = v_and, H = v_bcnt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 2000;
#define F1 "v_and_b32 %0, %4, %5\n"
#define F2 "v_and_b32 %1, %5, %6\n"
#define F3 "v_xor_b32 %2, %4, %6\n"
#define F4 "v_xor_b32 %3, %5, %7\n"
#define H1 "v_bcnt_u32_b32 %0, %4, %0\n"
#define H2 "v_bcnt_u32_b32 %1, %5, %1\n"
#define H3 "v_bcnt_u32_b32 %2, %6, %2\n"
#define H4 "v_bcnt_u32_b32 %3, %7, %3\n"
#define FD "v_and_b32 %0, %4, %0\n"
#define HD "v_bcnt_u32_b32 %0, %4, %0\n"
#define P3 "s_setprio 3\n"
#define P0 "s_setprio 0\n"
#define X2(x) x x
#define X4(x) X2(x) X2(x)
#define X8(x) X4(x) X4(x)
#define X16(x) X8(x) X8(x)
#define ASMV(BODY) asm volatile(BODY : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(a), "+v"(b), "+v"(c), "+v"(e))

template <int BODY>
__global__ void __launch_bounds__(1024) k_body(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
    uint32_t d0 = seed, d1 = seed * 3, d2 = seed * 5, d3 = seed * 7, a = threadIdx.x * seed, b = a ^ 0x55, c = a + 77, e = a * 9;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
        if (BODY == 0) ASMV(X16(F1 F2 F3 F4));                               // 64 F, 4 chains
        if (BODY == 1) ASMV(X16(H1 H2 H3 H4));                               // 64 H, 4 chains
        if (BODY == 2) ASMV(X16(F1 H2 F3 H4));                               // F H F H
        if (BODY == 3) ASMV(X4(F1 F2 F3 F4 F1 F2 F3 F4 H1 H2 H3 H4 H1 H2 H3 H4));   // runs of 8
        if (BODY == 4) ASMV(X4(P0 F1 F2 F3 F4 F1 F2 F3 F4 P3 H1 H2 H3 H4 H1 H2 H3 H4));   // runs of 8, priority raised for H
        if (BODY == 5) ASMV(X16(F1 F2) X16(H1 H2));                          // runs of 32
        if (BODY == 6) ASMV(P0 X16(F1 F2) P3 X16(H1 H2));                    // runs of 32 + priority
        if (BODY == 7) ASMV(X16(FD FD FD FD));                               // 64 F, one dependent chain
        if (BODY == 8) ASMV(X16(HD HD HD HD));                               // 64 H, one dependent chain
        if (BODY == 9) ASMV(X16(F1 F1 F2 F2));                               // 64 F, WAW on the same register back to back
        if (BODY == 10) ASMV(X2(P0 X8(F1 F2) P3 X8(H1 H2)));                  // runs of 16 + priority
        if (BODY == 11) ASMV(X8(P0 F1 F2 F3 F4 P3 H1 H2 H3 H4));              // runs of 4 + priority
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_setprio 0");
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 ^ d1 ^ d2 ^ d3 ^ a ^ b ^ c ^ e;
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}

template <int BODY> void run(uint32_t* out, unsigned long long* cyc, int cus, int waves_per_simd, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(cus), block(256 * waves_per_simd);
    hipLaunchKernelGGL(k_body<BODY>, grid, block, 0, 0, out, cyc, 1u);
    hipDeviceSynchronize();
    hipMemset(cyc, 0, 8 * cus);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_body<BODY>, grid, block, 0, 0, out, cyc, 2u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, cyc, 8 * cus, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < cus; ++i) s += (double)h[i];
    double ticks = s / cus / ITER;
    printf("%-44s W=%d  %7.1f ns  %7.1f ticks per body  -> %5.2f ns per instruction per SIMD\n", name, waves_per_simd, ms * 1e6 / ITER, ticks,
           ms * 1e6 / ITER / (64.0 * waves_per_simd));
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    printf("clock %d kHz, wall clock rate %d kHz\n", prop.clockRate, prop.memoryClockRate);
    uint32_t* out; hipMalloc(&out, (size_t)cus * 1024 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 8 * cus);
    for (int w = 1; w <= 4; w *= 2) {
#define ROW(B, NAME) run<B>(out, cyc, cus, w, NAME);
        ROW(0, "64 F (4 chains)") ROW(1, "64 H (4 chains)") ROW(7, "64 F one dependent chain") ROW(8, "64 H one dependent chain")
        ROW(9, "64 F, pairs writing the same register")
        ROW(2, "F H F H") ROW(3, "runs of 8") ROW(4, "runs of 8, prio 3 on H") ROW(11, "runs of 4, prio 3 on H") ROW(10, "runs of 16, prio 3 on H")
        ROW(5, "runs of 32") ROW(6, "runs of 32, prio 3 on H")
    }
    run<0>(out, cyc, cus, 3, "64 F (4 chains)"); run<1>(out, cyc, cus, 3, "64 H (4 chains)"); run<4>(out, cyc, cus, 3, "runs of 8, prio 3 on H");
    run<6>(out, cyc, cus, 3, "runs of 32, prio 3 on H");
    return 0;
}
