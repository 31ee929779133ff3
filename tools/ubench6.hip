// ubench6.hip -- (development tool) per-phase wall time of alternating long runs: 2048 v_and then 2048 v_bcnt per wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
#define A1 "v_and_b32 %8, %10, %11\n"
#define A2 "v_and_b32 %9, %10, %12\n"
#define B1 "v_bcnt_u32_b32 %0, %10, %0\n"
#define B2 "v_bcnt_u32_b32 %1, %11, %1\n"
#define R2(x,y) x y
#define R4(x,y) R2(x,y) R2(x,y)
#define R8(x,y) R4(x,y) R4(x,y)
#define R16(x,y) R8(x,y) R8(x,y)
#define R32(x,y) R16(x,y) R16(x,y)
#define ASMV(BODY) asm volatile(BODY : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(t0), "+v"(t1), "+v"(a), "+v"(b), "+v"(c), "+v"(e))
__global__ void __launch_bounds__(256) k_long(uint32_t* out, uint32_t seed, uint64_t* ph, int mode) {
    uint32_t d0=seed,d1=seed+1,d2=seed+2,d3=seed+3,d4=seed+4,d5=seed+5,d6=seed+6,d7=seed+7, t0=0, t1=0;
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55aa55aau, e = b + 7u;
    uint64_t ta = 0, tb = 0;
    for (int i = 0; i < 200; ++i) {
        uint64_t w0 = wall_clock64();
        if (mode & 1) for (int j = 0; j < 32; ++j) ASMV(R32(A1 A2, A1 A2));
        uint64_t w1 = wall_clock64();
        if (mode & 2) for (int j = 0; j < 32; ++j) ASMV(R32(B1 B2, B1 B2));
        uint64_t w2 = wall_clock64();
        ta += w1 - w0; tb += w2 - w1;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { ph[0] = ta; ph[1] = tb; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0^d1^d2^d3^d4^d5^d6^d7^t0^t1^a^b^c^e;
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 16 * 256 * 4));
    uint64_t* ph; CHECK(hipMalloc(&ph, 16));
    for (int mode : {1, 2, 3}) for (int wps : {1, 2, 4}) {
        hipLaunchKernelGGL(k_long, dim3(cus * wps), dim3(256), 0, 0, out, 1u, ph, mode);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_long, dim3(cus * wps), dim3(256), 0, 0, out, 2u, ph, mode);
        CHECK(hipDeviceSynchronize());
        uint64_t h[2]; CHECK(hipMemcpy(h, ph, 16, hipMemcpyDeviceToHost));
        double n = 200.0 * 2048;
        printf("mode %d (1=and 2=bcnt 3=both) w/SIMD=%d: and phase %.3f ns/instr/wave  bcnt phase %.3f ns/instr/wave -> per SIMD-instr %.3f / %.3f\n",
               mode, wps, h[0] * 10.0 / n, h[1] * 10.0 / n, h[0] * 10.0 / n / wps, h[1] * 10.0 / n / wps);
    }
    return 0;
}
