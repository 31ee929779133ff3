// ubench4.hip -- run-length test (development tool): R full-rate ops followed by R half-rate ops, per wave.
// Question: how long must a run of full-rate VALU instructions be before gfx950 issues it at 2 cycles/wave64?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
constexpr int ITER = 2000;
#define A1 "v_and_b32 %8, %10, %11\n"
#define A2 "v_and_b32 %9, %10, %12\n"
#define X1 "v_bitop3_b32 %8, %10, %11, %12 bitop3:0x96\n"
#define B1 "v_bcnt_u32_b32 %0, %10, %0\n"
#define B2 "v_bcnt_u32_b32 %1, %11, %1\n"
#define R2(x,y) x y
#define R4(x,y) R2(x,y) R2(x,y)
#define R8(x,y) R4(x,y) R4(x,y)
#define R16(x,y) R8(x,y) R8(x,y)
#define R32(x,y) R16(x,y) R16(x,y)
// each body has 32 full-rate and 32 half-rate instructions
#define BODY_R1  R32(A1 B1, A2 B2)
#define BODY_R2  R16(A1 A2 B1 B2, A1 A2 B1 B2)
#define BODY_R4  R8(R2(A1,A2) R2(A1,A2) R2(B1,B2) R2(B1,B2), R2(A1,A2) R2(A1,A2) R2(B1,B2) R2(B1,B2))
#define BODY_R8  R4(R4(A1,A2) R4(A1,A2) R4(B1,B2) R4(B1,B2), R4(A1,A2) R4(A1,A2) R4(B1,B2) R4(B1,B2))
#define BODY_R32 R32(A1,A2) R32(B1,B2)
#define BODY_AX  R32(A1 X1, A2 X1)   /* two different full-rate opcodes alternating: 64 full-rate */
#define BODY_A   R32(A1 A2, A1 A2)
#define BODY_B   R32(B1 B2, B1 B2)
#define KERNEL(NAME, BODY)                                                                           \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                          \
    uint32_t d0=seed,d1=seed+1,d2=seed+2,d3=seed+3,d4=seed+4,d5=seed+5,d6=seed+6,d7=seed+7, t0=0, t1=0; \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55aa55aau, e = b + 7u;               \
    for (int i = 0; i < ITER; ++i) {                                                                 \
        asm volatile(BODY                                                                            \
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), \
                       "+v"(t0), "+v"(t1), "+v"(a), "+v"(b), "+v"(c), "+v"(e));                       \
    }                                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0^d1^d2^d3^d4^d5^d6^d7^t0^t1^a^b^c^e;              \
}
KERNEL(k_r1, BODY_R1) KERNEL(k_r2, BODY_R2) KERNEL(k_r4, BODY_R4) KERNEL(k_r8, BODY_R8) KERNEL(k_r32, BODY_R32)
KERNEL(k_ax, BODY_AX) KERNEL(k_a, BODY_A) KERNEL(k_b, BODY_B)
struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t); };
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    Entry es[] = {{"32 and | 32 bcnt: run 1", k_r1}, {"run 2", k_r2}, {"run 4", k_r4}, {"run 8", k_r8}, {"run 32", k_r32},
                  {"and/bitop3 alternating", k_ax}, {"and only", k_a}, {"bcnt only", k_b}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-26s ns per 64-instruction body per SIMD-wave (ideal mixed = 32*0.9 + 32*1.75 = 85; all-half = 112)\n", "pattern");
    for (auto& e : es) {
        printf("%-26s", e.name);
        for (int wps : {1, 2, 3, 4, 6}) {
            dim3 grid(cus * wps), block(256);
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 1u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 2u);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("  w%d %7.1f", wps, ms * 1e6 / ITER / wps);
        }
        printf("\n");
    }
    return 0;
}
