// ubench7.hip -- (development tool) self-aligning phases with s_setprio:
// each wave runs R full-rate v_and, then raises its priority, runs R half-rate v_bcnt, lowers it again.
// If gfx950 co-issues full-rate instructions of two waves only when both are at one, this should pull the
// time per (32 and + 32 bcnt) from ~108 ns (everything at 4 cycles) towards 85 ns (and at 2 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
constexpr int ITER = 2000;
#define A1 "v_and_b32 %8, %10, %11\n"
#define A2 "v_and_b32 %9, %10, %12\n"
#define B1 "v_bcnt_u32_b32 %0, %10, %0\n"
#define B2 "v_bcnt_u32_b32 %1, %11, %1\n"
#define HI "s_setprio 3\n"
#define LO "s_setprio 0\n"
#define R2(x,y) x y
#define R4(x,y) R2(x,y) R2(x,y)
#define R8(x,y) R4(x,y) R4(x,y)
#define R16(x,y) R8(x,y) R8(x,y)
#define R32(x,y) R16(x,y) R16(x,y)
#define P4  R2(A1,A2) R2(A1,A2) HI R2(B1,B2) R2(B1,B2) LO
#define P8  R4(A1,A2) R4(A1,A2) HI R4(B1,B2) R4(B1,B2) LO
#define P16 R8(A1,A2) R8(A1,A2) HI R8(B1,B2) R8(B1,B2) LO
#define P32 R16(A1,A2) R16(A1,A2) HI R16(B1,B2) R16(B1,B2) LO
#define N4  R2(A1,A2) R2(A1,A2) R2(B1,B2) R2(B1,B2)
#define N32 R16(A1,A2) R16(A1,A2) R16(B1,B2) R16(B1,B2)
#define ASMV(BODY) asm volatile(BODY : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(t0), "+v"(t1), "+v"(a), "+v"(b), "+v"(c), "+v"(e))
#define KERNEL(NAME, BODY)                                                                           \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                          \
    uint32_t d0=seed,d1=seed+1,d2=seed+2,d3=seed+3,d4=seed+4,d5=seed+5,d6=seed+6,d7=seed+7, t0=0, t1=0; \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55aa55aau, e = b + 7u;               \
    for (int i = 0; i < ITER; ++i) ASMV(BODY);                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0^d1^d2^d3^d4^d5^d6^d7^t0^t1^a^b^c^e;              \
}
KERNEL(k_p4, R8(P4, P4)) KERNEL(k_p8, R4(P8, P8)) KERNEL(k_p16, R2(P16, P16)) KERNEL(k_p32, P32) KERNEL(k_n32, N32)
#define I32 HI R16(A1,A2) R16(A1,A2) LO R16(B1,B2) R16(B1,B2)
#define P2  A1 A2 HI B1 B2 LO
#define P1  A1 HI B1 LO
KERNEL(k_i32, I32) KERNEL(k_p2, R16(P2, P2)) KERNEL(k_p1, R32(P1, P1))
#define KERNEL_STATIC(NAME, BODY)                                                                    \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                          \
    uint32_t d0=seed,d1=seed+1,d2=seed+2,d3=seed+3,d4=seed+4,d5=seed+5,d6=seed+6,d7=seed+7, t0=0, t1=0; \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55aa55aau, e = b + 7u;               \
    if (blockIdx.x & 1) asm volatile("s_setprio 3"); else asm volatile("s_setprio 0");               \
    for (int i = 0; i < ITER; ++i) ASMV(BODY);                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0^d1^d2^d3^d4^d5^d6^d7^t0^t1^a^b^c^e;              \
}
KERNEL_STATIC(k_s1, R32(A1 B1, A2 B2)) KERNEL_STATIC(k_s32, N32)
KERNEL(k_p128, R4(R16(A1 A2, A1 A2), R16(A1 A2, A1 A2)) HI R4(R16(B1 B2, B1 B2), R16(B1 B2, B1 B2)) LO)
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 16 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char* name; void (*fn)(uint32_t*, uint32_t); double scale; } es[] = {
        {"no setprio, runs of 32", k_n32, 1}, {"setprio, runs of 4", k_p4, 1}, {"setprio, runs of 8", k_p8, 1},
        {"setprio, runs of 16", k_p16, 1}, {"setprio, runs of 32", k_p32, 1}, {"setprio, runs of 128", k_p128, 0.25},
        {"inverse (hi in and), 32", k_i32, 1}, {"setprio, runs of 2", k_p2, 1}, {"setprio, runs of 1", k_p1, 1},
        {"static prio by block, run 1", k_s1, 1}, {"static prio by block, run 32", k_s32, 1}};
    printf("ns per (32 v_and + 32 v_bcnt) per wave per SIMD; 85 = and at 2 cycles, 112 = everything at 4 cycles\n");
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
            printf("  w%d %7.1f", wps, ms * 1e6 / ITER / wps * e.scale);
        }
        printf("\n");
    }
    return 0;
}
