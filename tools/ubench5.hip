// ubench5.hip -- (development tool) does gfx950 co-issue full-rate VALU work of one wave with half-rate work of another?
// Pattern "spec": waves with even id run only v_and, odd waves only v_bcnt (same totals as the mixed kernels of ubench4).
// Pattern "long": 512 ands then 512 bcnts per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
constexpr int ITER = 2000;
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
#define PRE uint32_t d0=seed,d1=seed+1,d2=seed+2,d3=seed+3,d4=seed+4,d5=seed+5,d6=seed+6,d7=seed+7, t0=0, t1=0; \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55aa55aau, e = b + 7u;
#define POST out[blockIdx.x * blockDim.x + threadIdx.x] = d0^d1^d2^d3^d4^d5^d6^d7^t0^t1^a^b^c^e;
// mode 0: even blocks and-only / odd blocks bcnt-only (blocks of 256 = 1 wave per SIMD each)
__global__ void __launch_bounds__(256) k_spec(uint32_t* out, uint32_t seed) {
    PRE
    if (blockIdx.x & 1) { for (int i = 0; i < ITER; ++i) ASMV(R32(B1 B2, B1 B2)); }
    else               { for (int i = 0; i < ITER; ++i) ASMV(R32(A1 A2, A1 A2)); }
    POST
}
__global__ void __launch_bounds__(256) k_long(uint32_t* out, uint32_t seed) {
    PRE
    for (int i = 0; i < ITER / 16; ++i) {
        for (int j = 0; j < 8; ++j) ASMV(R32(A1 A2, A1 A2));
        for (int j = 0; j < 8; ++j) ASMV(R32(B1 B2, B1 B2));
    }
    POST
}
// 3 ands per bcnt, 48 + 16
__global__ void __launch_bounds__(256) k_31(uint32_t* out, uint32_t seed) {
    PRE
    for (int i = 0; i < ITER; ++i) ASMV(R16(A1 A2 A1 B1, A1 A2 A1 B1));
    POST
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 16 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char* name; void (*fn)(uint32_t*, uint32_t); double bodies; } es[] = {
        {"spec: half the waves and-only, half bcnt-only (ns per 64 and + 64 bcnt)", k_spec, 2.0},
        {"long: 512 and then 512 bcnt (ns per 32+32)", k_long, 1.0},
        {"3 and : 1 bcnt (ns per 48+16; ideal 71, all-half 112)", k_31, 1.0}};
    for (auto& e : es) {
        printf("%-76s", e.name);
        for (int wps : {2, 4, 6, 8}) {
            dim3 grid(cus * wps), block(256);
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 1u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 2u);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("  w%d %7.1f", wps, ms * 1e6 / ITER / wps * e.bodies);
        }
        printf("\n");
    }
    return 0;
}
