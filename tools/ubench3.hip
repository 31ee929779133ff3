// ubench3.hip -- mixed full-rate / half-rate VALU streams on gfx950 (development tool):
// how many cycles does the (v_and -> v_bcnt accumulate) pair of the bit-sliced pass 2 really cost at
// 1..4 waves per SIMD, with one shared temporary (as hipcc emits it) vs rotating temporaries?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
constexpr int ITER = 2000;
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

#define BODY_SHARED \
  "v_and_b32 %8, %10, %11\n v_bcnt_u32_b32 %0, %8, %0\n" \
  "v_and_b32 %8, %10, %12\n v_bcnt_u32_b32 %1, %8, %1\n" \
  "v_and_b32 %8, %10, %13\n v_bcnt_u32_b32 %2, %8, %2\n" \
  "v_and_b32 %8, %11, %12\n v_bcnt_u32_b32 %3, %8, %3\n" \
  "v_and_b32 %8, %11, %13\n v_bcnt_u32_b32 %4, %8, %4\n" \
  "v_and_b32 %8, %12, %13\n v_bcnt_u32_b32 %5, %8, %5\n" \
  "v_and_b32 %8, %10, %11\n v_bcnt_u32_b32 %6, %8, %6\n" \
  "v_and_b32 %8, %12, %11\n v_bcnt_u32_b32 %7, %8, %7\n"
#define BODY_ROT \
  "v_and_b32 %8, %10, %11\n v_and_b32 %9, %10, %12\n v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n" \
  "v_and_b32 %8, %10, %13\n v_and_b32 %9, %11, %12\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %9, %3\n" \
  "v_and_b32 %8, %11, %13\n v_and_b32 %9, %12, %13\n v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %9, %5\n" \
  "v_and_b32 %8, %10, %11\n v_and_b32 %9, %12, %11\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %9, %7\n"
#define BODY_ANDONLY \
  REP8("v_and_b32 %8, %10, %11\n v_and_b32 %9, %10, %12\n")
#define BODY_BCNTONLY \
  "v_bcnt_u32_b32 %0, %10, %0\n v_bcnt_u32_b32 %1, %11, %1\n v_bcnt_u32_b32 %2, %12, %2\n v_bcnt_u32_b32 %3, %13, %3\n" \
  "v_bcnt_u32_b32 %4, %10, %4\n v_bcnt_u32_b32 %5, %11, %5\n v_bcnt_u32_b32 %6, %12, %6\n v_bcnt_u32_b32 %7, %13, %7\n" \
  "v_bcnt_u32_b32 %0, %10, %0\n v_bcnt_u32_b32 %1, %11, %1\n v_bcnt_u32_b32 %2, %12, %2\n v_bcnt_u32_b32 %3, %13, %3\n" \
  "v_bcnt_u32_b32 %4, %10, %4\n v_bcnt_u32_b32 %5, %11, %5\n v_bcnt_u32_b32 %6, %12, %6\n v_bcnt_u32_b32 %7, %13, %7\n"
// ripple-like: dependent chain of bitop3 on 4 interleaved accumulators
#define BODY_RIPPLE \
  REP4("v_bitop3_b32 %0, %0, %10, %11 bitop3:0x71\n v_bitop3_b32 %1, %1, %11, %12 bitop3:0x71\n v_bitop3_b32 %2, %2, %12, %13 bitop3:0x71\n v_bitop3_b32 %3, %3, %10, %13 bitop3:0x71\n")

#define KERNEL(NAME, BODY, WAVES)                                                                    \
__global__ void __launch_bounds__(256, WAVES) NAME(uint32_t* out, uint32_t seed, uint64_t* cyc) {    \
    uint32_t d0=seed,d1=seed+1,d2=seed+2,d3=seed+3,d4=seed+4,d5=seed+5,d6=seed+6,d7=seed+7, t0=0, t1=0; \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55aa55aau, e = b + 7u;               \
    uint64_t c0 = __builtin_readcyclecounter();                                                      \
    for (int i = 0; i < ITER; ++i) {                                                                 \
        asm volatile(REP4(BODY)                                                                      \
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), \
                       "+v"(t0), "+v"(t1), "+v"(a), "+v"(b), "+v"(c), "+v"(e));                       \
    }                                                                                                \
    uint64_t c1 = __builtin_readcyclecounter();                                                      \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0^d1^d2^d3^d4^d5^d6^d7^t0^t1^a^b^c^e;              \
}
KERNEL(k_shared, BODY_SHARED, 1) KERNEL(k_rot, BODY_ROT, 1) KERNEL(k_and, BODY_ANDONLY, 1) KERNEL(k_bcnt, BODY_BCNTONLY, 1) KERNEL(k_ripple, BODY_RIPPLE, 1)
struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t, uint64_t*); int n_instr; };
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    uint64_t* cyc; CHECK(hipMalloc(&cyc, 8));
    Entry es[] = {{"and+bcnt shared tmp", k_shared, 16}, {"and+bcnt 2 tmps", k_rot, 16}, {"and only", k_and, 16}, {"bcnt only", k_bcnt, 16}, {"bitop3 4 chains", k_ripple, 16}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-22s", "pattern (per instr)");
    for (int wps : {1, 2, 3, 4, 6}) printf("  w/SIMD=%d(cyc,ns)", wps);
    printf("\n");
    for (auto& e : es) {
        printf("%-22s", e.name);
        for (int wps : {1, 2, 3, 4, 6}) {
            dim3 grid(cus * wps), block(256);
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 1u, cyc);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 2u, cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            double n = (double)ITER * 4 * e.n_instr;          // instructions per wave
            printf("   %5.2f %6.3f   ", (double)c / n / wps, ms * 1e6 / n / wps);
        }
        printf("\n");
    }
    return 0;
}
