// ubench8.hip -- (development tool) which wave supplies the co-issued second VALU instruction on gfx950?
// one block per CU, 4 waves per role (one per SIMD), fixed roles by wave index / 4: roles 0 and 1 run only half-rate v_bcnt at priority 3,
// role 2 runs only full-rate v_and at priority 0.  If the and-wave rides along in the bcnt waves' slots the kernel
// takes the time of the bcnt work alone (2 x N x 4 cycles per SIMD); if the arbiter only ever looks at the next
// high-priority wave it takes longer.  Variants: and-wave count 1 or 2 (roles 2,3 of 4), bcnt waves at equal priority.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
constexpr int ITER = 4000;
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
// nb = number of bcnt roles, na = number of and roles (roles cycle with the block index); mode bit 0: bcnt waves raise priority
__global__ void __launch_bounds__(1024) k_roles(uint32_t* out, uint32_t seed, int nb, int na, int mode) {
    uint32_t d0=seed,d1=seed+1,d2=seed+2,d3=seed+3,d4=seed+4,d5=seed+5,d6=seed+6,d7=seed+7, t0=0, t1=0;
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55aa55aau, e = b + 7u;
    const int role = threadIdx.x >> 8;   // waves w, w+4, w+8 ... of a block share a SIMD: one role per group of four waves
    if (role < nb) {
        if (mode & 1) asm volatile("s_setprio 3");
        for (int i = 0; i < ITER; ++i) ASMV(R32(B1 B2, B1 B2));
    } else {
        for (int i = 0; i < ITER; ++i) ASMV(R32(A1 A2, A1 A2));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0^d1^d2^d3^d4^d5^d6^d7^t0^t1^a^b^c^e;
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 16 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("ns per 64-instruction body of ONE bcnt wave (112 = a bcnt wave alone; n bcnt waves serialise: n x 112)\n");
    struct { int nb, na, mode; const char* name; } cfg[] = {
        {1, 0, 1, "1 bcnt wave"}, {2, 0, 1, "2 bcnt waves"}, {1, 1, 1, "1 bcnt(prio3) + 1 and"}, {2, 1, 1, "2 bcnt(prio3) + 1 and"},
        {2, 2, 1, "2 bcnt(prio3) + 2 and"}, {2, 1, 0, "2 bcnt(prio0) + 1 and"}, {1, 1, 0, "1 bcnt(prio0) + 1 and"}, {1, 2, 1, "1 bcnt(prio3) + 2 and"},
        {0, 1, 0, "1 and wave"}, {0, 2, 0, "2 and waves"}, {0, 3, 0, "3 and waves"}};
    for (auto& c : cfg) {
        const int wps = c.nb + c.na;
        dim3 grid(cus), block(256 * wps);
        hipLaunchKernelGGL(k_roles, grid, block, 0, 0, out, 1u, c.nb, c.na, c.mode);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_roles, grid, block, 0, 0, out, 2u, c.nb, c.na, c.mode);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-26s kernel %8.3f ms -> %7.1f ns per body\n", c.name, ms, ms * 1e6 / ITER);
    }
    return 0;
}
