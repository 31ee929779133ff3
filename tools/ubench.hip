// ubench.hip -- VALU instruction-throughput probe for gfx950 (development tool, not product).
// Each kernel runs ITER x 64 copies of one instruction pattern on 4 independent register sets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)

constexpr int ITER = 2000;

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL(NAME, ASM)                                                                       \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                      \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55aa55aau, d = b + 7u;           \
    uint32_t e = a + 11u, f = b ^ 5u, g = c + 13u, h = d ^ 17u;                                  \
    uint64_t p = ((uint64_t)a << 32) | b, q = ((uint64_t)c << 32) | d, r = p ^ q, s = p + q;     \
    for (int i = 0; i < ITER; ++i) {                                                            \
        asm volatile(REP16(ASM)                                                                  \
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h),   \
                       "+v"(p), "+v"(q), "+v"(r), "+v"(s)                                        \
                     : : "vcc");                                                                \
    }                                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)(p ^ q ^ r ^ s) ^ (uint32_t)((p^q^r^s) >> 32); \
}

// 4 instructions per ASM block (so 64 per loop iteration), independent destinations
KERNEL(k_alignbit, "v_alignbit_b32 %0, %1, %2, 6\n v_alignbit_b32 %2, %3, %0, 10\n v_alignbit_b32 %4, %5, %6, 12\n v_alignbit_b32 %6, %7, %4, 14\n")
KERNEL(k_and,      "v_and_b32 %0, %1, %0\n v_and_b32 %2, %3, %2\n v_and_b32 %4, %5, %4\n v_and_b32 %6, %7, %6\n")
KERNEL(k_perm,     "v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %2, %3, %0, %1\n v_perm_b32 %4, %5, %6, %7\n v_perm_b32 %6, %7, %4, %5\n")
KERNEL(k_dot4,     "v_dot4_u32_u8 %0, %1, %2, %0\n v_dot4_u32_u8 %2, %3, %1, %2\n v_dot4_u32_u8 %4, %5, %6, %4\n v_dot4_u32_u8 %6, %7, %5, %6\n")
KERNEL(k_bfrev,    "v_bfrev_b32 %0, %1\n v_bfrev_b32 %2, %3\n v_bfrev_b32 %4, %5\n v_bfrev_b32 %6, %7\n")
KERNEL(k_cmp64,    "v_cmp_lt_u64 vcc, %8, %9\n v_cmp_lt_u64 vcc, %10, %11\n v_cmp_lt_u64 vcc, %9, %10\n v_cmp_lt_u64 vcc, %11, %8\n")
KERNEL(k_cmp32,    "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %6, %7\n")
KERNEL(k_cmp64_cnd,"v_cmp_lt_u64 vcc, %8, %9\n s_nop 0\n v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_and_b32 %6, %7, %6\n")
KERNEL(k_cmp32_cnd,"v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_and_b32 %6, %7, %6\n")
KERNEL(k_add64,    "v_lshl_add_u64 %8, %9, 0, %8\n v_lshl_add_u64 %10, %11, 0, %10\n v_lshl_add_u64 %9, %8, 0, %9\n v_lshl_add_u64 %11, %10, 0, %11\n")
KERNEL(k_addco,    "v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, %3, %2, vcc\n v_add_co_u32 %4, vcc, %5, %4\n v_addc_co_u32 %6, vcc, %7, %6, vcc\n")
KERNEL(k_subco,    "v_sub_co_u32 %0, vcc, %1, %2\n v_subb_co_u32 %3, vcc, %4, %5, vcc\n v_sub_co_u32 %6, vcc, %7, %2\n v_subb_co_u32 %3, vcc, %4, %5, vcc\n")
KERNEL(k_cndmask,  "v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %6, %7, %1, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n")
KERNEL(k_lshr64,   "v_lshrrev_b64 %8, 2, %9\n v_lshrrev_b64 %10, 2, %11\n v_lshrrev_b64 %9, 4, %8\n v_lshrrev_b64 %11, 6, %10\n")
KERNEL(k_mad64,    "v_mad_u64_u32 %8, vcc, %0, %1, %8\n v_mad_u64_u32 %10, vcc, %2, %3, %10\n v_mad_u64_u32 %9, vcc, %4, %5, %9\n v_mad_u64_u32 %11, vcc, %6, %7, %11\n")
KERNEL(k_xor3,     "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96\n v_bitop3_b32 %3, %4, %5, %3 bitop3:0x96\n v_bitop3_b32 %6, %7, %1, %6 bitop3:0x96\n v_bitop3_b32 %2, %4, %5, %2 bitop3:0x96\n")
KERNEL(k_or3,      "v_or3_b32 %0, %1, %2, %0\n v_or3_b32 %3, %4, %5, %3\n v_or3_b32 %6, %7, %1, %6\n v_or3_b32 %2, %4, %5, %2\n")
KERNEL(k_andor,    "v_and_or_b32 %0, %1, %2, %0\n v_and_or_b32 %3, %4, %5, %3\n v_and_or_b32 %6, %7, %1, %6\n v_and_or_b32 %2, %4, %5, %2\n")
KERNEL(k_lshlor,   "v_lshl_or_b32 %0, %1, 8, %0\n v_lshl_or_b32 %3, %4, 8, %3\n v_lshl_or_b32 %6, %7, 8, %6\n v_lshl_or_b32 %2, %4, 8, %2\n")
KERNEL(k_bfi,      "v_bfi_b32 %0, %1, %2, %0\n v_bfi_b32 %3, %4, %5, %3\n v_bfi_b32 %6, %7, %1, %6\n v_bfi_b32 %2, %4, %5, %2\n")
KERNEL(k_min,      "v_min_u32 %0, %1, %0\n v_min_u32 %2, %3, %2\n v_min_u32 %4, %5, %4\n v_min_u32 %6, %7, %6\n")
KERNEL(k_add3,     "v_add3_u32 %0, %1, %2, %0\n v_add3_u32 %3, %4, %5, %3\n v_add3_u32 %6, %7, %1, %6\n v_add3_u32 %2, %4, %5, %2\n")
KERNEL(k_alignbit_s,"v_alignbit_b32 %0, %1, %2, %3\n v_alignbit_b32 %2, %3, %0, %1\n v_alignbit_b32 %4, %5, %6, %7\n v_alignbit_b32 %6, %7, %4, %5\n")
KERNEL(k_bfe,      "v_bfe_u32 %0, %1, 3, 30\n v_bfe_u32 %2, %3, 5, 20\n v_bfe_u32 %4, %5, 7, 12\n v_bfe_u32 %6, %7, 1, 30\n")
KERNEL(k_movb64,   "v_mov_b64 %8, %9\n v_mov_b64 %10, %11\n v_mov_b64 %9, %8\n v_mov_b64 %11, %10\n")
KERNEL(k_pkadd,    "v_pk_add_u16 %0, %1, %0\n v_pk_add_u16 %2, %3, %2\n v_pk_add_u16 %4, %5, %4\n v_pk_add_u16 %6, %7, %6\n")
KERNEL(k_snop,     "v_and_b32 %0, %1, %0\n s_nop 0\n v_and_b32 %2, %3, %2\n s_nop 0\n v_and_b32 %4, %5, %4\n s_nop 0\n v_and_b32 %6, %7, %6\n s_nop 0\n")

struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t); int instr_per_block; };

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    std::vector<Entry> es = {
        {"v_alignbit_b32(imm)", k_alignbit, 4}, {"v_alignbit_b32(vgpr sh)", k_alignbit_s, 4}, {"v_and_b32", k_and, 4},
        {"v_perm_b32", k_perm, 4}, {"v_dot4_u32_u8", k_dot4, 4}, {"v_bfrev_b32", k_bfrev, 4}, {"v_bfe_u32", k_bfe, 4},
        {"v_cmp_lt_u64", k_cmp64, 4}, {"v_cmp_lt_u32", k_cmp32, 4},
        {"cmp64+nop+2cnd+and (4 valu)", k_cmp64_cnd, 4}, {"cmp32+2cnd+and (4 valu)", k_cmp32_cnd, 4},
        {"v_lshl_add_u64", k_add64, 4}, {"v_add_co+v_addc_co", k_addco, 4}, {"v_sub_co+v_subb_co", k_subco, 4},
        {"v_cndmask_b32", k_cndmask, 4}, {"v_lshrrev_b64", k_lshr64, 4}, {"v_mad_u64_u32", k_mad64, 4},
        {"v_bitop3_b32", k_xor3, 4}, {"v_or3_b32", k_or3, 4}, {"v_and_or_b32", k_andor, 4}, {"v_lshl_or_b32", k_lshlor, 4},
        {"v_bfi_b32", k_bfi, 4}, {"v_min_u32", k_min, 4}, {"v_add3_u32", k_add3, 4}, {"v_mov_b64", k_movb64, 4},
        {"v_pk_add_u16", k_pkadd, 4}, {"v_and + s_nop 0 (4 valu)", k_snop, 4},
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD: 256-thread blocks -> 1 wave/SIMD per block
        printf("---- %d block(s)/CU = %d wave(s)/SIMD\n", wps, wps);
        for (auto& e : es) {
            dim3 grid(cus * wps), block(256);
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 1u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 2u);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double instr_per_wave = (double)ITER * 16 * e.instr_per_block;
            // cycles per wave-instruction per SIMD, assuming 2.4 GHz (report also ns)
            double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * wps);
            printf("%-30s %8.3f ms   %.3f ns per wave-instr per SIMD  (= %.2f cyc @2.4GHz)\n", e.name, ms,
                   ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
        }
    }
    return 0;
}
