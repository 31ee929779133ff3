// ubench2.hip -- systematic VALU issue-rate table for gfx950 (development tool).
// Every pattern writes 8 different destination registers from read-only sources, so there are no
// RAW dependencies between consecutive instructions; reported = shader cycles per wave64 instruction
// per SIMD at 1/2/4/8 waves per SIMD (clock measured with s_memtime against wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
constexpr int ITER = 1000;
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

// OP(d) expands to one instruction writing %d; sources: %8 %9 %10 %11 (32-bit), %12 %13 (64-bit), s: %14 %15
#define KERNEL(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                               \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed, uint64_t* cyc) {          \
    uint32_t d0=seed,d1=seed+1,d2=seed+2,d3=seed+3,d4=seed+4,d5=seed+5,d6=seed+6,d7=seed+7;          \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55aa55aau, e = b + 7u;               \
    uint64_t p = ((uint64_t)a << 32) | b, q = ((uint64_t)c << 32) | e;                               \
    uint64_t P0=p,P1=q,P2=p+1,P3=q+1;                                                                \
    uint32_t s0 = seed & 15u, s1 = seed | 3u;                                                        \
    uint64_t t0 = __builtin_readcyclecounter();                                                      \
    for (int i = 0; i < ITER; ++i) {                                                                 \
        asm volatile(REP8(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n")           \
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), \
                       "+v"(a), "+v"(b), "+v"(c), "+v"(e), "+v"(p), "+v"(q), "+s"(s0), "+s"(s1),       \
                       "+v"(P0), "+v"(P1), "+v"(P2), "+v"(P3)                                         \
                     : : "vcc");                                                                     \
    }                                                                                                \
    uint64_t t1 = __builtin_readcyclecounter();                                                      \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0^d1^d2^d3^d4^d5^d6^d7^a^b^c^e^(uint32_t)(p^q^P0^P1^P2^P3)^(uint32_t)((P0^P1^P2^P3)>>32); \
}
#define K8(NAME, OPFMT) KERNEL(NAME, OPFMT(0), OPFMT(1), OPFMT(2), OPFMT(3), OPFMT(4), OPFMT(5), OPFMT(6), OPFMT(7))

#define F_AND(d)      "v_and_b32 %" #d ", %8, %9"
#define F_OR(d)       "v_or_b32 %" #d ", %8, %9"
#define F_XOR(d)      "v_xor_b32 %" #d ", %8, %9"
#define F_NOT(d)      "v_not_b32 %" #d ", %8"
#define F_MOV(d)      "v_mov_b32 %" #d ", %8"
#define F_LSHL(d)     "v_lshlrev_b32 %" #d ", 5, %8"
#define F_LSHR(d)     "v_lshrrev_b32 %" #d ", 5, %8"
#define F_LSHRV(d)    "v_lshrrev_b32 %" #d ", %9, %8"
#define F_ADD(d)      "v_add_u32 %" #d ", %8, %9"
#define F_SUB(d)      "v_sub_u32 %" #d ", %8, %9"
#define F_ADDCO(d)    "v_add_co_u32 %" #d ", vcc, %8, %9"
#define F_ADDC(d)     "v_addc_co_u32 %" #d ", vcc, %8, %9, vcc"
#define F_CND(d)      "v_cndmask_b32 %" #d ", %8, %9, vcc"
#define F_ALIGN(d)    "v_alignbit_b32 %" #d ", %8, %9, 6"
#define F_ALIGNS(d)   "v_alignbit_b32 %" #d ", %8, %9, %14"
#define F_ALIGNBYTE(d) "v_alignbyte_b32 %" #d ", %8, %9, 1"
#define F_PERM(d)     "v_perm_b32 %" #d ", %8, %9, %10"
#define F_BFE(d)      "v_bfe_u32 %" #d ", %8, 3, 30"
#define F_BFI(d)      "v_bfi_b32 %" #d ", %8, %9, %10"
#define F_BITOP3(d)   "v_bitop3_b32 %" #d ", %8, %9, %10 bitop3:0x96"
#define F_ANDOR(d)    "v_and_or_b32 %" #d ", %8, %9, %10"
#define F_OR3(d)      "v_or3_b32 %" #d ", %8, %9, %10"
#define F_LSHLOR(d)   "v_lshl_or_b32 %" #d ", %8, 8, %9"
#define F_LSHLADD(d)  "v_lshl_add_u32 %" #d ", %8, 8, %9"
#define F_ADD3(d)     "v_add3_u32 %" #d ", %8, %9, %10"
#define F_XAD(d)      "v_xad_u32 %" #d ", %8, %9, %10"
#define F_MIN(d)      "v_min_u32 %" #d ", %8, %9"
#define F_MAX(d)      "v_max_u32 %" #d ", %8, %9"
#define F_MIN3(d)     "v_min3_u32 %" #d ", %8, %9, %10"
#define F_CMP32(d)    "v_cmp_lt_u32 vcc, %8, %9"
#define F_CMP64(d)    "v_cmp_lt_u64 vcc, %12, %13"
#define F_CMPX(d)     "v_cmp_lt_u32 s[20:21], %8, %9"
#define F_DOT4(d)     "v_dot4_u32_u8 %" #d ", %8, %9, %10"
#define F_BFREV(d)    "v_bfrev_b32 %" #d ", %8"
#define F_MUL24(d)    "v_mul_u32_u24 %" #d ", %8, %9"
#define F_MAD24(d)    "v_mad_u32_u24 %" #d ", %8, %9, %10"
#define F_MULLO(d)    "v_mul_lo_u32 %" #d ", %8, %9"
#define F_FMA(d)      "v_fma_f32 %" #d ", %8, %9, %10"
#define F_FMAC(d)     "v_fmac_f32 %" #d ", %8, %9"
#define F_ADDF(d)     "v_add_f32 %" #d ", %8, %9"
#define F_PKADD16(d)  "v_pk_add_u16 %" #d ", %8, %9"
#define F_PKLSHR16(d) "v_pk_lshrrev_b16 %" #d ", %8, %9"
#define F_PKMIN16(d)  "v_pk_min_u16 %" #d ", %8, %9"
#define F_SAD(d)      "v_sad_u32 %" #d ", %8, %9, %10"
#define F_BCNT(d)     "v_bcnt_u32_b32 %" #d ", %8, %9"
#define F_ANDSGPR(d)  "v_and_b32 %" #d ", %14, %8"
#define F_ANDLIT(d)   "v_and_b32 %" #d ", 0x3fffffff, %8"
#define F_ANDE64(d)   "v_and_b32_e64 %" #d ", %8, %9"
#define F_ALIGN_DPP(d) "v_mov_b32_dpp %" #d ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"
// 64-bit destination ops: use the four 64-bit regs %16..%19 (two per instruction pair)
#define KERNEL64(NAME, I0, I1, I2, I3) KERNEL(NAME, I0, I1, I2, I3, I0, I1, I2, I3)
#define F_ADD64_0 "v_lshl_add_u64 %16, %12, 0, %13"
#define F_ADD64_1 "v_lshl_add_u64 %17, %12, 0, %13"
#define F_ADD64_2 "v_lshl_add_u64 %18, %12, 0, %13"
#define F_ADD64_3 "v_lshl_add_u64 %19, %12, 0, %13"
#define F_LSHR64_0 "v_lshrrev_b64 %16, 2, %12"
#define F_LSHR64_1 "v_lshrrev_b64 %17, 4, %12"
#define F_LSHR64_2 "v_lshrrev_b64 %18, 6, %13"
#define F_LSHR64_3 "v_lshrrev_b64 %19, 8, %13"
#define F_MOV64_0 "v_mov_b64 %16, %12"
#define F_MOV64_1 "v_mov_b64 %17, %13"
#define F_MOV64_2 "v_mov_b64 %18, %12"
#define F_MOV64_3 "v_mov_b64 %19, %13"
#define F_PKADD32_0 "v_pk_add_f32 %16, %12, %13"
#define F_PKADD32_1 "v_pk_add_f32 %17, %12, %13"
#define F_PKADD32_2 "v_pk_add_f32 %18, %12, %13"
#define F_PKADD32_3 "v_pk_add_f32 %19, %12, %13"
#define F_MAD64_0 "v_mad_u64_u32 %16, vcc, %8, %9, %12"
#define F_MAD64_1 "v_mad_u64_u32 %17, vcc, %8, %9, %12"
#define F_MAD64_2 "v_mad_u64_u32 %18, vcc, %8, %9, %13"
#define F_MAD64_3 "v_mad_u64_u32 %19, vcc, %8, %9, %13"

K8(k_and, F_AND) K8(k_or, F_OR) K8(k_xor, F_XOR) K8(k_not, F_NOT) K8(k_mov, F_MOV) K8(k_lshl, F_LSHL) K8(k_lshr, F_LSHR)
K8(k_lshrv, F_LSHRV) K8(k_add, F_ADD) K8(k_sub, F_SUB) K8(k_addco, F_ADDCO) K8(k_addc, F_ADDC) K8(k_cnd, F_CND)
K8(k_align, F_ALIGN) K8(k_aligns, F_ALIGNS) K8(k_alignbyte, F_ALIGNBYTE) K8(k_perm, F_PERM) K8(k_bfe, F_BFE) K8(k_bfi, F_BFI)
K8(k_bitop3, F_BITOP3) K8(k_andor, F_ANDOR) K8(k_or3, F_OR3) K8(k_lshlor, F_LSHLOR) K8(k_lshladd, F_LSHLADD)
K8(k_add3, F_ADD3) K8(k_xad, F_XAD) K8(k_min, F_MIN) K8(k_max, F_MAX) K8(k_min3, F_MIN3) K8(k_cmp32, F_CMP32)
K8(k_cmp64, F_CMP64) K8(k_cmpx, F_CMPX) K8(k_dot4, F_DOT4) K8(k_bfrev, F_BFREV) K8(k_mul24, F_MUL24) K8(k_mad24, F_MAD24)
K8(k_mullo, F_MULLO) K8(k_fma, F_FMA) K8(k_fmac, F_FMAC) K8(k_addf, F_ADDF) K8(k_pkadd16, F_PKADD16)
K8(k_pklshr16, F_PKLSHR16) K8(k_pkmin16, F_PKMIN16) K8(k_sad, F_SAD) K8(k_bcnt, F_BCNT) K8(k_andsgpr, F_ANDSGPR)
K8(k_andlit, F_ANDLIT) K8(k_ande64, F_ANDE64) K8(k_dpp, F_ALIGN_DPP)
KERNEL64(k_add64, F_ADD64_0, F_ADD64_1, F_ADD64_2, F_ADD64_3)
KERNEL64(k_lshr64, F_LSHR64_0, F_LSHR64_1, F_LSHR64_2, F_LSHR64_3)
KERNEL64(k_mov64, F_MOV64_0, F_MOV64_1, F_MOV64_2, F_MOV64_3)
KERNEL64(k_pkaddf32, F_PKADD32_0, F_PKADD32_1, F_PKADD32_2, F_PKADD32_3)
KERNEL64(k_mad64, F_MAD64_0, F_MAD64_1, F_MAD64_2, F_MAD64_3)

struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t, uint64_t*); };

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    uint64_t* cyc; CHECK(hipMalloc(&cyc, 8));
    std::vector<Entry> es = {
#define E(n) {#n, n}
        E(k_and), E(k_or), E(k_xor), E(k_not), E(k_mov), E(k_lshl), E(k_lshr), E(k_lshrv), E(k_add), E(k_sub), E(k_addco), E(k_addc),
        E(k_cnd), E(k_align), E(k_aligns), E(k_alignbyte), E(k_perm), E(k_bfe), E(k_bfi), E(k_bitop3), E(k_andor), E(k_or3),
        E(k_lshlor), E(k_lshladd), E(k_add3), E(k_xad), E(k_min), E(k_max), E(k_min3), E(k_cmp32), E(k_cmp64), E(k_cmpx), E(k_dot4),
        E(k_bfrev), E(k_mul24), E(k_mad24), E(k_mullo), E(k_fma), E(k_fmac), E(k_addf), E(k_pkadd16), E(k_pklshr16), E(k_pkmin16),
        E(k_sad), E(k_bcnt), E(k_andsgpr), E(k_andlit), E(k_ande64), E(k_dpp), E(k_add64), E(k_lshr64), E(k_mov64), E(k_pkaddf32), E(k_mad64),
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-14s", "instr");
    for (int wps : {1, 2, 4, 8}) printf("  w/SIMD=%d(cyc,ns) ", wps);
    printf("\n");
    for (auto& e : es) {
        printf("%-14s", e.name);
        for (int wps : {1, 2, 4, 8}) {
            dim3 grid(cus * wps), block(256);
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 1u, cyc);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, grid, block, 0, 0, out, 2u, cyc);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t hc; CHECK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
            double n_instr = (double)ITER * 64;  // per wave
            // cycles (s_memtime ticks) per instruction per SIMD = ticks / (n_instr * wps)
            printf("  %6.2f %6.3f     ", (double)hc / (n_instr * wps), ms * 1e6 / (n_instr * wps));
        }
        printf("\n");
    }
    return 0;
}
