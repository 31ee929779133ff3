// kmx_bitslice.hip -- launchers of the bit-sliced canonical k-mer scan (kernel: kmx_bitslice_kernel.h); this
// translation unit holds the k = 31 and k = 63 instantiations, kmx_bitslice_k*.hip hold the other k.
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(31, true)

// `packed`: a SeqVector (16-byte aligned words).  ASCII reads may start anywhere: a base that is not 16-byte aligned costs
// a tile one more chunk, which the frames hold except for their very longest reads (L = 160 / 256)
static bool bs_domain(const void* p, u64 n_reads, u32 L, u32 k, bool packed) {
    if (L > 256 && !packed) return (reinterpret_cast<uintptr_t>(p) & 15u) == 0u && n_reads * (u64)L < (1ull << 62);   // (segments of long reads: launch_bs_seg)
    if (L < k || L > 256) return false;
    if (reinterpret_cast<uintptr_t>(p) & 15u) {
        if (packed || L == 160 || L == 256) return false;
    }
    return n_reads * (u64)L < (1ull << 62);
}

#define KMX_BS_CASE(K) \
    case K:            \
        *handled = true; \
        return launch_bs_k##K(bases, n_reads, L, false, want_hash, want_sumfw, out, queue, n_cu, stream);

// handled=false when (L,k) is outside the instantiated bit-sliced kernels
// `queue`: 32 zeroed u64 tile-queue heads, 128 B apart, owned by the caller for the duration of the launch
hipError_t launch_scan_bitsliced(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, u32 want_sumfw /* KMX_BS_*: bit 0 = sum_fw, the rest how the launch ends */,
                                 kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled) {
    *handled = false;
    if (!bs_domain(bases, n_reads, L, k, false)) return hipSuccess;
    switch (k) {
        KMX_BS_FOR_EACH_K(KMX_BS_CASE)
        default:
            return hipSuccess;
    }
}

// SeqVector input (kmx_seqvec.hip): read r = bases [r*L, (r+1)*L) of the 2-bit packed vector `words` (16-byte aligned)
hipError_t launch_scan_bitsliced_packed(const uint64_t* words, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw,
                                        kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled) {
    *handled = false;
    const uint8_t* bases = reinterpret_cast<const uint8_t*>(words);
    if (!bs_domain(bases, n_reads, L, k, true)) return hipSuccess;
#define KMX_BS_PCASE(K) \
    case K:             \
        *handled = true; \
        return launch_bs_k##K(bases, n_reads, L, true, want_hash, want_sumfw, out, queue, n_cu, stream);
    switch (k) {
        KMX_BS_FOR_EACH_K(KMX_BS_PCASE)
        default:
            return hipSuccess;
    }
}

KMX_BS2_DEFINE_K(63)

// Ragged reads (offsets array): `L_hint` = upper bound of the read lengths if the caller knows one (0 = unknown -> the
// 160-base frame; tiles holding a longer read roll per lane).
hipError_t launch_scan_bitsliced_ragged(const uint8_t* bases, const u64* offsets, u64 n_reads, u32 L_hint, u32 k, bool want_hash,
                                        kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                                        bool want_sumfw, const u64* ends) {
    *handled = false;
    if (!offsets || (reinterpret_cast<uintptr_t>(bases) & 15u) || L_hint > 256) return hipSuccess;
    u32 Lf = L_hint ? L_hint : 160u;
    if (Lf < k + 15u) Lf = k + 15u;       // keep at least 16 windows in the frame
    if (Lf > 256u) return hipSuccess;
#define KMX_BSR_CASE(K) \
    case K:             \
        *handled = true; \
        return launch_bs_ragged_k##K(bases, offsets, n_reads, Lf, want_hash, out, queue, n_cu, stream, want_sumfw ? 1u : 0u, ends);
    switch (k) {
        KMX_BSR_FOR_EACH_K(KMX_BSR_CASE)
        default:
            return hipSuccess;
    }
}

// Ragged reads, two-word k (round 4): the 10-word frame only -- a length bound above 160 leaves the call to the lane-per-read kernel
hipError_t launch_scan_bitsliced2_ragged(const uint8_t* bases, const u64* offsets, u64 n_reads, u32 L_hint, u32 k, bool want_hash,
                                         kmx_summary2* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled,
                                         const u64* ends) {
    *handled = false;
    if (!offsets || (reinterpret_cast<uintptr_t>(bases) & 15u) || L_hint > 160) return hipSuccess;
    u32 Lf = L_hint ? L_hint : 160u;
    if (Lf < k + 15u) Lf = k + 15u;       // keep at least 16 windows in the frame
    if (Lf > 160u) return hipSuccess;
#define KMX_BSR2_CASE(K) \
    case K:              \
        *handled = true; \
        return launch_bs2_ragged_k##K(bases, offsets, n_reads, Lf, want_hash, out, queue, n_cu, stream, ends);
    switch (k) {
        KMX_BS2_FOR_EACH_K(KMX_BSR2_CASE)
        default:
            return hipSuccess;
    }
}

// [u64;2] k-mers: every k from 33 to 64 is instantiated (k = 63 is BASELINE configs[2])
hipError_t launch_scan_bitsliced2(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, kmx_summary2* out,
                                  unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled) {
    *handled = false;
    if (L < k || ((reinterpret_cast<uintptr_t>(bases) & 15u) && (L == 160 || L >= 256))) return hipSuccess;
    if (n_reads * (u64)L >= (1ull << 62)) return hipSuccess;
#define KMX_BS2_CASE(K) \
    case K:             \
        *handled = true; \
        return launch_bs2_k##K(bases, n_reads, L, want_hash, out, queue, n_cu, stream);
    switch (k) {
        KMX_BS2_FOR_EACH_K(KMX_BS2_CASE)
        default:
            return hipSuccess;
    }
}

// segments a uniform read of L bases is cut into by the bit-sliced scan (1: none, it fits a frame): what the caller sizes the
// dirty-read masks from
u64 bitsliced_segments_per_read(u32 L, u32 k) { return (L > 256u && k >= 9u && k <= 64u && k != 32u) ? bs_seg_plan(L, k).J : 1u; }

}  // namespace kmx
