// kmx_bitslice.hip -- launchers of the bit-sliced canonical k-mer scan (kernel: kmx_bitslice_kernel.h); this
// translation unit holds the k = 31 and k = 63 instantiations, kmx_bitslice_k*.hip hold the other k.
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(31, true)

static bool bs_domain(const void* p, u64 n_reads, u32 L, u32 k) {
    if (L < k || L > 256 || (reinterpret_cast<uintptr_t>(p) & 15u)) return false;
    return n_reads * (u64)L < (1ull << 62);
}

#define KMX_BS_CASE(K) \
    case K:            \
        *handled = true; \
        return launch_bs_k##K(bases, n_reads, L, false, want_hash, want_sumfw, out, queue, n_cu, stream);

// handled=false when (L,k) is outside the instantiated bit-sliced kernels
// `queue`: 32 zeroed u64 tile-queue heads, 128 B apart, owned by the caller for the duration of the launch
hipError_t launch_scan_bitsliced(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, bool want_sumfw,
                                 kmx_summary* out, unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled) {
    *handled = false;
    if (!bs_domain(bases, n_reads, L, k)) return hipSuccess;
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
    if (!bs_domain(bases, n_reads, L, k)) return hipSuccess;
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

// [u64;2] k-mers: k = 63 is instantiated (BASELINE configs[2]); other k in 33..64 take the generic kernel
hipError_t launch_scan_bitsliced2(const uint8_t* bases, u64 n_reads, u32 L, u32 k, bool want_hash, kmx_summary2* out,
                                  unsigned long long* queue, int n_cu, hipStream_t stream, bool* handled) {
    *handled = false;
    if (k != 63 || L < k || L > 160 || (reinterpret_cast<uintptr_t>(bases) & 15u)) return hipSuccess;
    if (n_reads * (u64)L >= (1ull << 62)) return hipSuccess;
    const u32 W = L - k + 1u;
    *handled = true;
    if (W <= 64u) return launch_bs<63, 10, 2>(bases, n_reads, L, want_hash, 0, out, queue, n_cu, stream);
    return launch_bs<63, 10, 3>(bases, n_reads, L, want_hash, 0, out, queue, n_cu, stream);
}

}  // namespace kmx
