// kmx_hist32.hip -- the scan-kernel instantiations of the partition sink with 32-bit entries (first level of the two-level
// partitioned histogram, 2^23..2^28 buckets); see kmx_hist_part.h / kmx_hist.hip.
#include "kmx_hist_part.h"

namespace kmx {

hipError_t dispatch_part_u32(const uint8_t* bases, u64 n_reads, u32 L, u32 k, HistPartParams& p, unsigned long long* queue,
                             int n_cu, hipStream_t stream, HistPartPre pre, const u64* offsets) {
    return dispatch_part<HistPartPre, u32>(bases, n_reads, L, k, p, queue, n_cu, stream, pre, offsets);
}

}  // namespace kmx
