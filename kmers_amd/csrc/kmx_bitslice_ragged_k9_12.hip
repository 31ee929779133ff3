// kmx_bitslice_ragged_k9_12.hip -- bit-sliced scan instantiations for ragged reads, k = 9, 10, 11, 12 (kernel: kmx_bitslice_kernel.h; round 6)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR_DEFINE_K(9)
KMX_BSR_DEFINE_K(10)
KMX_BSR_DEFINE_K(11)
KMX_BSR_DEFINE_K(12)

}  // namespace kmx
