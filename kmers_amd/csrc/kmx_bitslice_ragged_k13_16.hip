// kmx_bitslice_ragged_k13_16.hip -- bit-sliced scan instantiations for ragged reads, k = 13, 14, 15, 16 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR_DEFINE_K(13)
KMX_BSR_DEFINE_K(14)
KMX_BSR_DEFINE_K(15)
KMX_BSR_DEFINE_K(16)

}  // namespace kmx
