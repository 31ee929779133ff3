// kmx_bitslice_ragged_k21_24.hip -- bit-sliced scan instantiations for ragged reads, k = 21, 22, 23, 24 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR_DEFINE_K(21)
KMX_BSR_DEFINE_K(22)
KMX_BSR_DEFINE_K(23)
KMX_BSR_DEFINE_K(24)

}  // namespace kmx
