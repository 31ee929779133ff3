// kmx_bitslice_ragged_k29_31.hip -- bit-sliced scan instantiations for ragged reads, k = 29, 30, 31 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR_DEFINE_K(29)
KMX_BSR_DEFINE_K(30)
KMX_BSR_DEFINE_K(31)

}  // namespace kmx
