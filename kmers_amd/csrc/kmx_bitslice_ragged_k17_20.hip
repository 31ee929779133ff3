// kmx_bitslice_ragged_k17_20.hip -- bit-sliced scan instantiations for ragged reads, k = 17, 18, 19, 20 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR_DEFINE_K(17)
KMX_BSR_DEFINE_K(18)
KMX_BSR_DEFINE_K(19)
KMX_BSR_DEFINE_K(20)

}  // namespace kmx
