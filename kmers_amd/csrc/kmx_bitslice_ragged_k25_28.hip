// kmx_bitslice_ragged_k25_28.hip -- bit-sliced scan instantiations for ragged reads, k = 25, 26, 27, 28 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR_DEFINE_K(25)
KMX_BSR_DEFINE_K(26)
KMX_BSR_DEFINE_K(27)
KMX_BSR_DEFINE_K(28)

}  // namespace kmx
