// kmx_bitslice_ragged2_k53_56.hip -- bit-sliced scan instantiations for ragged reads, two-word k = 53 .. 56 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR2_DEFINE_K(53)
KMX_BSR2_DEFINE_K(54)
KMX_BSR2_DEFINE_K(55)
KMX_BSR2_DEFINE_K(56)

}  // namespace kmx
