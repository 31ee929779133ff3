// kmx_bitslice_ragged2_k33_36.hip -- bit-sliced scan instantiations for ragged reads, two-word k = 33 .. 36 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR2_DEFINE_K(33)
KMX_BSR2_DEFINE_K(34)
KMX_BSR2_DEFINE_K(35)
KMX_BSR2_DEFINE_K(36)

}  // namespace kmx
