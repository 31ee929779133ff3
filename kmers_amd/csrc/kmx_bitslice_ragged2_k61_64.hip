// kmx_bitslice_ragged2_k61_64.hip -- bit-sliced scan instantiations for ragged reads, two-word k = 61 .. 64 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR2_DEFINE_K(61)
KMX_BSR2_DEFINE_K(62)
KMX_BSR2_DEFINE_K(63)
KMX_BSR2_DEFINE_K(64)

}  // namespace kmx
