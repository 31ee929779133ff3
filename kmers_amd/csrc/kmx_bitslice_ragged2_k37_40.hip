// kmx_bitslice_ragged2_k37_40.hip -- bit-sliced scan instantiations for ragged reads, two-word k = 37 .. 40 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR2_DEFINE_K(37)
KMX_BSR2_DEFINE_K(38)
KMX_BSR2_DEFINE_K(39)
KMX_BSR2_DEFINE_K(40)

}  // namespace kmx
