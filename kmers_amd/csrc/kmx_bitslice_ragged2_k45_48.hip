// kmx_bitslice_ragged2_k45_48.hip -- bit-sliced scan instantiations for ragged reads, two-word k = 45 .. 48 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR2_DEFINE_K(45)
KMX_BSR2_DEFINE_K(46)
KMX_BSR2_DEFINE_K(47)
KMX_BSR2_DEFINE_K(48)

}  // namespace kmx
