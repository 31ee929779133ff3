// kmx_bitslice_ragged2_k49_52.hip -- bit-sliced scan instantiations for ragged reads, two-word k = 49 .. 52 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR2_DEFINE_K(49)
KMX_BSR2_DEFINE_K(50)
KMX_BSR2_DEFINE_K(51)
KMX_BSR2_DEFINE_K(52)

}  // namespace kmx
