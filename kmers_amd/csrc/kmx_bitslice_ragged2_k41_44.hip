// kmx_bitslice_ragged2_k41_44.hip -- bit-sliced scan instantiations for ragged reads, two-word k = 41 .. 44 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR2_DEFINE_K(41)
KMX_BSR2_DEFINE_K(42)
KMX_BSR2_DEFINE_K(43)
KMX_BSR2_DEFINE_K(44)

}  // namespace kmx
