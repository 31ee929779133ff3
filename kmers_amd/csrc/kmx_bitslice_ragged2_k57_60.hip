// kmx_bitslice_ragged2_k57_60.hip -- bit-sliced scan instantiations for ragged reads, two-word k = 57 .. 60 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR2_DEFINE_K(57)
KMX_BSR2_DEFINE_K(58)
KMX_BSR2_DEFINE_K(59)
KMX_BSR2_DEFINE_K(60)

}  // namespace kmx
