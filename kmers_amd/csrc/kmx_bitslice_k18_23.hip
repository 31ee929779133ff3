// kmx_bitslice_k18_23.hip -- bit-sliced scan instantiations for k = 18, 19, 20, 22, 23 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(18, false)
KMX_BS_DEFINE_K(19, false)
KMX_BS_DEFINE_K(20, false)
KMX_BS_DEFINE_K(22, false)
KMX_BS_DEFINE_K(23, false)

}  // namespace kmx
