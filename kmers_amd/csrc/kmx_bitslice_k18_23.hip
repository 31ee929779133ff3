// kmx_bitslice_k18_23.hip -- bit-sliced scan instantiations for k = 18, 19, 20, 22, 23 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(18, true)
KMX_BS_DEFINE_K(19, true)
KMX_BS_DEFINE_K(20, true)
KMX_BS_DEFINE_K(22, true)
KMX_BS_DEFINE_K(23, true)

}  // namespace kmx
