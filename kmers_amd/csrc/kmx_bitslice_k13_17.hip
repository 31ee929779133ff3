// kmx_bitslice_k13_17.hip -- bit-sliced scan instantiations for k = 13, 14, 15, 16, 17 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(13, false)
KMX_BS_DEFINE_K(14, false)
KMX_BS_DEFINE_K(15, false)
KMX_BS_DEFINE_K(16, false)
KMX_BS_DEFINE_K(17, false)

}  // namespace kmx
