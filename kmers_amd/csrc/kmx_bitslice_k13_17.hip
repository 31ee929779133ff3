// kmx_bitslice_k13_17.hip -- bit-sliced scan instantiations for k = 13, 14, 15, 16, 17 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(13, true)
KMX_BS_DEFINE_K(14, true)
KMX_BS_DEFINE_K(15, true)
KMX_BS_DEFINE_K(16, true)
KMX_BS_DEFINE_K(17, true)

}  // namespace kmx
