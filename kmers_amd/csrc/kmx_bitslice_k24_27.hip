// kmx_bitslice_k24_27.hip -- bit-sliced scan instantiations for k = 24, 25, 26, 27 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(24, true)
KMX_BS_DEFINE_K(25, true)
KMX_BS_DEFINE_K(26, true)
KMX_BS_DEFINE_K(27, true)

}  // namespace kmx
