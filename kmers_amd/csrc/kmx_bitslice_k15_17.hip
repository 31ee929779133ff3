// kmx_bitslice_k15_17.hip -- bit-sliced scan instantiations for k = 15, 17 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(15, false)
KMX_BS_DEFINE_K(17, false)

}  // namespace kmx
