// kmx_bitslice_k19_23.hip -- bit-sliced scan instantiations for k = 19, 23 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(19, false)
KMX_BS_DEFINE_K(23, false)

}  // namespace kmx
