// kmx_bitslice_k29.hip -- bit-sliced scan instantiations for k = 29 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(29, false)

}  // namespace kmx
