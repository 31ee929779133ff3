// kmx_bitslice_k21.hip -- bit-sliced scan instantiations for k = 21 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(21, true)

}  // namespace kmx
