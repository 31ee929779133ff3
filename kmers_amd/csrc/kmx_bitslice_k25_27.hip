// kmx_bitslice_k25_27.hip -- bit-sliced scan instantiations for k = 25, 27 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(25, false)
KMX_BS_DEFINE_K(27, false)

}  // namespace kmx
