// kmx_bitslice_k28_30.hip -- bit-sliced scan instantiations for k = 28, 29, 30 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(28, true)
KMX_BS_DEFINE_K(29, true)
KMX_BS_DEFINE_K(30, true)

}  // namespace kmx
