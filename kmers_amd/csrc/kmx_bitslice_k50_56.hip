// kmx_bitslice_k50_56.hip -- bit-sliced [u64;2] scan instantiations for k = 50, 52, 54, 56 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS2_DEFINE_K(50)
KMX_BS2_DEFINE_K(52)
KMX_BS2_DEFINE_K(54)
KMX_BS2_DEFINE_K(56)

}  // namespace kmx
