// kmx_bitslice_k33_39.hip -- bit-sliced [u64;2] scan instantiations for k = 33, 35, 37, 39 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS2_DEFINE_K(33)
KMX_BS2_DEFINE_K(35)
KMX_BS2_DEFINE_K(37)
KMX_BS2_DEFINE_K(39)

}  // namespace kmx
