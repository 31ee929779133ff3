// kmx_bitslice_k58_64.hip -- bit-sliced [u64;2] scan instantiations for k = 58, 60, 62, 64 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS2_DEFINE_K(58)
KMX_BS2_DEFINE_K(60)
KMX_BS2_DEFINE_K(62)
KMX_BS2_DEFINE_K(64)

}  // namespace kmx
