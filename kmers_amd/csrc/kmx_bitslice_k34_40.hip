// kmx_bitslice_k34_40.hip -- bit-sliced [u64;2] scan instantiations for k = 34, 36, 38, 40 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS2_DEFINE_K(34)
KMX_BS2_DEFINE_K(36)
KMX_BS2_DEFINE_K(38)
KMX_BS2_DEFINE_K(40)

}  // namespace kmx
