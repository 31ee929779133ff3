// kmx_bitslice_k41_47.hip -- bit-sliced [u64;2] scan instantiations for k = 41, 43, 45, 47 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS2_DEFINE_K(41)
KMX_BS2_DEFINE_K(43)
KMX_BS2_DEFINE_K(45)
KMX_BS2_DEFINE_K(47)

}  // namespace kmx
