// kmx_bitslice_k49_55.hip -- bit-sliced [u64;2] scan instantiations for k = 49, 51, 53, 55 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS2_DEFINE_K(49)
KMX_BS2_DEFINE_K(51)
KMX_BS2_DEFINE_K(53)
KMX_BS2_DEFINE_K(55)

}  // namespace kmx
