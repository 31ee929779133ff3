// kmx_bitslice_k57_61.hip -- bit-sliced [u64;2] scan instantiations for k = 57, 59, 61 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS2_DEFINE_K(57)
KMX_BS2_DEFINE_K(59)
KMX_BS2_DEFINE_K(61)

}  // namespace kmx
