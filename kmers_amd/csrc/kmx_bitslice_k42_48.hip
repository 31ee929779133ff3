// kmx_bitslice_k42_48.hip -- bit-sliced [u64;2] scan instantiations for k = 42, 44, 46, 48 (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS2_DEFINE_K(42)
KMX_BS2_DEFINE_K(44)
KMX_BS2_DEFINE_K(46)
KMX_BS2_DEFINE_K(48)

}  // namespace kmx
