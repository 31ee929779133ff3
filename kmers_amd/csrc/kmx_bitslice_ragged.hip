// kmx_bitslice_ragged.hip -- bit-sliced scan instantiations for ragged reads (kernel: kmx_bitslice_kernel.h)
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BSR_DEFINE_K(21)
KMX_BSR_DEFINE_K(31)

}  // namespace kmx
