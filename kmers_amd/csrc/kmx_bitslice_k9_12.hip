// kmx_bitslice_k9_12.hip -- bit-sliced scan instantiations for k = 9, 10, 11, 12 (kernel: kmx_bitslice_kernel.h).  Round 6: below k = 13
// the reduce ran on the word-domain scan at 0.48 of the roofline (profiles/r06_k_sweep.txt); nothing in the bit-sliced kernel
// needs k >= 13 -- three accumulator blocks hold every diagonal of a window of up to 17 bases.
#include "kmx_bitslice_kernel.h"

namespace kmx {

KMX_BS_DEFINE_K(9, true)
KMX_BS_DEFINE_K(10, true)
KMX_BS_DEFINE_K(11, true)
KMX_BS_DEFINE_K(12, true)

}  // namespace kmx
