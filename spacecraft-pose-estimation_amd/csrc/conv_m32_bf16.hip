// bf16 instantiations of the 32x32x16-MFMA convolution kernel (conv_m32_kernel.h).
#include <type_traits>

#include "conv_m32_kernel.h"

namespace scpose {
int32_t conv_m32_dispatch_bf16(int mr, int wm, int nr, int occ, const ConvLaunch& L, size_t lds, hipStream_t st) {
  return m32_dispatch<0>(mr, wm, nr, occ, L, lds, st);
}
}  // namespace scpose
