// bf16 instantiations of the producer/consumer 32x32x16-MFMA convolution kernel (conv_m32p_kernel.h).
#include <type_traits>

#include "conv_m32p_kernel.h"

namespace scpose {
int32_t conv_m32p_dispatch_bf16(int stride, int mr, int nr, int c16, int cw2, const ConvLaunch& L, size_t lds, hipStream_t st) {
  return m32p_dispatch<0>(stride, mr, nr, c16, cw2, L, lds, st);
}
}  // namespace scpose
