// f16 instantiations of the pipelined convolution kernel (conv_pipe_kernel.h).
#include <type_traits>

#include "conv_pipe_kernel.h"
#include "conv_stag_kernel.h"

namespace scpose {
int32_t conv_pipe_dispatch_f16(int ks, int stride, int mrep, int nrep, int nt, int occ, const ConvLaunch& L, size_t lds, hipStream_t st) {
  return pipe_dispatch<1>(ks, stride, mrep, nrep, nt, occ, L, lds, st);
}
int32_t conv_stag_dispatch_f16(int mrep, int nrep, const ConvLaunch& L, size_t lds, hipStream_t st) {
  return stag_dispatch<1>(mrep, nrep, L, lds, st);
}
}  // namespace scpose
