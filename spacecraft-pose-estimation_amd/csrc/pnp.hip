#include "common.h"
namespace scpose {
int32_t pnp_launch(const float*, const double*, const double*, const double*, int, int, float, int, float, int, int, double, double, double*, double*, double*, int32_t*, hipStream_t) {
  set_error("pnp: not built yet");
  return SCPOSE_E_INVALID;
}
}
