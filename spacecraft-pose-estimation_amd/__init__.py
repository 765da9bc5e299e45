"""MI355X-native HRNet landmark-heatmap -> PnP pose inference path.

Drop-in for one hot path of mohsij/spacecraft-pose-estimation: pose_hrnet forward,
heatmap decode and per-frame EPnP+RANSAC, as hand-written gfx950 HIP kernels behind a C
ABI (include/scpose.h, csrc/).  The directory name carries a hyphen, so import it through
the top-level alias module ``scpose`` (``import scpose``) or importlib.
"""
from . import _native  # noqa: F401  (does not load the .so until first use)

__all__ = ["_native"]
__version__ = "0.1.0"
