"""models.<NAME>.get_pose_net(cfg, is_train) -- the lookup tools/test.py performs
(reference landmark_regression/tools/test.py:84-86)."""
from . import hrnet_cms, hrnet_cms_384, pose_hrnet  # noqa: F401
