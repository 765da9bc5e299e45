"""models.hrnet_cms_384.get_pose_net -- the hrnet_cms_384 member of the reference's model family on the HIP engine.

landmark_regression/lib/models/hrnet_cms_384.py: the pose_hrnet trunk with multi_scale_output=True in the last
stage-4 module (:321-322) and four heads ConvTranspose2d(C_b -> 32, k3, stride 2, padding 1, output_padding 1)
+ Conv2d(32 -> NUM_JOINTS) whose outputs are summed coarse-to-fine through bilinear x2 upsampling
(:353-419, :551-562); eval returns the finest map, float32 N x J x H/2 x W/2.
state_dict keys: the trunk's, plus final_layer{,2,3,4}_4x.{0,1}.{weight,bias}.
The parameter tree and engine handling are pose_hrnet's; csrc/hrnet.cpp folds each head pair and csrc/head.hip
evaluates it.
"""
from . import pose_hrnet as _base


class PoseHighResolutionNet(_base.PoseHighResolutionNet):
    MODEL_NAME = "hrnet_cms_384"
    HEAD = ("4x", 3, 2)


def get_pose_net(cfg, is_train, **kwargs):
    return _base._get_pose_net(PoseHighResolutionNet, cfg, is_train, **kwargs)
