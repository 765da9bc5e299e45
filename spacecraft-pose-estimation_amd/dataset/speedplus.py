"""PEdataset / lightbox / sunlamp: the SPEED+ dataset classes of landmark_regression/lib/dataset/
(PEdataset.py:25-27, lightbox.py:25-27, sunlamp.py:25-27).  They read the same COCO dict and write the same
.mat as EventsDataset -- the three files differ from events.py only in the class name and constructor
signature -- so they are EventsDataset under those names and signatures.  The multi-scale training targets
(JointsDataset.py:205-208) belong to training, which is out of scope."""
from .events import EventsDataset


class PEdataset(EventsDataset):
    def __init__(self, cfg, root, image_dir, image_set, is_train, transform=None, numpy_transform=None,
                 multi_scale_target=False):
        if multi_scale_target:
            raise ValueError("PEdataset(multi_scale_target=True): multi-scale targets are a training feature")
        super().__init__(cfg, root, image_dir, image_set, is_train, transform, numpy_transform)


class lightbox(EventsDataset):
    def __init__(self, cfg, root, image_dir, image_set, is_train, transform=None):
        super().__init__(cfg, root, image_dir, image_set, is_train, transform)


class sunlamp(EventsDataset):
    def __init__(self, cfg, root, image_dir, image_set, is_train, transform=None):
        super().__init__(cfg, root, image_dir, image_set, is_train, transform)
