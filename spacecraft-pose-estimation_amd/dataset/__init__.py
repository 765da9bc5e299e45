"""dataset.<NAME>(cfg, root, image_dir, image_set, is_train, transform) -- the lookup of
landmark_regression/tools/test.py:109-115.  PEdataset / lightbox / sunlamp of the reference
differ from EventsDataset only in constructor signature: speedplus.py."""
from .events import EventsDataset  # noqa: F401
from .events import EventsDataset as events  # noqa: F401
from .speedplus import PEdataset, lightbox, sunlamp  # noqa: F401
