"""Device-side data path.  The KITTI / Matterport file readers themselves (decode, resize, connected
components of the depth mask) stay host-side dataset plumbing and are out of this build's scope."""
from .device_path import AugParams, DeviceBatchAssembler, DeviceLoader, SyntheticSampleSource, draw_augmentation  # noqa: F401
