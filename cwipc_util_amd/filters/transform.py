"""transform filter (reference python/cwipc/filters/transform.py:6-60)."""
from .abstract import _TimedFilter
from ..util import cwipc_offset_scale, cwipc_pointcloud_wrapper


class TransformFilter(_TimedFilter):
    """
    transform - Adjust coordinate system of the point clouds.
        Arguments:
            x: offset to add to X coordinates
            y: offset to add to Y coordinates
            z: offset to add to Z coordinates
            scale: scale factor to apply (after the offsets)
    """
    filtername = "transform"

    def __init__(self, x: float, y: float, z: float, scale: float):
        super().__init__()
        self.x, self.y, self.z, self.scale = x, y, z, scale

    def filter(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        return self._run(pc, lambda p: cwipc_offset_scale(p, self.x, self.y, self.z, self.scale))


CustomFilter = TransformFilter
