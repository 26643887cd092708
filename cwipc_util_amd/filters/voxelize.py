"""voxelize filter (reference python/cwipc/filters/voxelize.py:6-37)."""
from .abstract import _TimedFilter
from ..util import cwipc_downsample, cwipc_pointcloud_wrapper


class VoxelizeFilter(_TimedFilter):
    """
    voxelize - Reduce number of points by voxelization (combining points within a cube by their average)
        Arguments:
            vsize: a cube of vsize*vsize*vsize is used (float)
    """
    filtername = "voxelize"

    def __init__(self, vsize: float):
        super().__init__()
        self.vsize = vsize

    def filter(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        return self._run(pc, lambda p: cwipc_downsample(p, self.vsize))


CustomFilter = VoxelizeFilter
