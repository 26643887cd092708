"""crop filter (reference python/cwipc/filters/crop.py:6-41)."""
from .abstract import _TimedFilter
from ..util import cwipc_crop, cwipc_pointcloud_wrapper


class CropFilter(_TimedFilter):
    """
    crop - Remove points outside a given bounding box
        Arguments: minx, maxx, miny, maxy, minz, maxz
    """
    filtername = "crop"

    def __init__(self, minx: float, maxx: float, miny: float, maxy: float, minz: float, maxz: float):
        super().__init__()
        self.bounding_box = (minx, maxx, miny, maxy, minz, maxz)

    def filter(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        return self._run(pc, lambda p: cwipc_crop(p, self.bounding_box))


CustomFilter = CropFilter
