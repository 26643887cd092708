"""remove_outliers filter (reference python/cwipc/filters/remove_outliers.py:6-43)."""
from .abstract import _TimedFilter
from ..util import cwipc_remove_outliers, cwipc_pointcloud_wrapper


class RemoveOutliersFilter(_TimedFilter):
    """
    remove_outliers - Remove outlier points by applying a statistical method on every point.
        Arguments:
            kNeighbours : How many neighbour points to take into account (int)
            threshold: threshold standard deviation multiplier (float)
            perTile: If true run the algorithm per tile (default: over the whole pointcloud)
    """
    filtername = "remove_outliers"

    def __init__(self, kNeighbours: int, threshold: float, perTile: bool = False):
        super().__init__()
        self.kNeighbours = kNeighbours
        self.threshold = threshold
        self.perTile = perTile

    def filter(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        return self._run(pc, lambda p: cwipc_remove_outliers(p, self.kNeighbours, self.threshold, self.perTile))


CustomFilter = RemoveOutliersFilter
