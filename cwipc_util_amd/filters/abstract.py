"""Filter plugin interface (same as reference python/cwipc/filters/abstract.py:4-20)."""
import time
from abc import ABC, abstractmethod
from typing import List, Union

from ..util import cwipc_pointcloud_wrapper


class cwipc_abstract_filter(ABC):
    filtername = "abstract"

    @abstractmethod
    def filter(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        """Feed a point cloud to the filter. Returns the resulting point cloud."""

    def statistics(self) -> None:
        """Print statistics on the usage of the filter."""

    def set_keep_source(self) -> None:
        """Keep the source point cloud instead of freeing it after processing."""


class _TimedFilter(cwipc_abstract_filter):
    """Bookkeeping shared by the hot-path filters: per-call wall clock and point counts,
    printed in the reference's format (e.g. reference python/cwipc/filters/voxelize.py:39-59)."""

    def __init__(self) -> None:
        self.count = 0
        self.times: List[float] = []
        self.original_pointcounts: List[int] = []
        self.pointcounts: List[int] = []
        self.keep_source = False

    def set_keep_source(self) -> None:
        self.keep_source = True

    def _run(self, pc: cwipc_pointcloud_wrapper, fn) -> cwipc_pointcloud_wrapper:
        self.count += 1
        t1 = time.time()
        self.original_pointcounts.append(pc.count())
        out = fn(pc)
        self.times.append(time.time() - t1)
        self.pointcounts.append(out.count())
        return out

    def statistics(self) -> None:
        if self.times:
            self.print1stat('duration', self.times)
        if self.original_pointcounts:
            self.print1stat('original_pointcount', self.original_pointcounts, True)
        if self.pointcounts:
            self.print1stat('pointcount', self.pointcounts, True)

    def print1stat(self, name: str, values: Union[List[int], List[float]], isInt: bool = False) -> None:
        count = len(values)
        if count == 0:
            print(f'{self.filtername}: {name}: count=0')
            return
        fmt = '{}: {}: count={}, average={:.3f}, min={:d}, max={:d}' if isInt else '{}: {}: count={}, average={:.3f}, min={:.3f}, max={:.3f}'
        print(fmt.format(self.filtername, name, count, sum(values) / count, min(values), max(values)))
