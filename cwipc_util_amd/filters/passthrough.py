"""The identity filter of the filter factory ('passthrough' in a FILTERDESC).

Same name, constructor and behaviour as reference python/cwipc/filters/passthrough.py:3-21 (the cloud
comes back untouched); written on top of this package's timing base so that, like the other filters here,
it also reports how many points went through and how long the (empty) call took.
"""
from ..util import cwipc_pointcloud_wrapper
from .abstract import _TimedFilter


class PassthroughFilter(_TimedFilter):
    """passthrough: hands every point cloud back as it came.  No arguments.  A placeholder for filter chains under test."""

    filtername = "passthrough"

    def filter(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        return self._run(pc, lambda same_cloud: same_cloud)

    def statistics(self) -> None:
        print(f"{self.filtername}: count={self.count}")   # the line the reference prints
        super().statistics()


CustomFilter = PassthroughFilter
