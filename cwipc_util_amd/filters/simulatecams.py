"""simulatecams filter (reference python/cwipc/filters/simulatecams.py:9-80)."""
from typing import Optional

import numpy

from .abstract import _TimedFilter
from ..util import cwipc_from_numpy_matrix, cwipc_hip_simulatecams, cwipc_pointcloud_wrapper


class SimulatecamsFilter(_TimedFilter):
    """
    simulatecams - Turn point cloud into multiple tiles by simulating cameras.
        Arguments:
            ncam: The number of cameras, spaced equidistantly on a circle around x=z=0.
            hard: If False or not specified, each point is assigned to the camera with the highest dot product
                  or the second highest dot product, with a probability proportional to the dot products.
                  If True, each point is assigned to the camera with the highest dot product.
            skew: If hard=False a skew > 1 will skew the distribution to the closest camera.
    """
    filtername = "simulatecams"

    def __init__(self, ncamera: int, hard: Optional[bool] = False, skew: Optional[float] = 1.0):
        super().__init__()
        self.ncamera = ncamera
        self.camera_vectors = numpy.zeros((ncamera, 3), dtype=float)
        for i in range(ncamera):
            angle = 2 * numpy.pi * i / ncamera
            self.camera_vectors[i, 0] = numpy.cos(angle)
            self.camera_vectors[i, 2] = numpy.sin(angle)
        self.hard = hard
        self.skew = skew

    def filter(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        return self._run(pc, self._assign)

    def _assign(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        # the centroid exactly as the reference forms it (:42-45): numpy's float32 mean over the rows of the N x 7 matrix
        point_matrix = pc.get_numpy_matrix()
        centroid = numpy.mean(point_matrix[:, :3], axis=0)
        centroid[1] = 0.0
        if self.hard:
            # the per-point loop (:47-58, :70) is one kernel: tile = 1 << camera with the largest dot product
            return cwipc_hip_simulatecams(pc, self.camera_vectors, centroid)
        # hard = False draws from numpy.random for every point (:60-69): the same rule, vectorised on the host -- same
        # distribution, not the same random stream as the reference's per-point calls
        flat = point_matrix[:, :3].copy()
        flat[:, 1] = 0.0
        flat -= centroid
        dots = flat.astype(float) @ self.camera_vectors.T
        order = numpy.argsort(dots, axis=1, kind="stable")[:, ::-1]
        first, second = order[:, 0], order[:, 1]
        rows = numpy.arange(len(flat))
        w0, w1 = dots[rows, first] ** self.skew, dots[rows, second] ** self.skew
        chance = numpy.random.uniform(-w0, w1)
        camera = numpy.where(chance < 0, first, second)
        point_matrix[:, 6] = (1 << camera).astype(numpy.float32)
        out = cwipc_from_numpy_matrix(point_matrix, pc.timestamp())
        out._set_cellsize(pc.cellsize())
        return out


CustomFilter = SimulatecamsFilter
