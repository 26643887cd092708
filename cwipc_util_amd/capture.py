"""The simulated multi-camera capture BASELINE configs 4 and 5 are stated on.

"8-tile synthetic capture" (SURVEY section 8d, configs 4 and 5): tile i is the synthetic cloud with
angle = i * pi/4, every point mapped to the camera mask 1 << i, rotated about Y by i * 45 degrees --
the way the reference builds its analysis captures: one camera mask per tile
(python/cwipc/scripts/cwipc_create_analysis_test.py:70-79: `tilemask = 1 << camnum`, `cwipc_tilefilter`,
`cwipc_transform` with a rotation about Y) and the per-tile chain of
python/cwipc/registration/util.py:170-182 (`cwipc_downsample_pertile`: tilefilter -> downsample -> join).

Everything here goes through the product's own C-ABI (synthetic source, tilemap, transform): the tiles are
device-resident clouds, no CPU arithmetic on points.
"""
from __future__ import annotations

import math
import struct
from typing import Callable, List, Optional, Sequence

import numpy

from . import util
from .abstract import cwipc_pointcloud_abstract, cwipc_source_abstract

__all__ = ["rotation_about_y", "capture_tile", "capture_tiles", "per_tile_chain", "TileSource"]


def rotation_about_y(angle: float) -> numpy.ndarray:
    """4x4 float64 matrix of a rotation about the Y axis (what scipy's Rotation.from_euler('y', angle) gives the
    reference, cwipc_create_analysis_test.py:84-88)."""
    c, s = math.cos(angle), math.sin(angle)
    m = numpy.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def _synthetic_cloud(npoints: int, angle: float) -> util.cwipc_pointcloud_wrapper:
    src = util.cwipc_synthetic(0, npoints)
    src.start()
    out = bytearray(4)
    if not src.auxiliary_operation("amd-fixangle", struct.pack("f", angle), out):
        raise util.CwipcError("synthetic source refused amd-fixangle")
    pc = src.get()
    src.stop()
    src.free()
    if pc is None:
        raise util.CwipcError("synthetic source produced no cloud")
    return pc


def capture_tile(npoints: int, tile: int, ntiles: int = 8, timestamp: Optional[int] = None) -> util.cwipc_pointcloud_wrapper:
    """Tile `tile` of the simulated capture: synthetic(npoints, angle = tile * 2 pi / ntiles), camera mask 1 << tile,
    rotated about Y by the same angle."""
    angle = tile * 2.0 * math.pi / ntiles
    pc = _synthetic_cloud(npoints, angle)
    pc = util.cwipc_tilemap(pc, bytes([1 << tile]) * 256)
    if tile:
        pc = util.cwipc_transform(pc, rotation_about_y(angle))
    if timestamp is not None:
        pc._set_timestamp(timestamp)
    return pc


def capture_tiles(npoints: int, ntiles: int = 8, timestamp: Optional[int] = None) -> List[util.cwipc_pointcloud_wrapper]:
    return [capture_tile(npoints, t, ntiles, timestamp) for t in range(ntiles)]


def per_tile_chain(pc: cwipc_pointcloud_abstract, tile: int, cellsize: float) -> cwipc_pointcloud_abstract:
    """What one rank does to its tile in config 4: tilefilter(1 << tile) -> downsample(cellsize)
    (registration/util.py:175-177)."""
    return util.cwipc_downsample(util.cwipc_tilefilter(pc, 1 << tile), cellsize)


class TileSource(cwipc_source_abstract):
    """One camera of a simulated capture as a cwipc_source: every get() hands out the tile's cloud with the next frame's
    timestamp, after an optional per-tile filter chain.  Feeds `net.source_synchronizer` the way the reference's per-tile
    decoders do (source_synchronizer.py:128-149).  With threaded=True the chain runs on a thread of the source's own, one
    frame ahead (the reference's decoders are threads too); every thread has its own streams and workspaces in the library."""

    def __init__(self, cloud: cwipc_pointcloud_abstract, nframes: int, filters: Sequence[Callable] = (), first_timestamp: int = 1,
                 timestamp_step: int = 33, threaded: bool = False):
        self.cloud = cloud
        self.left = nframes
        self.filters = list(filters)
        self.timestamp = first_timestamp
        self.step = timestamp_step
        self._queue = None
        self._thread = None
        if threaded:
            import queue
            import threading
            self._queue = queue.Queue(maxsize=2)
            self._thread = threading.Thread(target=self._produce_all, daemon=True, name="cwipc_util_amd.TileSource")
            self._thread.start()

    def _produce(self) -> cwipc_pointcloud_abstract:
        pc = self.cloud
        for f in self.filters:
            pc = f.filter(pc) if hasattr(f, "filter") else f(pc)
        if pc is self.cloud:
            pc = util.cwipc_tilefilter(pc, 0)   # a cloud of its own (shares the planes): timestamps differ per frame
        pc._set_timestamp(self.timestamp)
        self.timestamp += self.step
        return pc

    def _produce_all(self) -> None:
        import queue
        n = self.left
        for _ in range(n):
            pc = self._produce()
            while self.left > 0:
                try:
                    self._queue.put(pc, timeout=0.05)
                    break
                except queue.Full:
                    continue
            if self.left <= 0:
                return

    def free(self) -> None:
        self.left = 0
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        self.cloud = None

    def eof(self) -> bool:
        return self.left <= 0

    def available(self, wait: bool) -> bool:
        if self.left <= 0:
            return False
        if self._queue is None:
            return True
        if wait:
            while self._queue.empty() and self.left > 0:
                import time
                time.sleep(0.0002)
        return not self._queue.empty()

    def get(self) -> Optional[cwipc_pointcloud_abstract]:
        if self.left <= 0:
            return None
        pc = self._produce() if self._queue is None else self._queue.get()
        self.left -= 1
        return pc

    def seek(self, timestamp: int) -> bool:
        return False

    def statistics(self) -> None:
        for f in self.filters:
            if hasattr(f, "statistics"):
                f.statistics()
