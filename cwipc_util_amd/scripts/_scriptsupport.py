"""The grab -> filter chain -> sink loop that drives the filter path in a live stream.

Counterpart of `SourceServer` in reference python/cwipc/scripts/_scriptsupport.py:282-399 (constructor
arguments, `run()`, `stop()`, `statistics()` and the statistics lines keep their meaning); only what the filter path
needs of it: no argparse, no cameras, no seeking UI.  BASELINE config 5 runs on it:

    sources  = [TileSource(tile_i, nframes, filters=[colorize, voxelize, remove_outliers]) ...]   # one per camera
    grabber  = cwipc_source_synchronizer(None, sources)                                            # one GPU join per frame
    server   = SourceServer(grabber, sink, Namespace(count=nframes, filter=[], verbose=False, ...))
    server.run()
"""
from __future__ import annotations

import time
from types import SimpleNamespace
from typing import Any, List, Optional, Sequence, Union

from .. import filters
from ..abstract import cwipc_pointcloud_abstract, cwipc_source_abstract

__all__ = ["SourceServer", "CountingSink", "server_args"]


def server_args(count: Optional[int] = None, filter: Sequence[str] = (), verbose: bool = False, inpoint: Optional[int] = None,
                outpoint: Optional[int] = None, cameraconfig: Optional[str] = None) -> SimpleNamespace:
    """The fields of the argparse namespace SourceServer reads (reference _scriptsupport.py:283-300, 418-454)."""
    return SimpleNamespace(count=count, filter=list(filter), verbose=verbose, inpoint=inpoint, outpoint=outpoint, cameraconfig=cameraconfig)


class CountingSink:
    """A sink that only looks at what it is fed (frames, points, when): the `viewer` end of a benchmark run."""

    def __init__(self, touch_points: bool = False):
        self.frames = 0
        self.points = 0
        self.stamps: List[float] = []
        self.touch_points = touch_points
        self.last: Optional[cwipc_pointcloud_abstract] = None

    def feed(self, pc: cwipc_pointcloud_abstract) -> None:
        self.frames += 1
        self.points += pc.count()
        if self.touch_points:
            pc.get_numpy_array()      # the frame leaves the GPU, as it would towards an encoder or a renderer
        self.stamps.append(time.time())
        self.last = pc


class SourceServer:
    """Pull clouds from `grabber`, run them through the filter chain, feed them to `viewer`."""

    def __init__(self, grabber: cwipc_source_abstract, viewer: Any, args: Any, owns_grabber: bool = True):
        self.grabber: Optional[cwipc_source_abstract] = grabber
        self.viewer = viewer
        self.verbose = getattr(args, "verbose", False)
        self.count = getattr(args, "count", None)
        self.inpoint = getattr(args, "inpoint", None)
        self.outpoint = getattr(args, "outpoint", None)
        self.fps: Optional[float] = None
        self.stopped = False
        self.owns_grabber = owns_grabber
        self.times_grab: List[float] = []
        self.pointcounts_grab: List[int] = []
        self.latency_grab: List[float] = []
        self.lastGrabTime: Optional[float] = None
        self.pc_filters = [filters.factory(desc) if isinstance(desc, str) else desc for desc in (getattr(args, "filter", None) or [])]
        if owns_grabber and hasattr(grabber, "start") and not grabber.start():
            print("grab: failed to start() grabber", flush=True)
            self.grabber = None
            self.stopped = True

    def stop(self) -> None:
        if self.stopped:
            return
        if self.grabber is not None and self.owns_grabber and hasattr(self.grabber, "stop"):
            self.grabber.stop()
        self.stopped = True

    def grab_pc(self) -> Optional[cwipc_pointcloud_abstract]:
        if self.lastGrabTime and self.fps:
            wait = self.lastGrabTime + 1 / self.fps - time.time()
            if wait > 0:
                time.sleep(wait)
        g = self.grabber
        if g is None or g.eof():
            return None
        if not g.available(True):
            if not g.eof():
                time.sleep(0.001)
            return None
        pc = g.get()
        self.lastGrabTime = time.time()
        return pc

    def run(self) -> None:
        assert self.grabber is not None
        if self.inpoint and not self.grabber.seek(self.inpoint):
            raise RuntimeError(f"grab: seek to timestamp {self.inpoint} failed")
        while not self.stopped and not self.grabber.eof():
            t0 = time.time()
            pc = self.grab_pc()
            if not pc:
                continue
            for f in self.pc_filters:
                pc = f.filter(pc)
            self.pointcounts_grab.append(pc.count())
            stamp = pc.timestamp()
            t1 = time.time()
            if self.viewer is not None:
                if self.inpoint and stamp < self.inpoint:
                    continue
                if self.outpoint and stamp > self.outpoint:
                    self.count = 0
                    self.stop()
                    continue
                self.viewer.feed(pc)
            self.latency_grab.append(time.time() - stamp / 1000.0)
            self.times_grab.append(t1 - t0)
            if self.count is not None:
                self.count -= 1
                if self.count <= 0:
                    break

    def statistics(self) -> None:
        self.print1stat('capture_duration', self.times_grab)
        self.print1stat('capture_pointcount', self.pointcounts_grab, isInt=True)
        self.print1stat('capture_latency', self.latency_grab)
        if self.grabber is not None and hasattr(self.grabber, "statistics"):
            self.grabber.statistics()
        for f in self.pc_filters:
            f.statistics()

    def print1stat(self, name: str, values: Sequence[Union[int, float]], isInt: bool = False) -> None:
        if not values:
            print(f'grab: {name}: count=0')
            return
        lo, hi, avg = min(values), max(values), sum(values) / len(values)
        if isInt:
            print(f'grab: {name}: count={len(values)}, average={avg:.3f}, min={lo:d}, max={hi:d}')
        else:
            print(f'grab: {name}: count={len(values)}, average={avg:.3f}, min={lo:.3f}, max={hi:.3f}')
