"""Generate tests/golden/colorize_vectors.npz from the REFERENCE's own colorize filter.

Runs only in the build container (needs /root/reference); the GPU box and the test
suite use the committed .npz.  The reference package cannot be imported whole here
(``import cwipc`` needs open3d, reference python/cwipc/util.py:24), so the single
file python/cwipc/filters/colorize.py is executed with tiny stand-ins for the two
names it imports from its siblings (the filter base class and the ctypes point
container helpers).  The code under test -- the colour maps (colorize.py:15-55), the
constructor (:68-83) and the per-point blend ``_mapcolor`` (:100-119) -- is the
reference's, unmodified.

Usage: python tests/golden/make_colorize_vectors.py
"""
import ctypes
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/python/cwipc/filters/colorize.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "colorize_vectors.npz")


class cwipc_point(ctypes.Structure):
    _fields_ = [("x", ctypes.c_float), ("y", ctypes.c_float), ("z", ctypes.c_float),
                ("r", ctypes.c_ubyte), ("g", ctypes.c_ubyte), ("b", ctypes.c_ubyte), ("tile", ctypes.c_ubyte)]


class FakeCloud:
    """Duck-typed cwipc_pointcloud_wrapper: just enough for _mapcolor."""

    def __init__(self, points, ts=0):
        self._points, self._ts, self._cellsize = points, ts, 0.0

    def get_points(self):
        return self._points

    def timestamp(self):
        return self._ts

    def cellsize(self):
        return self._cellsize

    def _set_cellsize(self, c):
        self._cellsize = c

    def count(self):
        return len(self._points)


def load_reference_colorize():
    pkg = types.ModuleType("cwipc")
    pkg.__path__ = []
    filters = types.ModuleType("cwipc.filters")
    filters.__path__ = []
    abstract = types.ModuleType("cwipc.filters.abstract")
    abstract.cwipc_abstract_filter = object
    util = types.ModuleType("cwipc.util")
    util.cwipc_pointcloud_wrapper = FakeCloud
    util.cwipc_point_array = lambda **kw: None
    util.cwipc_from_points = lambda points, ts: FakeCloud(points, ts)
    sys.modules.update({"cwipc": pkg, "cwipc.filters": filters, "cwipc.filters.abstract": abstract, "cwipc.util": util})
    spec = importlib.util.spec_from_file_location("cwipc.filters.colorize", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cwipc.filters.colorize"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_colorize()
    rng = np.random.default_rng(20260129)
    n = 4096
    dtype = np.dtype([('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('r', 'u1'), ('g', 'u1'), ('b', 'u1'), ('tile', 'u1')])
    pts = np.zeros(n, dtype=dtype)
    pts['x'], pts['y'], pts['z'] = rng.random(n), rng.random(n), rng.random(n)
    pts['r'], pts['g'], pts['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    tiles = rng.integers(0, 256, n)
    tiles[:512] = rng.choice([1, 2, 4, 8, 16, 32, 64, 128], 512)   # make sure the "camera" map is exercised
    tiles[512:520] = [0, 255, 254, 3, 7, 15, 31, 63]
    pts['tile'] = tiles
    # every old colour value at least once
    pts['r'][1024:1280] = np.arange(256)
    pts['g'][1024:1280] = np.arange(256)[::-1]
    pts['b'][1024:1280] = (np.arange(256) * 7) % 256

    cases = [("camera", 0.8), ("contributions", 0.8), ("camera", 1.0), ("camera", 0.0), ("contributions", 0.3),
             ((0.25, 0.5, 0.75), 0.6), ((1, 1, 1), 0.5)]
    out = {"input": pts}
    for idx, (cmap, weight) in enumerate(cases):
        flt = ref.ColorizeFilter(weight, cmap)
        arr = (cwipc_point * n).from_buffer_copy(pts.tobytes())
        res = flt._mapcolor(FakeCloud(arr, 42))
        got = np.frombuffer(bytes(res.get_points()), dtype=dtype).copy()
        out[f"case{idx}_output"] = got
        out[f"case{idx}_weight"] = np.float64(weight)
        out[f"case{idx}_cmap"] = np.array(cmap if isinstance(cmap, str) else "uniform")
        out[f"case{idx}_uniform"] = np.array(cmap if not isinstance(cmap, str) else (0, 0, 0), dtype=np.float64)
        # the reference's tables, for checking our restated colour maps
        lut = np.zeros((256, 3))
        valid = np.zeros(256, dtype=np.uint8)
        for t in range(256):
            c = flt.colorMap.map(t)
            if c is not None:
                lut[t] = c
                valid[t] = 1
        out[f"case{idx}_lut"] = lut
        out[f"case{idx}_valid"] = valid
    out["ncases"] = np.int64(len(cases))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
