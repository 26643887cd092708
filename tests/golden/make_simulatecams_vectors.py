"""Generate tests/golden/simulatecams_vectors.npz from the REFERENCE's own SimulatecamsFilter (hard assignment).

Runs only in the build container (needs /root/reference); the GPU box and the test suite use the committed .npz.
The reference package cannot be imported whole here (``import cwipc`` needs open3d, reference python/cwipc/util.py:24),
so the single file python/cwipc/filters/simulatecams.py is executed with stand-ins for the names it imports from its
siblings: the filter base class, and a point cloud that is just its N x 7 float32 matrix (what get_numpy_matrix /
cwipc_from_numpy_matrix exchange, reference util.py:671-694, 1188-1201).  The code under test -- camera vectors
(simulatecams.py:21-28), centroid (:42-45) and the per-point loop (:47-58, :70) -- is the reference's, unmodified.

Usage: python tests/golden/make_simulatecams_vectors.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/python/cwipc/filters/simulatecams.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "simulatecams_vectors.npz")


class MatrixCloud:
    def __init__(self, matrix, ts=0):
        self.m, self.ts, self.cs = matrix, ts, 0.0

    def get_numpy_matrix(self):
        return self.m.copy()

    def timestamp(self):
        return self.ts

    def cellsize(self):
        return self.cs

    def _set_cellsize(self, c):
        self.cs = c


def load_reference():
    pkg = types.ModuleType("cwipc"); pkg.__path__ = []
    filters = types.ModuleType("cwipc.filters"); filters.__path__ = []
    abstract = types.ModuleType("cwipc.filters.abstract"); abstract.cwipc_abstract_filter = object
    util = types.ModuleType("cwipc.util")
    util.cwipc_pointcloud_wrapper = MatrixCloud
    util.cwipc_from_numpy_matrix = lambda m, ts: MatrixCloud(m, ts)
    sys.modules.update({"cwipc": pkg, "cwipc.filters": filters, "cwipc.filters.abstract": abstract, "cwipc.util": util})
    spec = importlib.util.spec_from_file_location("cwipc.filters.simulatecams", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cwipc.filters.simulatecams"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    rng = np.random.default_rng(20260129)
    out = {}
    cases = [("blob4", 4, 3000), ("blob8", 8, 3000), ("ring3", 3, 2000), ("ring8", 8, 2500), ("symmetric6", 6, 1500)]
    for name, ncam, n in cases:
        m = np.zeros((n, 7), np.float32)
        if name.startswith("blob"):
            m[:, 0:3] = rng.normal(0, 0.4, (n, 3)) + rng.normal(0, 2.0, 3)
        elif name.startswith("ring"):
            a = rng.random(n) * 2 * np.pi
            m[:, 0], m[:, 1], m[:, 2] = 0.3 * np.cos(a) + 0.7, rng.random(n) * 2, 0.3 * np.sin(a) - 0.2
        else:
            # points ON the bisectors between cameras and mirror pairs: exact ties of the dot products
            a = (rng.integers(0, 12, n) * (np.pi / 6)).astype(np.float64)
            r = rng.integers(1, 5, n) * 0.25
            m[:, 0], m[:, 1], m[:, 2] = r * np.cos(a), rng.random(n), r * np.sin(a)
        m[:, 3:6] = rng.integers(0, 256, (n, 3))
        m[:, 6] = rng.integers(0, 4, n)
        got = ref.SimulatecamsFilter(ncam, hard=True).filter(MatrixCloud(m, 77)).m
        assert (got[:, :6] == m[:, :6]).all()
        out[name + "_in"] = m
        out[name + "_ncam"] = np.int32(ncam)
        out[name + "_tile"] = got[:, 6].astype(np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("_tile")})


if __name__ == "__main__":
    main()
