"""ctypes binding of the CPU oracle (oracle/cwipc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing in cwipc_util_amd/ may import this module.

All functions take and return numpy structured arrays with the reference's point
dtype (python/cwipc/util.py:291): x,y,z <f4 ; r,g,b,tile u1  (16 bytes).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

POINT_DTYPE = np.dtype([('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('r', 'u1'), ('g', 'u1'), ('b', 'u1'), ('tile', 'u1')])
assert POINT_DTYPE.itemsize == 16

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib: Optional[ctypes.CDLL] = None


def build(native: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Returns the path of the shared object."""
    target = "native" if native else "all"
    subprocess.run(["make", "-C", _HERE, target], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "liboracle_cwipc_native.so" if native else "liboracle_cwipc.so")


def load(native: bool = False) -> ctypes.CDLL:
    global _lib
    if _lib is not None and not native:
        return _lib
    path = os.path.join(_HERE, "liboracle_cwipc_native.so" if native else "liboracle_cwipc.so")
    src = os.path.join(_HERE, "cwipc_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        path = build(native)
    lib = ctypes.CDLL(path)
    P = ctypes.c_void_p
    lib.oracle_synthetic_count.argtypes = [ctypes.c_int]
    lib.oracle_synthetic_count.restype = ctypes.c_int
    lib.oracle_synthetic_cellsize.argtypes = [ctypes.c_int]
    lib.oracle_synthetic_cellsize.restype = ctypes.c_float
    lib.oracle_synthetic.argtypes = [ctypes.c_int, ctypes.c_float, P]
    lib.oracle_synthetic.restype = None
    lib.oracle_guess_cellsize.argtypes = [P, ctypes.c_size_t]
    lib.oracle_guess_cellsize.restype = ctypes.c_float
    lib.oracle_tilefilter.argtypes = [P, ctypes.c_size_t, ctypes.c_int, P]
    lib.oracle_tilefilter.restype = ctypes.c_size_t
    lib.oracle_tilemap.argtypes = [P, ctypes.c_size_t, P, P]
    lib.oracle_tilemap.restype = None
    lib.oracle_crop.argtypes = [P, ctypes.c_size_t, P, P]
    lib.oracle_crop.restype = ctypes.c_size_t
    lib.oracle_colormap.argtypes = [P, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, P]
    lib.oracle_colormap.restype = None
    lib.oracle_join.argtypes = [P, ctypes.c_size_t, P, ctypes.c_size_t, P]
    lib.oracle_join.restype = ctypes.c_size_t
    lib.oracle_colorize.argtypes = [P, ctypes.c_size_t, ctypes.c_double, P, P, P]
    lib.oracle_colorize.restype = None
    lib.oracle_downsample_voxelgrid.argtypes = [P, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, P, ctypes.c_size_t, P]
    lib.oracle_downsample_voxelgrid.restype = ctypes.c_long
    lib.oracle_downsample.argtypes = [P, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, P, ctypes.c_size_t, P, P, P]
    lib.oracle_downsample.restype = ctypes.c_long
    lib.oracle_downsample_audit.argtypes = [P, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, P, ctypes.c_size_t, P, P, P]
    lib.oracle_downsample_audit.restype = ctypes.c_long
    lib.oracle_remove_outliers.argtypes = [P, ctypes.c_size_t, ctypes.c_int, ctypes.c_float, ctypes.c_int, P, P, P]
    lib.oracle_remove_outliers.restype = ctypes.c_long
    lib.oracle_knn_mean_dist.argtypes = [P, ctypes.c_size_t, ctypes.c_int, P]
    lib.oracle_knn_mean_dist.restype = ctypes.c_int
    if not native:
        _lib = lib
    return lib


def _pts(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=POINT_DTYPE)
    return a


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def empty(n: int) -> np.ndarray:
    return np.zeros(n, dtype=POINT_DTYPE)


def synthetic(npoints: int = 0, angle: float = 0.0) -> Tuple[np.ndarray, float]:
    """Points and cellsize of cwipc_synthetic(fps, npoints).get() with m_angle = angle."""
    lib = load()
    n = lib.oracle_synthetic_count(npoints)
    out = empty(n)
    lib.oracle_synthetic(npoints, angle, _p(out))
    return out, float(lib.oracle_synthetic_cellsize(npoints))


def guess_cellsize(pts: np.ndarray) -> float:
    pts = _pts(pts)
    return float(load().oracle_guess_cellsize(_p(pts), len(pts)))


def tilefilter(pts: np.ndarray, tile: int) -> np.ndarray:
    pts = _pts(pts)
    out = empty(len(pts))
    m = load().oracle_tilefilter(_p(pts), len(pts), tile, _p(out))
    return out[:m].copy()


def tilemap(pts: np.ndarray, mapping) -> np.ndarray:
    pts = _pts(pts)
    m = np.ascontiguousarray(np.frombuffer(bytes(mapping), dtype=np.uint8))
    assert m.size == 256
    out = empty(len(pts))
    load().oracle_tilemap(_p(pts), len(pts), _p(m), _p(out))
    return out


def crop(pts: np.ndarray, bbox) -> np.ndarray:
    pts = _pts(pts)
    bb = np.ascontiguousarray(bbox, dtype=np.float32)
    assert bb.size == 6
    out = empty(len(pts))
    m = load().oracle_crop(_p(pts), len(pts), _p(bb), _p(out))
    return out[:m].copy()


def colormap(pts: np.ndarray, clear_bits: int, set_bits: int) -> np.ndarray:
    pts = _pts(pts)
    out = empty(len(pts))
    load().oracle_colormap(_p(pts), len(pts), clear_bits & 0xffffffff, set_bits & 0xffffffff, _p(out))
    return out


def join(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a, b = _pts(a), _pts(b)
    out = empty(len(a) + len(b))
    load().oracle_join(_p(a), len(a), _p(b), len(b), _p(out))
    return out


def colorize(pts: np.ndarray, weight: float, lut: np.ndarray, valid: np.ndarray) -> np.ndarray:
    """lut: (256,3) float64 colours, valid: (256,) bool -- see cwipc_util_amd.filters.colorize.ColorMap.tables()."""
    pts = _pts(pts)
    lut = np.ascontiguousarray(lut, dtype=np.float64).reshape(256, 3)
    valid = np.ascontiguousarray(valid, dtype=np.uint8).reshape(256)
    out = empty(len(pts))
    load().oracle_colorize(_p(pts), len(pts), float(weight), _p(lut), _p(valid), _p(out))
    return out


class OracleError(RuntimeError):
    pass


def downsample(pts: np.ndarray, pc_cellsize: float, cellsize: float, info: Optional[dict] = None) -> Tuple[np.ndarray, float]:
    """cwipc_downsample(pc, cellsize): returns (points, cellsize of result).  Raises OracleError where the reference returns NULL."""
    pts = _pts(pts)
    out = empty(max(len(pts), 1))
    ocs = ctypes.c_float(0)
    nl = ctypes.c_int(0)
    dp = ctypes.c_int(0)
    m = load().oracle_downsample(_p(pts), len(pts), pc_cellsize, cellsize, _p(out), len(out),
                                 ctypes.addressof(ocs), ctypes.addressof(nl), ctypes.addressof(dp))
    if m < 0:
        raise OracleError(f"downsample: reference returns NULL (code {m})")
    if info is not None:
        info['n_leaves'] = nl.value
        info['depth'] = dp.value
    return out[:m].copy(), float(ocs.value)


def downsample_audit(pts: np.ndarray, pc_cellsize: float, cellsize: float):
    """downsample() plus, per output, the float64 mean of its contributors (m x 3) and their number (m): how far pcl's fp32
    running sums and the HIP path each are from the exact mean is then a matter of arithmetic, not of trust."""
    pts = _pts(pts)
    out = empty(max(len(pts), 1))
    mean64 = np.zeros((max(len(pts), 1), 3), dtype=np.float64)
    count = np.zeros(max(len(pts), 1), dtype=np.uint32)
    ocs = ctypes.c_float(0)
    m = load().oracle_downsample_audit(_p(pts), len(pts), pc_cellsize, cellsize, _p(out), len(out), ctypes.addressof(ocs),
                                       mean64.ctypes.data, count.ctypes.data)
    if m < 0:
        raise OracleError(f"downsample: reference returns NULL (code {m})")
    return out[:m].copy(), float(ocs.value), mean64[:m].copy(), count[:m].copy()


def remove_outliers(pts: np.ndarray, k: int, stddev_mul: float, per_tile: bool,
                    want_stats: bool = False):
    pts = _pts(pts)
    n = len(pts)
    out = empty(max(2 * n, 1))
    md = np.zeros(max(n, 1), dtype=np.float32) if (want_stats and not per_tile) else None
    thr = ctypes.c_double(float('nan'))
    m = load().oracle_remove_outliers(_p(pts), n, k, stddev_mul, 1 if per_tile else 0, _p(out),
                                      _p(md) if md is not None else None, ctypes.addressof(thr))
    if m < 0:
        raise OracleError("remove_outliers failed")
    res = out[:m].copy()
    if want_stats:
        return res, (md[:n] if md is not None else None), float(thr.value)
    return res


def knn_mean_dist(pts: np.ndarray, k: int) -> np.ndarray:
    pts = _pts(pts)
    md = np.zeros(max(len(pts), 1), dtype=np.float32)
    rc = load().oracle_knn_mean_dist(_p(pts), len(pts), k, _p(md))
    if rc != 0:
        raise OracleError("knn_mean_dist failed")
    return md[:len(pts)]


# ---------------------------------------------------------------------------
# numpy-side helpers of the reference that are really per-point kernels
# ---------------------------------------------------------------------------
def transform(pts: np.ndarray, matrix4x4) -> np.ndarray:
    """reference python/cwipc/registration/util.py:295-309 (cwipc_transform): the same numpy operations,
    float32 columns, float64 rotation and translation, the result stored back as float32."""
    pts = _pts(pts)
    m = np.asarray(matrix4x4, dtype=np.float64)
    xyz = np.stack([pts['x'], pts['y'], pts['z']], axis=1)          # float32 N x 3, as get_numpy_matrix builds it
    rotmat, transvec = m[:3, :3], m[:3, 3].transpose()
    moved = (rotmat @ xyz.transpose()).transpose() + transvec         # float64
    out = pts.copy()
    out['x'], out['y'], out['z'] = moved[:, 0].astype(np.float32), moved[:, 1].astype(np.float32), moved[:, 2].astype(np.float32)
    return out


def offset_scale(pts: np.ndarray, x: float, y: float, z: float, scale: float) -> np.ndarray:
    """reference python/cwipc/filters/transform.py:38-52 (TransformFilter.filter): p.x = (p.x + x) * scale in
    Python floats (float64), stored into a c_float."""
    pts = _pts(pts)
    out = pts.copy()
    for f, o in (('x', x), ('y', y), ('z', z)):
        out[f] = ((pts[f].astype(np.float64) + float(o)) * float(scale)).astype(np.float32)
    return out


def tilefilter_masked(pts: np.ndarray, mask: int) -> np.ndarray:
    """reference python/cwipc/registration/util.py:98-112 (cwipc_tilefilter_masked): points whose tile ANDed with the mask is
    non-zero, in input order.  Pinned by tests/golden/helper_vectors.npz (outputs of the reference function itself)."""
    pts = _pts(pts)
    return pts[(pts['tile'] & np.uint8(mask & 0xff)) != 0].copy() if 0 <= mask <= 255 else pts[:0].copy()


def downsample_pertile_plan(tiles_in_cloud, cellsize: float, calls: list):
    """reference python/cwipc/registration/util.py:170-182 (cwipc_downsample_pertile), its calls only: for every tile number of
    get_tiles_used (ascending) tilefilter -> downsample, results folded by pairwise joins, left to right.  Appends the calls to
    `calls` in the form tests/golden/make_helper_vectors.py records them from the reference function; returns the fold's tag."""
    result = None
    for t in sorted(set(int(v) for v in tiles_in_cloud)):
        calls.append(["tilefilter", t])
        calls.append(["downsample", ["tile", t], float(cellsize)])
        down = ["down", ["tile", t]]
        if result is None:
            result = down
        else:
            calls.append(["join", result, down])
            result = ["join", result, down]
    return result


def tiles_used(pts: np.ndarray):
    """reference python/cwipc/registration/util.py:285-293 (get_tiles_used)."""
    return sorted(np.unique(_pts(pts)['tile']).tolist())


def simulatecams(matrix: np.ndarray, ncamera: int) -> np.ndarray:
    """SimulatecamsFilter with hard = True, restated (reference python/cwipc/filters/simulatecams.py:17-30, 40-58, 70): the tile
    of every row of an N x 7 float32 point matrix.  Camera c looks along (cos, 0, sin) of 2 pi c / ncamera; the centroid is
    numpy's float32 mean of the coordinates with y set to 0; a point's vector is its float32 position, y = 0, minus the centroid
    (float32 arithmetic), its dot products with the float64 camera vectors are taken one by one (numpy.dot of a float32 and a
    float64 vector: products and sum in float64, in index order), and the camera is the LAST index of numpy.argsort -- the
    highest dot product, ties going to the higher camera index of the sorted order.  Pure numpy, row by row as the reference
    does it: for small inputs."""
    m = np.asarray(matrix, dtype=np.float32)
    cams = np.zeros((ncamera, 3), dtype=float)
    for c in range(ncamera):
        a = 2 * np.pi * c / ncamera
        cams[c, 0], cams[c, 2] = np.cos(a), np.sin(a)
    centroid = np.mean(m[:, :3], axis=0)
    centroid[1] = 0.0
    tiles = np.zeros(len(m), dtype=np.uint8)
    for i in range(len(m)):
        v = np.array(m[i, :3])
        v[1] = 0.0
        v -= centroid
        dots = np.zeros(ncamera, dtype=float)
        for c in range(ncamera):
            dots[c] = np.dot(v, cams[c])
        tiles[i] = 1 << int(np.argsort(dots)[::-1][0])
    return tiles
