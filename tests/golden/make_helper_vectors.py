"""Generate tests/golden/helper_vectors.npz from the REFERENCE's own Python: the tile synchroniser's policy loop and the
numpy-side helpers of SURVEY section 8f rows 2 and 3.

Runs only in the build container (needs /root/reference); the GPU box and the test suite use the committed .npz.  The
function bodies under test run UNMODIFIED from the files where they lie:

  * `_Synchronizer.run`                      /root/reference/python/cwipc/net/source_synchronizer.py:106-200
  * `cwipc_tilefilter_masked`                /root/reference/python/cwipc/registration/util.py:98-112
  * `cwipc_downsample_pertile`               /root/reference/python/cwipc/registration/util.py:170-182  (control flow)
  * `get_tiles_used`, `cwipc_transform`      /root/reference/python/cwipc/registration/util.py:285-309
  * `TransformFilter.filter`                 /root/reference/python/cwipc/filters/transform.py:32-49

What stands in, in memory only, and for what:
  * `open3d` (absent from this image, imported at module level by python/cwipc/util.py:24 and registration/util.py:10-11) and
    scipy's `RigidTransform` (absent from scipy 1.15, registration/util.py:15): placeholders that none of the functions above touch;
  * the native library behind the reference's `cwipc.util` wrapper: THIS repository's libcwipc_util.so, for the container calls
    only (cwipc_from_points / from_numpy_array / get_points / get_numpy_matrix: byte copies on the host, themselves pinned by the
    reference's own test cases, tests/test_reference_wrapper.py);
  * the three filters the helpers call -- `cwipc_join`, `cwipc_tilefilter`, `cwipc_downsample` need a GPU in this library and
    PCL in the reference's: `cwipc_join` = concatenation with ts = min, cellsize = min (reference src/cwipc_filters.cpp:403-414,
    pinned by test_join), the other two only RECORD their calls (the fixture pins the order of calls, not PCL's arithmetic);
  * the per-tile decoders that feed the synchroniser: scripted sources (a list of clouds, each available after a number of polls);
  * `cwipc/registration/__init__.py` and `cwipc/net/__init__.py` are not executed (the one file each is loaded by path).

Usage: python tests/golden/make_helper_vectors.py
"""
import importlib.machinery
import importlib.util
import json
import os
import queue
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "helper_vectors.npz")


def placeholder(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []

    class _Anything:
        def __init__(self, *a, **k): pass
        def __getattr__(self, n): return _Anything

    m.__getattr__ = lambda n: _Anything
    sys.modules[name] = m
    return m


def load_by_path(modname, path, package_path=None):
    """Execute one file of the reference as module `modname` without running its package's __init__."""
    pkgname = modname.rsplit(".", 1)[0]
    if pkgname not in sys.modules:
        pkg = types.ModuleType(pkgname)
        pkg.__path__ = [package_path or os.path.dirname(path)]
        pkg.__spec__ = importlib.machinery.ModuleSpec(pkgname, None, is_package=True)
        sys.modules[pkgname] = pkg
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    sys.dont_write_bytecode = True                      # nothing is written into the reference tree
    top = placeholder("open3d")
    for sub in ("geometry", "utility", "visualization", "io", "pipelines"):
        setattr(top, sub, placeholder("open3d." + sub))
    setattr(sys.modules["open3d.pipelines"], "registration", placeholder("open3d.pipelines.registration"))
    import scipy.spatial.transform as sst
    if not hasattr(sst, "RigidTransform"):
        sst.RigidTransform = type("RigidTransform", (), {})
    sys.path.insert(0, os.path.join(REF_ROOT, "python"))
    import cwipc.util
    cwipc.util.cwipc_util_dll_load(os.path.join(REPO, "cwipc_util_amd", "lib", "libcwipc_util.so"))
    import cwipc
    reg = load_by_path("cwipc.registration.util", os.path.join(REF_ROOT, "python/cwipc/registration/util.py"))
    sync = load_by_path("cwipc.net.source_synchronizer", os.path.join(REF_ROOT, "python/cwipc/net/source_synchronizer.py"))
    import cwipc.filters.transform as tf                 # (cwipc/filters/__init__.py is empty)
    return cwipc, reg, sync, tf


POINT_DTYPE = np.dtype([('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('r', 'u1'), ('g', 'u1'), ('b', 'u1'), ('tile', 'u1')])


def random_points(rng, n, tiles):
    p = np.zeros(n, dtype=POINT_DTYPE)
    p['x'] = (rng.random(n) * 4 - 2).astype(np.float32)
    p['y'] = (rng.random(n) * 2).astype(np.float32)
    p['z'] = (rng.random(n) * 4 - 2).astype(np.float32)
    p['r'], p['g'], p['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    p['tile'] = rng.choice(tiles, n)
    return p


# ---------------------------------------------------------------------------
# the synchroniser
# ---------------------------------------------------------------------------
class ScriptedSource:
    """Stands for a per-tile decoder: hands out `clouds` in order; `gates[k]` polls must pass before cloud k is available."""

    def __init__(self, clouds, gates):
        self.clouds, self.gates, self.next, self.polls = list(clouds), list(gates), 0, 0

    def free(self): pass
    def start(self): return True
    def stop(self): pass
    def eof(self): return self.next >= len(self.clouds)

    def available(self, wait=False):
        if self.eof():
            return False
        self.polls += 1
        return self.polls > self.gates[self.next]

    def get(self):
        pc = self.clouds[self.next]
        self.next += 1
        self.polls = 0
        return pc


class Reader:
    def start(self): return True
    def stop(self): pass


def make_script(rng, n_tile, n_frames):
    """Per tile a list of (timestamp, cellsize, npoints, gate): equal timestamps, gaps, tiles that skip frames, late tiles."""
    script = []
    for t in range(n_tile):
        ts, rows = int(rng.integers(0, 3)), []
        for _ in range(n_frames):
            ts += int(rng.integers(0, 3))
            rows.append([ts, float(rng.choice([0.001, 0.002, 0.004])), int(rng.integers(1, 6)), int(rng.integers(0, 4))])
            ts += int(rng.integers(0, 4) > 0)     # now and then a tile repeats a timestamp: its second cloud comes too late
        script.append(rows)
    return script


def run_synchronizer(cwipc, sync_mod, script, prefer_partial):
    """The reference's loop on scripted sources, on this thread.  Every point carries (tile, index of its cloud in the tile's script)
    in its y and z, so the produced cloud tells which inputs were combined and in which order."""
    joins = []

    def join2(a, b):   # reference src/cwipc_filters.cpp:388-418
        pa, pb = a.get_numpy_array(), b.get_numpy_array()
        out = cwipc.cwipc_from_numpy_array(np.concatenate([pa, pb]), min(a.timestamp(), b.timestamp()))
        out._set_cellsize(min(a.cellsize(), b.cellsize()))
        joins.append(1)
        return out

    sync_mod.cwipc_join = join2
    sources = []
    for t, rows in enumerate(script):
        clouds = []
        for k, (ts, cs, n, _gate) in enumerate(rows):
            p = np.zeros(n, dtype=POINT_DTYPE)
            p['x'] = np.arange(n)
            p['y'], p['z'], p['tile'] = t, k, 1 << t
            pc = cwipc.cwipc_from_numpy_array(p, ts)
            pc._set_cellsize(cs)
            clouds.append(pc)
        sources.append(ScriptedSource(clouds, [r[3] for r in rows]))
    s = sync_mod.cwipc_source_synchronizer(Reader(), sources)   # the reference's own factory (:278-281): _MQSynchronizer, whose run() is _Synchronizer's
    s.prefer_partial_over_unsynced = prefer_partial
    s.output_queue = queue.Queue()          # (unbounded: nobody consumes while the loop runs on this thread)
    s.running = True
    s.run()                                 # :106-200, until a source reports end of file
    produced = []
    while not s.output_queue.empty():
        pc = s.output_queue.get()
        if pc is None:
            continue
        arr = pc.get_numpy_array()
        parts = []
        for i in range(len(arr)):
            key = [int(arr['y'][i]), int(arr['z'][i])]
            if not parts or parts[-1] != key:
                parts.append(key)
        produced.append({"timestamp": int(pc.timestamp()), "cellsize": float(pc.cellsize()), "count": int(pc.count()), "parts": parts})
    stats = {"late": [int(v) for v in s.late_per_occurrence], "desync": [int(v) for v in s.desync_per_occurrence],
             "missing": [int(v) for v in s.missing_per_occurrence]}
    return produced, stats


# ---------------------------------------------------------------------------
def main():
    cwipc, reg, sync_mod, tf = load_reference()
    rng = np.random.default_rng(20260129)
    out = {}
    meta = {"synchronizer": [], "pertile": [], "notes": __doc__.split("Usage:")[0]}

    # ---- _Synchronizer.run ----
    for case in range(24):
        n_tile, n_frames = int(rng.integers(2, 5)), int(rng.integers(4, 9))
        script = make_script(rng, n_tile, n_frames)
        for prefer in (True, False):
            produced, stats = run_synchronizer(cwipc, sync_mod, script, prefer)
            meta["synchronizer"].append({"script": script, "prefer_partial_over_unsynced": prefer, "produced": produced, "stats": stats})

    # ---- cwipc_tilefilter_masked, get_tiles_used ----
    tile_sets = [[1, 2, 4, 8], [1, 2], [0, 1, 3, 255], [16, 32, 64, 128], [5, 6, 9, 10, 12], [7]]
    masks = [0, 1, 2, 3, 4, 8, 15, 128, 255]     # (256 and beyond: numpy 2 refuses the comparison with a uint8 array inside the reference function)
    for i, tiles in enumerate(tile_sets):
        pts = random_points(rng, 400, np.array(tiles, dtype=np.uint8))
        pc = cwipc.cwipc_from_numpy_array(pts, 100 + i)
        pc._set_cellsize(0.003 + 0.001 * i)
        out["masked%d_in" % i] = pts
        out["masked%d_tiles_used" % i] = np.array(reg.get_tiles_used(pc), dtype=np.int64)
        for m in masks:
            res = reg.cwipc_tilefilter_masked(pc, m)
            out["masked%d_mask%d_out" % (i, m)] = res.get_numpy_array()
            out["masked%d_mask%d_meta" % (i, m)] = np.array([res.timestamp(), res.cellsize()], dtype=np.float64)

    # ---- cwipc_transform ----
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    matrices = {
        "identity": np.eye(4),
        "rot_y_45": np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1.0]]),
        "translate": np.array([[1, 0, 0, 0.125], [0, 1, 0, -1.5], [0, 0, 1, 3.0000001], [0, 0, 0, 1.0]]),
        "general": np.concatenate([rng.normal(size=(3, 4)), [[0, 0, 0, 1.0]]]),
        "scale_shear": np.array([[1.7, 0.2, 0, 0.01], [0, 0.9, -0.3, 0], [0.05, 0, 1.1, -0.02], [0, 0, 0, 1.0]]),
    }
    pts = random_points(rng, 1000, np.array([1, 2, 4, 8], dtype=np.uint8))
    out["transform_in"] = pts
    for name, m in matrices.items():
        pc = cwipc.cwipc_from_numpy_array(pts, 7)
        pc._set_cellsize(0.004)
        res = reg.cwipc_transform(pc, m)
        out["transform_%s_matrix" % name] = m
        out["transform_%s_out" % name] = res.get_numpy_array()
        out["transform_%s_meta" % name] = np.array([res.timestamp(), res.cellsize()], dtype=np.float64)

    # ---- TransformFilter.filter ----
    pts = random_points(rng, 600, np.array([1, 2], dtype=np.uint8))
    out["offsetscale_in"] = pts
    for i, (x, y, z, scale) in enumerate([(0, 0, 0, 1), (0.1, -0.25, 3.0, 2.0), (-1e-3, 1e-3, 0.333333, 0.001), (7.5, 0.0, -7.5, 1.0 / 3.0)]):
        pc = cwipc.cwipc_from_numpy_array(pts, 9)
        pc._set_cellsize(0.005)
        res = tf.TransformFilter(x, y, z, scale).filter(pc)
        out["offsetscale%d_params" % i] = np.array([x, y, z, scale], dtype=np.float64)
        out["offsetscale%d_out" % i] = res.get_numpy_array()
        out["offsetscale%d_meta" % i] = np.array([res.timestamp(), res.cellsize()], dtype=np.float64)

    # ---- cwipc_downsample_pertile: which calls, in which order ----
    for i, tiles in enumerate([[1, 2, 4, 8], [8, 2], [3], [0, 5, 255], [128, 64, 32, 16, 8, 4, 2, 1]]):
        pts = random_points(rng, 400, np.array(tiles, dtype=np.uint8))
        pc = cwipc.cwipc_from_numpy_array(pts, 55)
        calls = []

        class Tagged:
            def __init__(self, tag): self.tag = tag

        reg.cwipc_tilefilter = lambda p, t: (calls.append(["tilefilter", int(t)]), Tagged(["tile", int(t)]))[1]
        reg.cwipc_downsample = lambda p, cs: (calls.append(["downsample", p.tag, float(cs)]), Tagged(["down", p.tag]))[1]
        reg.cwipc_join = lambda a, b: (calls.append(["join", a.tag, b.tag]), Tagged(["join", a.tag, b.tag]))[1]
        res = reg.cwipc_downsample_pertile(pc, 0.0125)
        meta["pertile"].append({"tiles_in_cloud": sorted(set(pts['tile'].tolist())), "calls": calls, "result": res.tag})

    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(meta["synchronizer"]), "synchroniser runs")


if __name__ == "__main__":
    main()
