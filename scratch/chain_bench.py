"""BASELINE config 5: per frame 8 x synthetic(300000) tiles -> colorize(0.8, camera) -> downsample(0.01)
-> remove_outliers(16, 1.0) per tile -> join.  Reports frames/s and per-stage time, host buffers in (PCIe inclusive)
and device-resident variants."""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from cwipc_util_amd.filters.colorize import ColorizeFilter
from bench import make_input
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
tiles = []
for i in range(8):
    pc = make_input(cw, 300000, 0.7 * i)
    a = pc.get_numpy_array().copy(); a['tile'] = 1 << i
    tiles.append((a, pc.cellsize()))
flt = ColorizeFilter(0.8, "camera")
def frame(resident=None):
    outs = []
    t = {}
    def lap(name, t0):
        sync(); t[name] = t.get(name, 0) + time.perf_counter() - t0
    for i, (a, cs) in enumerate(tiles):
        t0 = time.perf_counter()
        if resident is None:
            pc = cw.cwipc_from_numpy_array(a, 1000 + i); pc._set_cellsize(cs); cw.cwipc_hip_upload(pc)
        else:
            pc = resident[i]
        lap('upload', t0); t0 = time.perf_counter()
        pc = flt.filter(pc); lap('colorize', t0); t0 = time.perf_counter()
        pc = cw.cwipc_downsample(pc, 0.01); lap('downsample', t0); t0 = time.perf_counter()
        pc = cw.cwipc_remove_outliers(pc, 16, 1.0, False); lap('outliers', t0)
        outs.append(pc)
    t0 = time.perf_counter()
    fused = cw.cwipc_join_multi(outs); lap('join', t0); t0 = time.perf_counter()
    arr = fused.get_numpy_array(); lap('download', t0)
    return len(arr), t
res = {}
for mode in ('host_in', 'resident'):
    resident = None
    if mode == 'resident':
        resident = []
        for i, (a, cs) in enumerate(tiles):
            pc = cw.cwipc_from_numpy_array(a, 1000 + i); pc._set_cellsize(cs); cw.cwipc_hip_upload(pc, drop_host_copy=True); resident.append(pc)
    for _ in range(5): frame(resident)
    lat = []; acc = {}
    t_all = time.perf_counter()
    for _ in range(100):
        t0 = time.perf_counter(); n, t = frame(resident); lat.append(time.perf_counter() - t0)
        for k, v in t.items(): acc[k] = acc.get(k, 0) + v
    total = time.perf_counter() - t_all
    lat = np.array(lat) * 1e3
    res[mode] = {'fps': 100 / total, 'p50_ms': float(np.percentile(lat, 50)), 'p99_ms': float(np.percentile(lat, 99)), 'fused_points': n,
                 'stage_ms_per_frame': {k: v / 100 * 1e3 for k, v in acc.items()}}
# The same chain without the per-stage waits (they are there for the stage times above), tiles spread over T worker
# threads (every thread has its own streams and workspaces in the library): frames/s, device-resident input.
from concurrent.futures import ThreadPoolExecutor
resident = []
for i, (a, cs) in enumerate(tiles):
    pc = cw.cwipc_from_numpy_array(a, 1000 + i); pc._set_cellsize(cs); cw.cwipc_hip_upload(pc, drop_host_copy=True); resident.append(pc)
def one_tile(i):
    pc = flt.filter(resident[i])
    pc = cw.cwipc_downsample(pc, 0.01)
    return cw.cwipc_remove_outliers(pc, 16, 1.0, False)
for T in (1, 2, 4, 8):
    with ThreadPoolExecutor(max_workers=T) as pool:
        def frame_fast():
            outs = list(pool.map(one_tile, range(8))) if T > 1 else [one_tile(i) for i in range(8)]
            return cw.cwipc_join_multi(outs).get_numpy_array().shape[0]
        for _ in range(5): frame_fast()
        lat = []
        t_all = time.perf_counter()
        for _ in range(100):
            t0 = time.perf_counter(); n = frame_fast(); lat.append(time.perf_counter() - t0)
        total = time.perf_counter() - t_all
        lat = np.array(lat) * 1e3
        res['resident_no_stage_waits_%d_threads' % T] = {'fps': 100 / total, 'p50_ms': float(np.percentile(lat, 50)), 'p99_ms': float(np.percentile(lat, 99)), 'fused_points': n}
print(json.dumps(res, indent=1))
