"""Supplementary numbers for DESIGN.md: host-in/host-out rate of the 10M downsample (PCIe inclusive),
and device-resident rates of the other filters on the same cloud.  Not the bench contract."""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from bench import make_input
base = make_input(cw, 10_000_000, 0.0)
pts = base.get_numpy_array().copy(); cs = base.cellsize(); n = len(pts)
res = {}
def timeit(f, reps=5):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); cw.util.cwipc_util_dll_load().cwipc_hip_synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))
def e2e():
    pc = cw.cwipc_from_numpy_array(pts, 1); pc._set_cellsize(cs)
    out = cw.cwipc_downsample(pc, 0.01); a = out.get_numpy_array(); return len(a)
t = timeit(e2e); res['e2e_downsample_host_in_host_out'] = {'s': t, 'Mpoints_s': n / t / 1e6}
pc = cw.cwipc_from_numpy_array(pts, 1); pc._set_cellsize(cs); cw.cwipc_hip_upload(pc, drop_host_copy=True)
for name, f in [('downsample_octree(+0.01)', lambda: cw.cwipc_downsample(pc, 0.01)), ('downsample_grid(-0.01)', lambda: cw.cwipc_downsample(pc, -0.01)),
                ('tilefilter(1)', lambda: cw.cwipc_tilefilter(pc, 1)), ('colormap', lambda: cw.cwipc_colormap(pc, 0xff000000, 0x01000000)),
                ('tilemap', lambda: cw.cwipc_tilemap(pc, list(range(256)))), ('crop', lambda: cw.cwipc_crop(pc, [-0.1, 0.1, 0.0, 1.0, -1, 1])),
                ('join(pc,pc)', lambda: cw.cwipc_join(pc, pc))]:
    t = timeit(f, 10); res[name] = {'us': t * 1e6, 'Gpoints_s': n / t / 1e9}
rng = np.random.default_rng(20260129)
perm = pts[rng.permutation(n)]
pp = cw.cwipc_from_numpy_array(perm, 1); pp._set_cellsize(cs); cw.cwipc_hip_upload(pp, drop_host_copy=True)
for name, f in [('downsample_octree_permuted', lambda: cw.cwipc_downsample(pp, 0.01)), ('downsample_grid_permuted', lambda: cw.cwipc_downsample(pp, -0.01))]:
    t = timeit(f, 10); res[name] = {'us': t * 1e6, 'Gpoints_s': n / t / 1e9}
t = timeit(lambda: cw.cwipc_remove_outliers(pc, 16, 1.0, False), 3); res['remove_outliers(16,1.0) 10M'] = {'ms': t * 1e3, 'Mpoints_s': n / t / 1e6}
with cw.cwipc_hip_profile() as prof:
    cw.cwipc_remove_outliers(pc, 16, 1.0, False); cw.cwipc_downsample(pc, -0.01); cw.cwipc_downsample(pp, 0.01); cw.cwipc_tilefilter(pc, 1)
res['kernels_ms'] = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.kernels.items()}
print(json.dumps(res, indent=1))
