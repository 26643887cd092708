import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cwipc_util_amd as cw
from bench import make_input
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
res = {}
for npts in (300000, 10000000):
    pc = make_input(cw, npts, 0.0)
    cw.cwipc_hip_upload(pc, drop_host_copy=True)
    if npts == 300000:
        pc = cw.cwipc_downsample(pc, 0.01)   # the config-5 stage input: ~36 k voxel centroids
    n = pc.count()
    for _ in range(3): cw.cwipc_remove_outliers(pc, 16, 1.0, False)
    sync(); t = []
    for _ in range(5):
        t0 = time.perf_counter(); cw.cwipc_remove_outliers(pc, 16, 1.0, False); sync(); t.append(time.perf_counter() - t0)
    with cw.cwipc_hip_profile() as prof:
        cw.cwipc_remove_outliers(pc, 16, 1.0, False)
    res[n] = {'ms': float(np.median(t)) * 1e3, 'kernels_ms': {k: round(v[0], 4) for k, v in prof.kernels.items()}}
print(json.dumps(res, indent=1))
