import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from bench import make_input
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
for npts in (300000, 2000000):
    pc = make_input(cw, npts, 0.0)
    pc = cw.cwipc_downsample(pc, 0.01) if npts == 300000 else pc
    n = pc.count()
    for _ in range(3): cw.cwipc_remove_outliers(pc, 16, 1.0, False)
    sync(); t0 = time.perf_counter()
    for _ in range(20): cw.cwipc_remove_outliers(pc, 16, 1.0, False)
    sync(); dt = (time.perf_counter() - t0) / 20
    with cw.cwipc_hip_profile() as prof:
        for _ in range(10): cw.cwipc_remove_outliers(pc, 16, 1.0, False)
    print(n, 'points: %.1f us per call' % (dt * 1e6), {k: round(v[0] / v[1] * 1000, 1) for k, v in prof.kernels.items()}, 'sum %.1f' % sum(v[0] / 10 * 1000 for v in prof.kernels.values()))
