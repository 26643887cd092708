"""cwipc_downsample(+0.01) on mid-size clouds (config 5's 300 k-point tiles, config 4's 2 M): which accumulate kernel runs and what it costs."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from bench import make_input
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
for npts in (100000, 300000, 1000000, 2000000):
    pc = make_input(cw, npts, 0.0)
    cw.cwipc_hip_upload(pc, drop_host_copy=True)
    for _ in range(8): m = cw.cwipc_downsample(pc, 0.01).count()
    sync(); t0 = time.perf_counter()
    for _ in range(50): cw.cwipc_downsample(pc, 0.01)
    sync(); dt = (time.perf_counter() - t0) / 50
    with cw.cwipc_hip_profile() as prof:
        for _ in range(5): cw.cwipc_downsample(pc, 0.01)
    print(npts, "points ->", m, "stream %.1f us per call |" % (dt * 1e6), ", ".join("%s %.1f" % (k, v[0] / v[1] * 1e3) for k, v in sorted(prof.kernels.items(), key=lambda kv: -kv[1][0])), flush=True)
