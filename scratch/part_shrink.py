"""The permuted 10 M cloud through the partition pass, by kernel, for CWIPC_PART_SHRINK = 0, 1, 2 (set by the caller)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from bench import make_input
base = make_input(cw, 10_000_000, 0.0)
pts = base.get_numpy_array().copy(); cs = base.cellsize()
perm = pts[np.random.default_rng(7).permutation(len(pts))]
pc = cw.cwipc_from_numpy_array(np.ascontiguousarray(perm), 1); pc._set_cellsize(cs)
cw.cwipc_hip_upload(pc, drop_host_copy=True)
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
for _ in range(12): n_out = cw.cwipc_downsample(pc, 0.01).count()
sync(); t0 = time.perf_counter()
for _ in range(20): cw.cwipc_downsample(pc, 0.01).count()
sync(); dt = (time.perf_counter() - t0) / 20
with cw.cwipc_hip_profile() as prof:
    for _ in range(5): cw.cwipc_downsample(pc, 0.01)
print("PART_SHRINK=%s: %.1f us per call, %d out | %s" % (os.environ.get("CWIPC_PART_SHRINK", "0"), dt * 1e6, n_out,
      ", ".join("%s %.1f" % (k, v[0] / v[1] * 1e3) for k, v in sorted(prof.kernels.items(), key=lambda kv: -kv[1][0]))), flush=True)
