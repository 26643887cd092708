"""HIP API calls of a stream of cwipc_downsample(+0.01) calls on a 1 M-point cloud (host-bound: ~22 us per call): run under
rocprofv3 --hip-trace --stats."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import cwipc_util_amd as cw
from bench import make_input
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
pcs = []
for i in range(3):
    pc = make_input(cw, npts, 0.0); cw.cwipc_hip_upload(pc, drop_host_copy=True); pcs.append(pc)
for i in range(40): cw.cwipc_downsample(pcs[i % 3], 0.01).count()
N = 2000
sync(); t0 = time.perf_counter()
for i in range(N): cw.cwipc_downsample(pcs[i % 3], 0.01)
t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
print("%d points: loop %.1f us per call, done %.1f us per call" % (npts, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
