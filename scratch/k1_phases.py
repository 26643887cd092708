"""Phase time stamps of two workgroups of the fast accumulate kernel (debug-knob build only: CWIPC_LIBRARY_DIR=scratch/lib_dbg,
CWIPC_FAST_STAMPS=1).  Calls are waited for one by one (profiling mode), so the kernel runs alone."""
import sys, os
sys.path.insert(0, os.getcwd())
import cwipc_util_amd as cw
from bench import make_input
for npts in (300000, 10000000):
    pc = make_input(cw, npts, 0.0)
    cw.cwipc_hip_upload(pc, drop_host_copy=True)
    for _ in range(6): cw.cwipc_downsample(pc, 0.01).count()
    print("==== %d points" % npts, file=sys.stderr, flush=True)
    with cw.cwipc_hip_profile() as prof:
        for _ in range(3): cw.cwipc_downsample(pc, 0.01)
    print("   K1 by events: %.1f us" % (prof.kernels["voxel_accumulate"][0] / prof.kernels["voxel_accumulate"][1] * 1e3), file=sys.stderr, flush=True)
