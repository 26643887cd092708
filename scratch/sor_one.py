import sys, os
sys.path.insert(0, os.getcwd())
import cwipc_util_amd as cw
from bench import make_input
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
pc = make_input(cw, npts, 0.0)
cw.cwipc_hip_upload(pc, drop_host_copy=True)
for _ in range(3): cw.cwipc_remove_outliers(pc, 16, 1.0, False).count()
