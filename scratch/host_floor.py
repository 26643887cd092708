"""What a call of cwipc_downsample(+0.01) in a stream costs the HOST: the loop's own time (calls return before their kernels ran) against the
time to the last kernel, by cloud size.  If the 10 M stream's ~47 us per call is the host's, small clouds show the same figure."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from bench import make_input
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
for npts in (100000, 1000000, 2000000, 5000000, 10485760):
    pcs = []
    for i in range(3):
        pc = make_input(cw, npts, 0.0)
        cw.cwipc_hip_upload(pc, drop_host_copy=True)
        pcs.append(pc)
    for i in range(40): m = cw.cwipc_downsample(pcs[i % 3], 0.01).count()
    N = 200
    sync(); t0 = time.perf_counter()
    for i in range(N): cw.cwipc_downsample(pcs[i % 3], 0.01)
    t1 = time.perf_counter()
    sync(); t2 = time.perf_counter()
    print("%9d points -> %7d: loop returned after %.1f us per call, all done after %.1f us per call" % (npts, m, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
    del pcs
