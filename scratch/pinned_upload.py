"""Upload of eight 4.8 MB camera tiles from page-locked arrays: the de-interleave kernel reading the host buffer (default) against DMA + kernel
(CWIPC_PINNED_UPLOAD=dma), and from ordinary arrays."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from bench import make_input
base = make_input(cw, 300000, 0.0).get_numpy_array().copy()
pinned = []
for i in range(8):
    p = cw.cwipc_hip_pinned_points(len(base)); p[:] = base; pinned.append(p)
plain = [base.copy() for _ in range(8)]
big = make_input(cw, 10_000_000, 0.0).get_numpy_array().copy()
bigp = cw.cwipc_hip_pinned_points(len(big)); bigp[:] = big
def rate(arrays, reps=20):
    ts = []
    for r in range(reps + 3):
        t0 = time.perf_counter()
        keep = [cw.cwipc_from_numpy_array(a, 1) for a in arrays]
        for k in keep: cw.cwipc_hip_upload(k)
        dt = time.perf_counter() - t0
        if r >= 3: ts.append(dt)
        del keep
    return sum(a.nbytes for a in arrays) / float(np.median(ts)) / 1e9
print(os.environ.get("CWIPC_PINNED_UPLOAD", "kernel"), "8 x 4.8 MB page-locked: %.1f GB/s; ordinary arrays: %.1f GB/s; one 160 MB page-locked: %.1f GB/s; ordinary: %.1f GB/s" % (
    rate(pinned), rate(plain), rate([bigp], 5), rate([big], 5)))
