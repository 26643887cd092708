"""Downsample timing on a depth-camera like cloud (wide scan lines) and on the permuted synthetic cloud."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from oracle import oracle
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
rng = np.random.default_rng(44)
w, h = 2560, 1440
u, v = np.meshgrid(np.arange(w), np.arange(h)); n = w * h
pts = oracle.empty(n)
pts['x'] = (u.ravel() * (3.0 / w) - 1.5 + rng.normal(0, 0.0005, n)).astype(np.float32)
pts['y'] = (2.0 - v.ravel() * (2.0 / h) + rng.normal(0, 0.0005, n)).astype(np.float32)
pts['z'] = (1.5 + 0.4 * np.sin(u.ravel() * 0.0025) + rng.normal(0, 0.002, n)).astype(np.float32)
pts['tile'] = 1
def bench(name, p, cell):
    pc = cw.cwipc_from_numpy_array(p, 1); pc._set_cellsize(0.0); cw.cwipc_hip_upload(pc, drop_host_copy=True)
    for _ in range(3): out = cw.cwipc_downsample(pc, cell)
    sync(); t0 = time.perf_counter()
    for _ in range(10): out = cw.cwipc_downsample(pc, cell)
    sync(); dt = (time.perf_counter() - t0) / 10
    with cw.cwipc_hip_profile() as prof:
        cw.cwipc_downsample(pc, cell)
    print(name, len(p), 'points ->', out.count(), ': %.1f us per call, %.2f Gpoints/s' % (dt * 1e6, len(p) / dt / 1e9), {k: round(v[0] * 1000, 1) for k, v in prof.kernels.items()})
bench('scanlines 3.7M, cell 0.01', pts, 0.01)
bench('scanlines 3.7M, cell 0.005', pts, 0.005)
bench('scanlines 3.7M, plain grid 0.01', pts, -0.01)
syn, cs = oracle.synthetic(10_000_000, 0.0)
perm = syn[np.random.default_rng(20260129).permutation(len(syn))]
bench('synthetic 10M permuted, cell 0.01', perm, 0.01)
bench('synthetic 10M permuted, plain grid', perm, -0.01)
