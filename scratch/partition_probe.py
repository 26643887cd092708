"""What would the accumulate kernel cost on a permuted cloud AFTER a bucket partition?  The partition is done on the host here
(numpy), only to time the kernels on the order it would produce -- a design probe for the device-side partition pass."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cwipc_util_amd as cw
from bench import make_input

cw.cwipc_hip_set_device(0)
base = make_input(cw, 10_000_000, 0.0)
pts = base.get_numpy_array().copy()
cs = base.cellsize()
n = len(pts)
rng = np.random.default_rng(7)
perm = pts[rng.permutation(n)]
inv = np.float32(1.0) / np.float32(0.01)


def bucket_order(p, shift, nb):
    v = [np.floor(p[f] * inv).astype(np.int64) >> shift for f in "xyz"]
    h = (v[0] * 73856093) ^ (v[1] * 19349663) ^ (v[2] * 83492791)
    b = (h ^ (h >> 15)) % nb
    return np.argsort(b, kind="stable")


orders = {
    "original": pts,
    "permuted": perm,
    "hash256 of 16^3": perm[bucket_order(perm, 4, 256)],
    "hash1024 of 16^3": perm[bucket_order(perm, 4, 1024)],
    "hash256 of 8^3": perm[bucket_order(perm, 3, 256)],
    "hash4096 of 8^3": perm[bucket_order(perm, 3, 4096)],
}
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
want = None
for name, p in orders.items():
    pc = cw.cwipc_from_numpy_array(np.ascontiguousarray(p), 1)
    pc._set_cellsize(cs)
    cw.cwipc_hip_upload(pc, drop_host_copy=True)
    for _ in range(4):
        out = cw.cwipc_downsample(pc, 0.01)
        cnt = out.count()
    sync()
    t0 = time.perf_counter()
    for _ in range(10):
        cw.cwipc_downsample(pc, 0.01).count()
    sync()
    dt = (time.perf_counter() - t0) / 10
    with cw.cwipc_hip_profile() as prof:
        for _ in range(5):
            cw.cwipc_downsample(pc, 0.01)
    ks = ", ".join("%s %.1f us x%d" % (k, v[0] / v[1] * 1e3, v[1] // 5) for k, v in sorted(prof.kernels.items(), key=lambda kv: -kv[1][0]))
    got = np.sort(out.get_numpy_array(), order=["z", "y", "x"])
    if want is None:
        want = got
    same = len(got) == len(want) and bool((got == want).all()) if name != "original" else True
    print("%-18s %8.1f us per call, %d points out, same result as original order: %s | %s" % (name, dt * 1e6, cnt, same, ks), flush=True)
    pc.free()

# a stream of calls on the permuted cloud, one by one: which kernels ran
pc = cw.cwipc_from_numpy_array(np.ascontiguousarray(perm), 1)
pc._set_cellsize(cs)
cw.cwipc_hip_upload(pc, drop_host_copy=True)
for call in range(24):
    if call % 3 == 2:
        with cw.cwipc_hip_profile() as prof:
            cw.cwipc_downsample(pc, 0.01).count()
        print("call %2d (profiled):" % call, ", ".join("%s %.1f" % (k, v[0] * 1e3) for k, v in sorted(prof.kernels.items(), key=lambda kv: -kv[1][0])), flush=True)
    else:
        sync(); t0 = time.perf_counter()
        cw.cwipc_downsample(pc, 0.01).count()
        sync(); print("call %2d: %.1f us" % (call, (time.perf_counter() - t0) * 1e6), flush=True)
