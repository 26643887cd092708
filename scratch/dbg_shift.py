import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as gpu
from oracle import oracle
pts, cs = oracle.synthetic(100000, 0.0)
rng = np.random.default_rng(5)
def run(p, cell, name):
    pc = gpu.cwipc_from_numpy_array(p, 1); pc._set_cellsize(cs)
    with gpu.cwipc_hip_profile() as prof:
        got = gpu.cwipc_downsample(pc, cell).get_numpy_array()
    exp, _ = oracle.downsample(p, cs, cell)
    ks = {k: v[1] for k, v in prof.kernels.items()}
    if len(got) != len(exp):
        print(name, "COUNT", len(got), len(exp), ks); return
    dx = np.abs(got['x'].astype(np.float64) - exp['x']); dy = np.abs(got['y'].astype(np.float64) - exp['y']); dz = np.abs(got['z'].astype(np.float64) - exp['z'])
    bad = (dx > 1e-5) | (dy > 1e-5) | (dz > 1e-5)
    print(name, "n", len(got), "bad", int(bad.sum()), "max", dx.max(), dy.max(), dz.max(), ks)
    if bad.any():
        i = np.flatnonzero(bad)[:5]
        print("  first bad idx", i, "got", got[i], "exp", exp[i])
        # same multiset?
        gs = np.sort(got, order=['x','y','z']); es = np.sort(exp, order=['x','y','z'])
        print("  sorted max dx", np.abs(gs['x']-es['x']).max(), np.abs(gs['y']-es['y']).max(), np.abs(gs['z']-es['z']).max())
for shift in ((10.0, -3.0, 7.5), (-25.0, 0.0, -0.125), (0.64, 0.64, 0.64)):
    p = pts.copy()
    p['x'] += np.float32(shift[0]); p['y'] += np.float32(shift[1]); p['z'] += np.float32(shift[2])
    run(p, 0.01, f"{shift} fwd")
    run(p[::-1].copy(), 0.01, f"{shift} rev")
    run(p[rng.permutation(len(p))], 0.02, f"{shift} perm")
