"""One-off: 100 M points (1.6 GB) through cwipc_downsample (+/-0.01) and cwipc_tilefilter against the oracle
(more workgroups than compute units, 32-bit index arithmetic near its limits)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from oracle import oracle
from bench import make_input
t0 = time.time()
pc = make_input(cw, 100_000_000, 0.0)
pts = pc.get_numpy_array(); cs = pc.cellsize(); n = len(pts)
print("points", n, "generated in %.1f s" % (time.time() - t0), flush=True)
for cell in (0.01, -0.01):
    t0 = time.time(); got = cw.cwipc_downsample(pc, cell).get_numpy_array(); t1 = time.time()
    exp, _ = oracle.downsample(pts, cs, cell); t2 = time.time()
    ok = len(got) == len(exp) and (got['tile'] == exp['tile']).all() and (got['r'] == exp['r']).all() and (got['g'] == exp['g']).all()
    dx = float(np.abs(got['x'].astype(np.float64) - exp['x']).max()) if len(got) == len(exp) else -1
    print("cell", cell, "gpu %d oracle %d same-set %s max|dx| %.3g (gpu %.2f s incl. upload, oracle %.2f s)" % (len(got), len(exp), ok, dx, t1 - t0, t2 - t1), flush=True)
got = cw.cwipc_tilefilter(pc, 1).get_numpy_array()
exp = oracle.tilefilter(pts, 1)
print("tilefilter", len(got), len(exp), got.tobytes() == exp.tobytes())
