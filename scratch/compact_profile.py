"""Per-kernel times of the compaction at 10 M points (hipEvents, every kernel alone)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from bench import make_input
base = make_input(cw, 10_000_000, 0.0)
pts = base.get_numpy_array().copy(); cs = base.cellsize()
pcs = []
for _ in range(4):
    pc = cw.cwipc_from_numpy_array(pts, 1); pc._set_cellsize(cs); cw.cwipc_hip_upload(pc, drop_host_copy=True); pcs.append(pc)
for name, f in [('tilefilter(1)', lambda pc: cw.cwipc_tilefilter(pc, 1)), ('crop', lambda pc: cw.cwipc_crop(pc, [-0.1, 0.1, 0.0, 1.0, -1, 1])),
                ('masked(3)', lambda pc: cw.cwipc_tilefilter_masked(pc, 3))]:
    for i in range(5): f(pcs[i % 4]).count()
    with cw.cwipc_hip_profile() as prof:
        for i in range(20): f(pcs[i % 4])
    print(name, {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.kernels.items()}, 'kept', f(pcs[0]).count())
