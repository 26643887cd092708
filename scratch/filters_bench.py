"""Device-resident rates of the exact filters on the 10 M cloud: a stream of calls (one wait at the end, as
bench.py times the downsample) and single calls each followed by a wait.  Not the bench contract."""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
from bench import make_input
base = make_input(cw, 10_000_000, 0.0)
pts = base.get_numpy_array().copy(); cs = base.cellsize(); n = len(pts)
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
pcs = []
for _ in range(4):   # 640 MB: more than the Infinity Cache
    pc = cw.cwipc_from_numpy_array(pts, 1); pc._set_cellsize(cs); cw.cwipc_hip_upload(pc, drop_host_copy=True); pcs.append(pc)
res = {}
def stream(f, reps=100):
    for i in range(40): f(pcs[i % 4])   # (a thread whose downsample calls come back to back takes a second and a third workspace: 0.3 GB each, allocated once)
    sync(); t0 = time.perf_counter()
    for i in range(reps): out = f(pcs[i % 4])   # the previous result is released while this one is being made
    sync(); return (time.perf_counter() - t0) / reps
def single(f, reps=20):
    ts = []
    for i in range(reps):
        sync(); t0 = time.perf_counter(); f(pcs[i % 4]); sync(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))
def counted(f, reps=20):   # a caller that asks for the count right after every call (the reference's VoxelizeFilter statistics do)
    ts = []
    for i in range(reps + 5):
        sync(); t0 = time.perf_counter(); f(pcs[i % 4]).count(); dt = time.perf_counter() - t0
        if i >= 5: ts.append(dt)
    return float(np.median(ts))
for name, f in [('tilefilter(1)', lambda pc: cw.cwipc_tilefilter(pc, 1)), ('crop', lambda pc: cw.cwipc_crop(pc, [-0.1, 0.1, 0.0, 1.0, -1, 1])),
                ('colormap', lambda pc: cw.cwipc_colormap(pc, 0xff000000, 0x01000000)), ('tilemap', lambda pc: cw.cwipc_tilemap(pc, list(range(256)))),
                ('join(pc,pc)', lambda pc: cw.cwipc_join(pc, pc)), ('downsample(+0.01)', lambda pc: cw.cwipc_downsample(pc, 0.01)),
                ('downsample(-0.01)', lambda pc: cw.cwipc_downsample(pc, -0.01))]:
    a, b, c = stream(f), single(f), counted(f)
    res[name] = {'stream_us': round(a * 1e6, 1), 'single_us': round(b * 1e6, 1), 'call_then_count_us': round(c * 1e6, 1), 'stream_Gpoints_s': round(n / a / 1e9, 1)}
print(json.dumps(res, indent=1))
