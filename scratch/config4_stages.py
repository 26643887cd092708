import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cwipc_util_amd as cw
from cwipc_util_amd.capture import capture_tile
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
tiles = [capture_tile(2_000_000, t, 8, timestamp=1000 + t) for t in range(8)]
def frame(stages):
    outs = []
    for t, pc in enumerate(tiles):
        t0 = time.perf_counter(); f = cw.cwipc_tilefilter(pc, 1 << t)
        if stages is not None: sync(); stages['tilefilter'] += time.perf_counter() - t0
        t0 = time.perf_counter(); d = cw.cwipc_downsample(f, 0.01)
        if stages is not None: d.count(); sync(); stages['downsample'] += time.perf_counter() - t0
        outs.append(d)
    t0 = time.perf_counter(); j = cw.cwipc_join_multi(outs)
    if stages is not None: sync(); stages['join'] += time.perf_counter() - t0
    return j
for _ in range(5): frame(None)
sync()
t0 = time.perf_counter()
for _ in range(30): frame(None)
sync()
print('frame ms (no stage waits)', (time.perf_counter() - t0) / 30 * 1e3)
st = {'tilefilter': 0.0, 'downsample': 0.0, 'join': 0.0}
for _ in range(30): frame(st)
print({k: round(v / 30 / (8 if k != 'join' else 1) * 1e6, 1) for k, v in st.items()}, 'us per call')
with cw.cwipc_hip_profile() as prof:
    frame(None)
print({k: (round(v[0] * 1e3, 1), v[1]) for k, v in prof.kernels.items()})
