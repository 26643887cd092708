"""Does RCCL let two ranks share one GPU here?  (If it does, the library's exchange can be run with a real peer on a one-GPU
box; if it refuses -- "duplicate GPU" -- the N > 1 wire stays unmeasured until a multi-GPU node runs bench.py.)
Two child processes, both on device 0, id handed over through a file.  Run under `timeout`."""
import os, subprocess, sys, tempfile, time

CHILD = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cwipc_util_amd as cw
rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cw.cwipc_hip_set_device(0)
if rank == 0:
    uid = cw.cwipc_hip_comm_unique_id()
    open(path + ".tmp", "wb").write(uid); os.rename(path + ".tmp", path)
else:
    while not os.path.exists(path): time.sleep(0.01)
    uid = open(path, "rb").read()
comm = cw.cwipc_hip_comm(uid, rank, world)
print("rank", rank, "communicator up", flush=True)
n = 1000 * (rank + 1) + 3
pts = np.zeros(n, dtype=cw.cwipc_point_numpy_dtype)
pts['x'] = np.arange(n) + 10000 * rank; pts['y'] = rank; pts['tile'] = 1 << rank
pc = cw.cwipc_from_numpy_array(pts, 100 - rank); pc._set_cellsize(0.5 + rank)
for frame in range(5):
    out = comm.join(pc if (frame != 2 or rank != 1) else None)
    got = out.get_numpy_array()
    print("rank", rank, "frame", frame, "fused", len(got), "ts", out.timestamp(), "cellsize", out.cellsize(), "tiles", sorted(set(got['tile'].tolist())),
          "x ok", bool((np.diff(got['x']) > 0).all()), flush=True)
comm.free()
'''

world = 2
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "id")
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, str(r), str(world), path]) for r in range(world)]
    t_end = time.time() + 90
    for p in procs:
        try:
            p.wait(timeout=max(1, t_end - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
    print("exit codes", [p.returncode for p in procs])
