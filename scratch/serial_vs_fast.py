"""Debug aid: the serial accumulate kernel against the fast one on the same cloud (two processes: the knob is read once).
usage: python scratch/serial_vs_fast.py npoints  -> runs itself twice and compares the dumps"""
import sys, os, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np
if len(sys.argv) > 2:
    import cwipc_util_amd as cw
    from bench import make_input
    pc = make_input(cw, int(sys.argv[1]), 0.0)
    out = cw.cwipc_downsample(pc, 0.01).get_numpy_array()
    np.save(sys.argv[2], out)
    sys.exit(0)
n = sys.argv[1]
env = dict(os.environ)
env1 = dict(os.environ); env1["CWIPC_VOXEL_SERIAL"] = "1"; subprocess.check_call([sys.executable, __file__, n, "/tmp/ser.npy"], env=env1)
env["CWIPC_VOXEL_SERIAL"] = "0"; env0 = dict(os.environ); env0["CWIPC_VOXEL_SERIAL"] = "1"
subprocess.check_call([sys.executable, __file__, n, "/tmp/fast.npy"], env=env)
a, b = np.load("/tmp/ser.npy"), np.load("/tmp/fast.npy")
print("counts", len(a), len(b))
if len(a) == len(b):
    bad = np.nonzero((a['x'] != b['x']) | (a['y'] != b['y']) | (a['z'] != b['z']) | (a['r'] != b['r']) | (a['tile'] != b['tile']))[0]
    print("differing outputs:", len(bad))
    for i in bad[:12]:
        print(i, a[i], b[i])
