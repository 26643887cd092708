"""The partition pass against the general kernel alone (CWIPC_VOXEL_PARTITION=0) on shuffled clouds of many sizes: one child
process per setting hashes six consecutive cwipc_downsample(+0.01 / -0.01) results per cloud (the pass switches itself on after the first)."""
import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [70001, 100000, 262144, 300000, 999999, 1048576 + 3, 2500000, 4194304, 5000001, 7777777, 10000000]
CHILD = r"""
import sys, json, hashlib, numpy as np
sys.path.insert(0, %r)
import cwipc_util_amd as cw
from bench import make_input
out = {}
for n in %r:
    base = make_input(cw, n, 0.0)
    pts = base.get_numpy_array().copy(); cs = base.cellsize(); base.free()
    perm = pts[np.random.default_rng(n).permutation(len(pts))]
    pc = cw.cwipc_from_numpy_array(np.ascontiguousarray(perm), 1); pc._set_cellsize(cs)
    cw.cwipc_hip_upload(pc, drop_host_copy=True)
    for c in (0.01, -0.01):
        hs = []
        for rep in range(6):
            a = cw.cwipc_downsample(pc, c).get_numpy_array()
            hs.append(hashlib.sha256(a.tobytes()).hexdigest()[:16] + ':%%d' %% len(a))
        out['%%d/%%s' %% (n, c)] = hs
    pc.free()
print(json.dumps(out))
""" % (root, SIZES)
res = {}
for name, env in (("partition", {}), ("general only", {"CWIPC_VOXEL_PARTITION": "0"})):
    p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=dict(os.environ, **env), timeout=1000)
    if p.returncode != 0:
        print(name, "FAILED", p.stderr[-3000:]); sys.exit(1)
    res[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(name, "done", flush=True)
bad = 0
for k in res["partition"]:
    a, b = res["partition"][k], res["general only"][k]
    same = len(set(a)) == 1 and a == b
    if not same: bad += 1
    print(k, "OK" if same else "DIFFERENT", a[0] if same else (a, b))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
