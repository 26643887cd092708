"""Staggered ranges against equal ranges over many cloud sizes (the range arithmetic's edges: the last range, ranges of 24 steps, clouds
that are not whole steps): one child process per setting hashes cwipc_downsample(+0.01 / -0.01) of synthetic clouds of the listed sizes."""
import sys, os, subprocess, json, hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [1523757, 1600000, 1777777, 2000000, 2345678, 3000001, 3999999, 5000000, 6000000, 6350400, 7654321, 9000000, 10000000, 11111111, 13000000, 15728640 + 7, 20000000]
CHILD = r"""
import sys, json, hashlib, numpy as np
sys.path.insert(0, %r)
import cwipc_util_amd as cw
from bench import make_input
out = {}
for n in %r:
    pc = make_input(cw, n, 0.0)
    cw.cwipc_hip_upload(pc, drop_host_copy=True)
    for c in (0.01, -0.01):
        hs = set()
        for rep in range(3):
            a = cw.cwipc_downsample(pc, c).get_numpy_array()
            hs.add(hashlib.sha256(a.tobytes()).hexdigest()[:16] + ':%%d' %% len(a))
        out['%%d/%%s' %% (n, c)] = sorted(hs)
    pc.free()
print(json.dumps(out))
""" % (root, SIZES)
res = {}
for name, env in (("default", {}), ("equal", {"CWIPC_K1_STAGGER": "0"}), ("p40rev", {"CWIPC_K1_STAGGER": "40", "CWIPC_K1_STAGGER_REV": "1"}), ("general", {"CWIPC_VOXEL_GENERAL": "1"})):
    p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    if p.returncode != 0:
        print(name, "FAILED", p.stderr[-2000:]); sys.exit(1)
    res[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(name, "done", flush=True)
bad = 0
for k in res["default"]:
    vals = {name: tuple(res[name][k]) for name in res}
    same = len(set(vals.values())) == 1 and len(vals["default"]) == 1
    if not same: bad += 1
    print(k, "OK" if same else "DIFFERENT", vals["default"] if same else vals)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
