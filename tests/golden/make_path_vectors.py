"""Generate tests/golden/path_vectors.npz: small inputs and the outputs of THIS repository's CPU oracle for
every filter of the path (the reference's own heavy arithmetic lives in PCL, which cannot be built here: these
are regression vectors of the restatement, not reference outputs -- DESIGN.md section 4 says which parts are
pinned by the reference).  The test suite (CPU: the oracle reproduces them; GPU: the HIP path matches them)
uses the committed .npz.

Usage: python tests/golden/make_path_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "path_vectors.npz")


def main() -> None:
    full, cellsize = oracle.synthetic(20000, 0.0)
    rng = np.random.default_rng(20260129)
    # 600 points: every 29th point of the synthetic cloud (scan order kept) plus a shell of scattered outliers
    pts = full[::29][:560].copy()
    extra = oracle.empty(40)
    extra['x'] = rng.uniform(-0.6, 0.6, 40).astype(np.float32)
    extra['y'] = rng.uniform(0.0, 2.0, 40).astype(np.float32)
    extra['z'] = rng.uniform(-0.6, 0.6, 40).astype(np.float32)
    extra['r'], extra['g'], extra['b'] = rng.integers(0, 256, 40), rng.integers(0, 256, 40), rng.integers(0, 256, 40)
    extra['tile'] = rng.integers(1, 4, 40)
    pts = np.concatenate([pts, extra])
    out = {"input": pts, "cellsize": np.float32(cellsize)}
    for name, cell in (("down_p05", 0.05), ("down_p20", 0.2), ("down_m05", -0.05), ("down_m20", -0.2)):
        res, cs = oracle.downsample(pts, float(cellsize), cell)
        out[name] = res
        out[name + "_cellsize"] = np.float32(cs)
    out["knn8"] = oracle.knn_mean_dist(pts, 8)
    out["sor_k8_s1"] = oracle.remove_outliers(pts, 8, 1.0, False)
    out["sor_k8_s1_pertile"] = oracle.remove_outliers(pts, 8, 1.0, True)
    out["tilefilter_1"] = oracle.tilefilter(pts, 1)
    out["crop"] = oracle.crop(pts, [-0.1, 0.2, 0.5, 1.5, -0.3, 0.05])
    out["colormap"] = oracle.colormap(pts, 0x00ff00ff, 0x05000007)
    out["tilemap"] = oracle.tilemap(pts, bytes((i * 7 + 3) % 256 for i in range(256)))
    out["join"] = oracle.join(pts[:100], pts[300:])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
