"""Pin the CPU oracle against everything the reference's own tests hold for this path
(reference python/test_cwipc_util.py) and against vectors produced by the reference's
own Python code (tests/golden/colorize_vectors.npz).  CPU only.

What the reference pins for downsample / remove_outliers are count invariants only
(:528-594) -- their values live in PCL, which is absent, so those two stay
"parity unpinned" (DESIGN.md); the invariants are checked here all the same.
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_point_layout(oracle):
    assert oracle.POINT_DTYPE.itemsize == 16   # reference api.h:88-96


def test_synthetic_shape(oracle):
    # reference cwipc_synthetic.cpp:41-47,131: default 160000 points, cellsize 2/hsteps
    pts, cs = oracle.synthetic()
    assert len(pts) == 160000 and cs == pytest.approx(2.0 / 400)
    pts, cs = oracle.synthetic(1000)
    assert len(pts) == 31 * 31
    # SURVEY section 8 size table
    for npoints, n, t1, t2 in [(100000, 99856, 49928, 49928), (300000, 299209, 149878, 149331)]:
        pts, _ = oracle.synthetic(npoints)
        assert len(pts) == n
        assert (pts['tile'] == 1).sum() == t1 and (pts['tile'] == 2).sum() == t2
    # first and last point differ (reference test _verify_pointcloud, :663-672)
    assert tuple(pts[0])[:3] != tuple(pts[-1])[:3]


def test_cellsize_heuristic(oracle):
    # reference test_cwipc_timestamp_cellsize (:182-193): 4 unit-spaced points -> 1.0
    pts = oracle.empty(4)
    pts['x'] = [0, 1, 2, 3]
    assert oracle.guess_cellsize(pts) == 1.0
    assert oracle.guess_cellsize(pts[:1]) == 0.0


def test_tilefilter(oracle, synth):
    # reference test_tilefilter (:428-443), test_tilefilter_empty (:445-450)
    pts, _ = synth(0)
    assert len(oracle.tilefilter(pts, 0)) == len(pts)
    t1, t2 = oracle.tilefilter(pts, 1), oracle.tilefilter(pts, 2)
    assert len(t1) + len(t2) == len(pts)
    assert (t1['tile'] == 1).all() and (t2['tile'] == 2).all()
    assert len(oracle.tilefilter(pts, 256 + 1)) == 0   # int compare against a u8: nothing matches
    assert len(oracle.tilefilter(oracle.empty(0), 0)) == 0
    # BASELINE config 1
    pts, _ = synth(100000)
    assert len(oracle.tilefilter(pts, 1)) == 49928


def test_join(oracle, synth):
    # reference test_join (:452-464)
    a, _ = synth(0)
    b, _ = synth(1000)
    j = oracle.join(a, b)
    assert len(j) == len(a) + len(b)
    assert j[:len(a)].tobytes() == a.tobytes() and j[len(a):].tobytes() == b.tobytes()


def test_tilemap(oracle, synth):
    # reference test_tilemap (:466-487)
    pts, _ = synth(0)
    m = [0] * 256
    m[1], m[2] = 5, 6
    mapped = oracle.tilemap(pts, bytes(m))
    for a, b in ((1, 5), (2, 6), (5, 1), (6, 2)):
        assert len(oracle.tilefilter(pts, a)) == len(oracle.tilefilter(mapped, b))


def test_colormap(oracle, synth):
    # reference test_colormap (:489-506): clear everything, set 0x010203 -> (r,g,b,tile) == (1,2,3,0)
    pts, _ = synth(0)
    out = oracle.colormap(pts, 0xffffffff, 0x010203)
    assert len(out) == len(pts)
    assert (out['x'] == pts['x']).all() and (out['y'] == pts['y']).all() and (out['z'] == pts['z']).all()
    assert (out['r'] == 1).all() and (out['g'] == 2).all() and (out['b'] == 3).all() and (out['tile'] == 0).all()
    # bits 24-31 are the tile
    out = oracle.colormap(pts, 0xff000000, 0x07000000)
    assert (out['tile'] == 7).all() and (out['r'] == pts['r']).all()


def test_crop(oracle, synth):
    # reference test_crop (:508-526)
    pts, _ = synth(0)
    left = oracle.crop(pts, [-999, 0, -999, 999, -999, 999])
    right = oracle.crop(pts, [0, 999, -999, 999, -999, 999])
    assert len(left) + len(right) == len(pts)
    assert (left['x'] < 0).all() and (right['x'] >= 0).all()


def test_remove_outliers_invariant(oracle, synth):
    # reference test_remove_outliers (:528-541): 0 < n_out < n_in for (30, 1.0, perTile=True)
    pts, _ = synth(20000)
    out = oracle.remove_outliers(pts, 30, 1.0, True)
    assert 0 < len(out) < len(pts)


@pytest.mark.parametrize("sign", [1, -1])
def test_downsample_invariants(oracle, synth, sign):
    # reference test_downsample / test_downsample_voxelgrid (:543-587)
    pts, cs = synth(0)
    cellsize = cs / 2
    count = len(pts)
    while cellsize < 16:
        out, ocs = oracle.downsample(pts, cs, sign * cellsize)
        count = len(out)
        assert 1 <= count <= len(pts)
        assert ocs == pytest.approx(max(cellsize, cs))
        if count < 2:
            break
        cellsize *= 2
    assert count <= 8


def test_downsample_empty(oracle):
    # reference test_downsample_empty (:589-594): empty in -> empty out for the default path
    out, _ = oracle.downsample(oracle.empty(0), 0.0, 1.0)
    assert len(out) == 0
    with pytest.raises(oracle.OracleError):   # the plain VoxelGrid path reports an empty result as an error
        oracle.downsample(oracle.empty(0), 0.0, -1.0)


def test_downsample_voxel_counts_match_survey(oracle, synth):
    # distinct 0.01-voxels on the global lattice, from the survey's independent numpy restatement
    for npoints, expected in [(100000, 33368), (300000, 36088)]:
        pts, cs = synth(npoints)
        out, _ = oracle.downsample(pts, cs, -0.01)
        assert len(out) == expected
        # independent check of the voxel assignment: count distinct floor(p * 100) triples
        inv = np.float32(1.0) / np.float32(0.01)
        ijk = np.stack([np.floor(pts[f] * inv).astype(np.int64) for f in ('x', 'y', 'z')], axis=1)
        assert len(np.unique(ijk, axis=0)) == expected


def test_downsample_means_against_numpy(oracle, synth):
    # voxel means in float64 via numpy vs the oracle's fp32 sequential sums
    pts, cs = synth(100000)
    out, _ = oracle.downsample(pts, cs, -0.01)
    inv = np.float32(1.0) / np.float32(0.01)
    ijk = np.stack([np.floor(pts[f] * inv).astype(np.int64) for f in ('x', 'y', 'z')], axis=1)
    key = (ijk[:, 2] - ijk[:, 2].min()) * 10**8 + (ijk[:, 1] - ijk[:, 1].min()) * 10**4 + (ijk[:, 0] - ijk[:, 0].min())
    order = np.argsort(key, kind='stable')
    uk, start, cnt = np.unique(key[order], return_index=True, return_counts=True)
    assert len(uk) == len(out)
    for f in ('x', 'y', 'z'):
        mean = np.add.reduceat(pts[f][order].astype(np.float64), start) / cnt
        assert np.abs(mean - out[f]).max() < 2e-6
    tile_or = np.bitwise_or.reduceat(pts['tile'][order], start)
    assert (tile_or == out['tile']).all()
    for f in ('r', 'g', 'b'):
        s = np.add.reduceat(pts[f][order].astype(np.int64), start)
        expect = (s.astype(np.float32) / cnt.astype(np.float32)).astype(np.uint32)
        assert (expect == out[f]).all()


def test_octree_split_adds_duplicates_only(oracle, synth):
    # the positive path cuts voxels at leaf faces: never fewer outputs than the plain grid, same point mass
    pts, cs = synth(100000)
    info = {}
    pos, _ = oracle.downsample(pts, cs, 0.01, info)
    neg, _ = oracle.downsample(pts, cs, -0.01)
    assert info['n_leaves'] >= 1 and info['depth'] >= 1
    assert len(neg) <= len(pos) <= len(neg) + len(neg) // 20


def test_colorize_golden(oracle):
    # vectors produced by the reference's own ColorizeFilter._mapcolor (tests/golden/make_colorize_vectors.py)
    from cwipc_util_amd.filters.colorize import ColorizeFilter
    d = np.load(os.path.join(GOLDEN, "colorize_vectors.npz"))
    pts = d['input']
    for i in range(int(d['ncases'])):
        name, w = str(d[f'case{i}_cmap']), float(d[f'case{i}_weight'])
        cmap = name if name != 'uniform' else tuple(d[f'case{i}_uniform'])
        lut, valid = ColorizeFilter(w, cmap).colorMap.tables()
        assert (lut == d[f'case{i}_lut']).all() and (valid == d[f'case{i}_valid']).all(), "restated colour map differs"
        got = oracle.colorize(pts, w, lut, valid)
        assert got.tobytes() == d[f'case{i}_output'].tobytes(), f"case {i} ({name}, {w})"


def test_knn_against_bruteforce(oracle):
    rng = np.random.default_rng(7)
    n = 1500
    pts = oracle.empty(n)
    pts['x'], pts['y'], pts['z'] = rng.random(n).astype(np.float32), rng.random(n).astype(np.float32), (rng.random(n) * 0.05).astype(np.float32)
    k = 8
    got = oracle.knn_mean_dist(pts, k)
    xyz = np.stack([pts['x'], pts['y'], pts['z']], axis=1)
    for i in range(0, n, 37):
        d = xyz - xyz[i]
        d2 = (d[:, 0] * d[:, 0]).astype(np.float32)
        d2 = (d2 + (d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
        d2 = (d2 + (d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)
        nearest = np.sort(d2)[1:k + 1]
        expect = np.float32(np.sqrt(nearest, dtype=np.float32).astype(np.float64).sum() / k)
        assert got[i] == expect


def test_path_vectors_are_reproduced(oracle):
    """tests/golden/path_vectors.npz (made by tests/golden/make_path_vectors.py from this oracle): the restatement
    still says what it said when the vectors were committed -- a guard against drift of the checker itself."""
    d = np.load(os.path.join(GOLDEN, "path_vectors.npz"))
    pts, cs = d["input"], float(d["cellsize"])
    same = lambda a, b: len(a) == len(b) and a.tobytes() == b.tobytes()
    for name, cell in (("down_p05", 0.05), ("down_p20", 0.2), ("down_m05", -0.05), ("down_m20", -0.2)):
        res, out_cs = oracle.downsample(pts, cs, cell)
        assert same(res, d[name]) and np.float32(out_cs) == d[name + "_cellsize"]
    assert oracle.knn_mean_dist(pts, 8).tobytes() == d["knn8"].tobytes()
    assert same(oracle.remove_outliers(pts, 8, 1.0, False), d["sor_k8_s1"])
    assert same(oracle.remove_outliers(pts, 8, 1.0, True), d["sor_k8_s1_pertile"])
    assert same(oracle.tilefilter(pts, 1), d["tilefilter_1"])
    assert same(oracle.crop(pts, [-0.1, 0.2, 0.5, 1.5, -0.3, 0.05]), d["crop"])
    assert same(oracle.colormap(pts, 0x00ff00ff, 0x05000007), d["colormap"])
    assert same(oracle.tilemap(pts, bytes((i * 7 + 3) % 256 for i in range(256))), d["tilemap"])
    assert same(oracle.join(pts[:100], pts[300:]), d["join"])


def test_simulatecams_restatement_against_the_reference_vectors():
    """oracle.simulatecams against tiles produced by the reference's own SimulatecamsFilter(hard=True)
    (tests/golden/make_simulatecams_vectors.py ran /root/reference/python/cwipc/filters/simulatecams.py in isolation)."""
    import os
    from oracle import oracle as o
    data = np.load(os.path.join(os.path.dirname(__file__), "golden", "simulatecams_vectors.npz"))
    names = sorted(k[:-3] for k in data.keys() if k.endswith("_in"))
    assert len(names) >= 5
    for name in names:
        m, ncam, want = data[name + "_in"], int(data[name + "_ncam"]), data[name + "_tile"]
        got = o.simulatecams(m, ncam)
        assert (got == want).all(), (name, int((got != want).sum()))
