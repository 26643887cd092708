"""Parity of the HIP path against the CPU oracle, through the C-ABI (ctypes wrapper).

Bars:
  tilefilter / tilemap / crop / colormap / colorize / join: bit-exact content, count, order
  downsample: identical voxel set, count AND order; xyz within 1e-5 abs; rgb and tile exact
  remove_outliers: identical mask except points whose d_i lies within 1e-6 relative of the threshold
"""
import os

import numpy as np
import pytest

from conftest import make_cloud

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
XYZ_TOL = 1e-5   # north_star: "voxel centroids within 1e-5 float tolerance"


def same(a, b):
    return len(a) == len(b) and a.tobytes() == b.tobytes()


# ---------------------------------------------------------------------------
# container / copy path
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("npoints", [0, 1, 63, 64, 65, 1000, 100000])
def test_upload_download_roundtrip(gpu, oracle, synth, npoints):
    pts = synth(max(2 * npoints, 1000))[0][:npoints] if npoints else oracle.empty(0)
    assert len(pts) == npoints
    pc = make_cloud(gpu, pts, 0.5)
    if npoints:
        gpu.cwipc_hip_upload(pc, drop_host_copy=True)   # force the device -> host path
    assert pc.count() == npoints
    assert same(pc.get_numpy_array(), pts)
    pkt = pc.get_packet()
    assert same(gpu.cwipc_from_packet(pkt).get_numpy_array(), pts)


def test_page_locked_buffers_both_ways(gpu, synth, tmp_path):
    """The upload from a page-locked buffer has two ways (DMA + kernel, the default; the kernel reading the host buffer,
    CWIPC_PINNED_UPLOAD=kernel): the same cloud either way."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pts, cs = synth(200000, 0.3)
    np.save(str(tmp_path / "pts.npy"), pts)
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import cwipc_util_amd as cw\n"
        "pts = np.load(%r)\n"
        "p = cw.cwipc_hip_pinned_points(len(pts) + 5); p[5:] = pts\n"
        "pc = cw.cwipc_from_numpy_array(p[5:], 1); p[:] = np.zeros(1, dtype=p.dtype)[0]\n"
        "assert pc.get_numpy_array().tobytes() == pts.tobytes()\n"
        "np.save(sys.argv[1], cw.cwipc_tilefilter(pc, 2).get_numpy_array())\n"
    ) % (root, str(tmp_path / "pts.npy"))
    outs = []
    for mode in ("dma", "kernel"):
        out = str(tmp_path / (mode + ".npy"))
        subprocess.run([sys.executable, "-c", code, out], check=True, timeout=600, env=dict(os.environ, CWIPC_PINNED_UPLOAD=mode))
        outs.append(np.load(out))
    assert same(outs[0], outs[1]) and same(outs[0], pts[pts['tile'] == 2])


def test_page_locked_buffers_of_the_caller(gpu, oracle, synth):
    """Round 4, the copy path with buffers the DMA engines can reach (include/cwipc_util_amd/hip_ext.h: cwipc_hip_host_alloc /
    cwipc_hip_host_register): cwipc_from_points reads such a buffer from the device where it lies and has finished with it when it
    returns (the reference's from_points owns a copy at that point, src/cwipc_util.cpp:329-354), copy_uncompressed writes it from
    the device (:226-250).  Same bytes as through ordinary memory, whatever the offset inside the allocation."""
    pts, cs = synth(300000, 0.4)
    n = len(pts)
    pinned = gpu.cwipc_hip_pinned_points(n + 1000)
    pinned[37:37 + n] = pts                                     # a buffer that starts somewhere inside the allocation
    pc = gpu.cwipc_from_numpy_array(pinned[37:37 + n], 5)
    pc._set_cellsize(cs)
    pinned[:] = np.zeros(1, dtype=pinned.dtype)[0]              # the caller reuses its buffer right away
    assert same(pc.get_numpy_array(), pts)                      # (the host copy is made from the device now)
    out = gpu.cwipc_tilefilter(pc, 1)
    exp = oracle.tilefilter(pts, 1)
    dst = gpu.cwipc_hip_pinned_points(len(exp))
    assert out.copy_into(dst) == len(exp) and same(dst, exp)
    plain = np.zeros(len(exp), dtype=dst.dtype)
    assert out.copy_into(plain) == len(exp) and same(plain, exp)
    with pytest.raises(gpu.CwipcError):
        out.copy_into(np.zeros(len(exp) - 1, dtype=dst.dtype))  # the reference's size check (:231) stays
    # the caller's own array, registered
    own = pts.copy()
    with gpu.cwipc_hip_pin_array(own) as arr:
        pc2 = gpu.cwipc_from_numpy_array(arr, 6)
        pc2._set_cellsize(cs)
        got, _ = gpu.cwipc_downsample(pc2, 0.01).get_numpy_array(), None
        back = np.zeros_like(own)
        with gpu.cwipc_hip_pin_array(back) as b:
            assert pc2.copy_into(b) == n
        assert same(back, pts)
    e, _ = oracle.downsample(pts, cs, 0.01)
    assert len(got) == len(e) and (got['tile'] == e['tile']).all()
    # small clouds and empty ones take the ordinary way
    small = gpu.cwipc_hip_pinned_points(100)
    small[:] = pts[:100]
    assert same(gpu.cwipc_from_numpy_array(small, 1).get_numpy_array(), pts[:100])
    assert gpu.cwipc_from_numpy_array(small[:0], 1).count() == 0



# ---------------------------------------------------------------------------
# exact filters
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("npoints", [1000, 100000, 300000])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 255, 256, 257, -1])
def test_tilefilter(gpu, oracle, synth, npoints, tile):
    pts, cs = synth(npoints)
    pc = make_cloud(gpu, pts, cs, 777)
    out = gpu.cwipc_tilefilter(pc, tile)
    assert same(out.get_numpy_array(), oracle.tilefilter(pts, tile))
    assert out.timestamp() == 777 and out.cellsize() == pc.cellsize()


def test_tilefilter_config1(gpu, oracle, synth):
    """BASELINE config 1: synthetic 100k -> tilefilter(1) -> 49928 points."""
    pts, cs = synth(100000)
    out = gpu.cwipc_tilefilter(make_cloud(gpu, pts, cs), 1)
    assert out.count() == 49928
    got = out.get_numpy_array()
    assert (got['tile'] == 1).all() and same(got, oracle.tilefilter(pts, 1))


def test_tilefilter_ragged_and_random(gpu, oracle):
    rng = np.random.default_rng(1)
    for n in (1, 3, 4, 5, 255, 256, 257, 4095, 4096, 4097, 12345):
        pts = oracle.empty(n)
        pts['x'] = rng.random(n)
        pts['tile'] = rng.integers(0, 4, n)
        pts['r'] = rng.integers(0, 256, n)
        for tile in (0, 1, 3):
            assert same(gpu.cwipc_tilefilter(make_cloud(gpu, pts), tile).get_numpy_array(), oracle.tilefilter(pts, tile))


def test_compaction_on_either_side_of_the_small_cloud_limit(gpu, oracle):
    """Up to 256 k points a compaction runs on tiles of 1024 points with count and scan in one launch (the last workgroup to
    finish scans: csrc/kernels_basic.hip), beyond on tiles of 4096 with three launches: the same stable result either way, at
    the tile edges and at the limit itself, for every predicate (tile, crop box, keep everything, keep nothing)."""
    rng = np.random.default_rng(7)
    for n in (1023, 1024, 1025, 2048, 262143, 262144, 262145, 266240, 300001):
        pts = oracle.empty(n)
        pts['x'] = rng.random(n).astype(np.float32)
        pts['y'] = rng.random(n).astype(np.float32)
        pts['z'] = np.arange(n, dtype=np.float32)                    # (tells where a point came from: stability)
        pts['tile'] = rng.integers(1, 4, n)
        pts['g'] = rng.integers(0, 256, n)
        pc = make_cloud(gpu, pts)
        for tile in (1, 2, 7, 8):                                     # 7: every point, 8: none
            assert same(gpu.cwipc_tilefilter_masked(pc, tile).get_numpy_array(), pts[(pts['tile'] & tile) != 0]), (n, tile)
        box = [0.25, 0.75, 0.1, 0.9, -1.0, float(n)]
        assert same(gpu.cwipc_crop(pc, box).get_numpy_array(), oracle.crop(pts, box)), n
        # twice in a row on the same thread: the ticket word of the fused count + scan is left as it was found
        assert same(gpu.cwipc_tilefilter(pc, 1).get_numpy_array(), oracle.tilefilter(pts, 1)), n


def test_empty_inputs(gpu, oracle):
    pc = gpu.cwipc_from_points([], 0)
    assert gpu.cwipc_tilefilter(pc, 0).count() == 0                 # reference test_tilefilter_empty
    assert gpu.cwipc_downsample(pc, 1).count() == 0                 # reference test_downsample_empty
    assert gpu.cwipc_crop(pc, [0, 1, 0, 1, 0, 1]).count() == 0
    assert gpu.cwipc_colormap(pc, 0, 0).count() == 0
    assert gpu.cwipc_tilemap(pc, {1: 2}).count() == 0
    assert gpu.cwipc_join(pc, pc).count() == 0
    assert gpu.cwipc_remove_outliers(pc, 8, 1.0, False).count() == 0
    with pytest.raises(gpu.CwipcError):                              # plain VoxelGrid: empty result is an error
        gpu.cwipc_downsample(pc, -1)


def test_masked_tilefilter(gpu, oracle, synth):
    pts, cs = synth(100000)
    pts = pts.copy()
    pts['tile'] = (np.arange(len(pts)) % 7).astype(np.uint8)
    for mask in (1, 2, 6, 0):
        got = gpu.cwipc_tilefilter_masked(make_cloud(gpu, pts, cs), mask).get_numpy_array()
        assert same(got, pts[(pts['tile'] & mask) != 0])


def test_tilemap(gpu, oracle, synth):
    pts, cs = synth(100000)
    pc = make_cloud(gpu, pts, cs)
    m = [0] * 256
    m[1], m[2] = 5, 6
    assert same(gpu.cwipc_tilemap(pc, {1: 5, 2: 6}).get_numpy_array(), oracle.tilemap(pts, bytes(m)))
    perm = np.random.default_rng(3).permutation(256).astype(np.uint8)
    pts2 = pts.copy()
    pts2['tile'] = (np.arange(len(pts)) * 37 % 256).astype(np.uint8)
    assert same(gpu.cwipc_tilemap(make_cloud(gpu, pts2), bytes(perm)).get_numpy_array(), oracle.tilemap(pts2, bytes(perm)))
    # reference test_tilemap count identities
    mapped = gpu.cwipc_tilemap(pc, {1: 5, 2: 6})
    for a, b in ((1, 5), (2, 6), (5, 1), (6, 2)):
        assert gpu.cwipc_tilefilter(pc, a).count() == gpu.cwipc_tilefilter(mapped, b).count()


@pytest.mark.parametrize("clear,setb", [(0xffffffff, 0x010203), (0, 0), (0xff000000, 0x07000000), (0x00ff0000, 0x00800000),
                                        (0x000000ff, 0x00000011), (0x0000ff00, 0), (0x12345678, 0x9abcdef0)])
def test_colormap(gpu, oracle, synth, clear, setb):
    pts, cs = synth(100000, 1.25)
    got = gpu.cwipc_colormap(make_cloud(gpu, pts, cs), clear, setb).get_numpy_array()
    assert same(got, oracle.colormap(pts, clear, setb))


def test_colormap_known_answer(gpu, synth):
    """reference test_colormap: (0xffffffff, 0x010203) -> (r,g,b,tile) == (1,2,3,0), xyz unchanged."""
    pts, cs = synth(0)
    got = gpu.cwipc_colormap(make_cloud(gpu, pts, cs), 0xffffffff, 0x010203).get_numpy_array()
    assert (got['x'] == pts['x']).all() and (got['y'] == pts['y']).all() and (got['z'] == pts['z']).all()
    assert (got['r'] == 1).all() and (got['g'] == 2).all() and (got['b'] == 3).all() and (got['tile'] == 0).all()


def test_crop(gpu, oracle, synth):
    pts, cs = synth(100000)
    pc = make_cloud(gpu, pts, cs)
    for bbox in ([-999, 0, -999, 999, -999, 999], [0, 999, -999, 999, -999, 999], [-0.1, 0.1, 0.5, 1.5, -0.3, 0.0],
                 [0.1, 0.1, 0, 1, 0, 1], [float(pts['x'][5]), 1, -1, 3, -1, 1]):
        assert same(gpu.cwipc_crop(pc, bbox).get_numpy_array(), oracle.crop(pts, bbox))
    left = gpu.cwipc_crop(pc, [-999, 0, -999, 999, -999, 999])
    right = gpu.cwipc_crop(pc, [0, 999, -999, 999, -999, 999])
    assert left.count() + right.count() == len(pts)                 # reference test_crop


def test_join(gpu, oracle, synth):
    a, csa = synth(100000)
    b, csb = synth(1000, 0.5)
    pa, pb = make_cloud(gpu, a, csa, 50), make_cloud(gpu, b, csb, 40)
    j = gpu.cwipc_join(pa, pb)
    assert same(j.get_numpy_array(), oracle.join(a, b))
    assert j.timestamp() == 40 and j.cellsize() == min(pa.cellsize(), pb.cellsize())   # reference :411-414
    # n-ary join == left fold of the binary join (reference util.py:1330-1332)
    c, csc = synth(20000, 0.25)
    pcs = [pa, pb, make_cloud(gpu, c, csc, 45), gpu.cwipc_from_points([], 99)]
    multi = gpu.cwipc_join_multi(pcs)
    fold = oracle.join(oracle.join(oracle.join(a, b), c), oracle.empty(0))
    assert same(multi.get_numpy_array(), fold)
    assert multi.timestamp() == 40


def test_colorize_golden(gpu, oracle):
    """HIP colorize against vectors made by the reference's own ColorizeFilter._mapcolor."""
    from cwipc_util_amd.filters.colorize import ColorizeFilter
    d = np.load(os.path.join(GOLDEN, "colorize_vectors.npz"))
    pts = d['input']
    pc = make_cloud(gpu, pts, 0.01, 5)
    for i in range(int(d['ncases'])):
        name, w = str(d[f'case{i}_cmap']), float(d[f'case{i}_weight'])
        cmap = name if name != 'uniform' else tuple(d[f'case{i}_uniform'])
        flt = ColorizeFilter(w, cmap)
        out = flt.filter(pc)
        assert same(out.get_numpy_array(), d[f'case{i}_output']), f"case {i} ({name}, {w})"
        assert out.timestamp() == 5 and out.cellsize() == pc.cellsize()


def test_colorize_large(gpu, oracle, synth):
    from cwipc_util_amd.filters.colorize import ColorizeFilter
    pts, cs = synth(300000, 2.0)
    pts = pts.copy()
    pts['tile'] = (1 << (np.arange(len(pts)) % 9)).astype(np.uint8)   # includes 0 (1<<8 wraps)
    flt = ColorizeFilter(0.8, "camera")
    lut, valid = flt.colorMap.tables()
    assert same(flt.filter(make_cloud(gpu, pts, cs)).get_numpy_array(), oracle.colorize(pts, 0.8, lut, valid))


# ---------------------------------------------------------------------------
# geometry helpers (reference: numpy / Python loops in registration/util.py and filters/transform.py)
# ---------------------------------------------------------------------------
def test_transform_matrix(gpu, oracle, synth):
    pts, cs = synth(200000, 1.0)
    rng = np.random.default_rng(8)
    ang = 0.7
    m = np.eye(4)
    m[:3, :3] = [[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]
    m[:3, :3] = m[:3, :3] @ (np.eye(3) + 0.01 * rng.normal(size=(3, 3)))
    m[:3, 3] = [0.25, -1.5, 3.0]
    pc = make_cloud(gpu, pts, cs, 77)
    out = gpu.cwipc_transform(pc, m)
    got, exp = out.get_numpy_array(), oracle.transform(pts, m)
    assert out.timestamp() == 77 and out.cellsize() == pc.cellsize()
    assert same(got[['r', 'g', 'b', 'tile']], exp[['r', 'g', 'b', 'tile']])
    # bit for bit: the kernel rounds where numpy's matrix product rounds (a chain of fused multiply-adds in index order)
    for f in ('x', 'y', 'z'):
        assert (got[f] == exp[f]).all(), (f, int((got[f] != exp[f]).sum()))
    # the identity leaves the cloud as it is (but for the sign of zeros: -0 + 0 = +0, in numpy as here)
    assert same(gpu.cwipc_transform(pc, np.eye(4)).get_numpy_array(), oracle.transform(pts, np.eye(4)))


def test_transform_filter_offset_scale(gpu, oracle, synth):
    from cwipc_util_amd.filters.transform import TransformFilter
    pts, cs = synth(100000, 0.3)
    pc = make_cloud(gpu, pts, cs, 5)
    flt = TransformFilter(0.1, -1.0, 2.5, 1.7)
    out = flt.filter(pc)
    assert same(out.get_numpy_array(), oracle.offset_scale(pts, 0.1, -1.0, 2.5, 1.7))   # bit-exact: same f64 operations
    assert out.timestamp() == 5
    assert out.cellsize() == pytest.approx(np.float32(np.float64(np.float32(cs)) * 1.7), rel=0, abs=0)
    assert same(gpu.cwipc_offset_scale(make_cloud(gpu, pts[:0], cs), 1, 2, 3, 4).get_numpy_array(), pts[:0])


# ---------------------------------------------------------------------------
# outputs of the reference's own Python (tests/golden/helper_vectors.npz, made by tests/golden/make_helper_vectors.py from the
# function bodies under /root/reference in the build container): the GPU paths must reproduce them, the oracle is not in the loop
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def helper_vectors():
    import json
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "helper_vectors.npz"))
    return d, json.loads(bytes(d["meta_json"]).decode())


def test_reference_vectors_masked_tilefilter_and_tiles_used(gpu, helper_vectors):
    """cwipc_tilefilter_masked, get_tiles_used (reference python/cwipc/registration/util.py:98-112, 285-293)."""
    d, _ = helper_vectors
    for i in range(6):
        pts = d["masked%d_in" % i]
        pc = make_cloud(gpu, pts, 0.003 + 0.001 * i, 100 + i)
        assert gpu.get_tiles_used(pc) == d["masked%d_tiles_used" % i].tolist()
        for m in (0, 1, 2, 3, 4, 8, 15, 128, 255):
            out = gpu.cwipc_tilefilter_masked(pc, m)
            exp = d["masked%d_mask%d_out" % (i, m)]
            assert same(out.get_numpy_array(), exp), (i, m)
            ts, cs = d["masked%d_mask%d_meta" % (i, m)]
            assert out.timestamp() == ts, (i, m)
            if len(exp):   # (the reference's empty result is cwipc_from_points([], ts): its cellsize is 0; an empty cloud's cellsize says nothing)
                assert out.cellsize() == cs, (i, m)


def test_reference_vectors_transform(gpu, helper_vectors):
    """cwipc_transform (reference python/cwipc/registration/util.py:295-309: numpy's float64 matrix product of the 3 x 3 block with
    the float32 coordinates, plus the translation, stored as float32) -- bit for bit what the reference function returned."""
    d, _ = helper_vectors
    pts = d["transform_in"]
    for name in ("identity", "rot_y_45", "translate", "general", "scale_shear"):
        out = gpu.cwipc_transform(make_cloud(gpu, pts, 0.004, 7), d["transform_%s_matrix" % name])
        got, exp = out.get_numpy_array(), d["transform_%s_out" % name]
        assert same(got, exp), (name, int((got['x'] != exp['x']).sum()), int((got['y'] != exp['y']).sum()), int((got['z'] != exp['z']).sum()))
        assert [out.timestamp(), out.cellsize()] == d["transform_%s_meta" % name].tolist()


def test_reference_vectors_transform_filter(gpu, helper_vectors):
    """TransformFilter (reference python/cwipc/filters/transform.py:32-49: a Python loop over the points, float64 arithmetic on
    C floats): points, timestamp and cellsize as the reference filter returned them."""
    from cwipc_util_amd.filters.transform import TransformFilter
    d, _ = helper_vectors
    pts = d["offsetscale_in"]
    for i in range(4):
        x, y, z, scale = d["offsetscale%d_params" % i]
        out = TransformFilter(x, y, z, scale).filter(make_cloud(gpu, pts, 0.005, 9))
        assert same(out.get_numpy_array(), d["offsetscale%d_out" % i]), i
        assert [out.timestamp(), out.cellsize()] == d["offsetscale%d_meta" % i].tolist(), i


def test_reference_vectors_downsample_pertile(gpu, oracle, helper_vectors):
    """cwipc_downsample_pertile: the reference function's calls (recorded in the fixture: tiles ascending, tilefilter -> downsample
    per tile, a left fold of joins) replayed with the oracle's filters give the cloud the product's one call must return."""
    _, meta = helper_vectors
    rng = np.random.default_rng(5)
    for case in meta["pertile"]:
        tiles = np.array(case["tiles_in_cloud"], dtype=np.uint8)
        pts = oracle.empty(20000)
        pts['x'], pts['y'], pts['z'] = rng.random(20000) * 0.5, rng.random(20000) * 0.5, rng.random(20000) * 0.5
        pts['r'], pts['g'], pts['b'] = rng.integers(0, 256, 20000), rng.integers(0, 256, 20000), rng.integers(0, 256, 20000)
        pts['tile'] = tiles[rng.integers(0, len(tiles), 20000)]
        clouds = {}
        for call in case["calls"]:      # the reference's calls, in its order
            if call[0] == "tilefilter":
                clouds[json_key(["tile", call[1]])] = oracle.tilefilter(pts, call[1])
            elif call[0] == "downsample":
                clouds[json_key(["down", call[1]])] = oracle.downsample(clouds[json_key(call[1])], 0.001, call[2])[0]
            else:
                clouds[json_key(["join", call[1], call[2]])] = oracle.join(clouds[json_key(call[1])], clouds[json_key(call[2])])
        exp = clouds[json_key(case["result"])]
        out = gpu.cwipc_downsample_pertile(make_cloud(gpu, pts, 0.001, 55), 0.0125)
        got = out.get_numpy_array()
        assert len(got) == len(exp) and out.timestamp() == 55
        for f in ('r', 'g', 'b', 'tile'):
            assert (got[f] == exp[f]).all(), f
        for f in ('x', 'y', 'z'):
            assert np.abs(got[f].astype(np.float64) - exp[f]).max() <= XYZ_TOL, f


def json_key(obj):
    import json
    return json.dumps(obj)


def test_reference_vectors_synchronizer_with_the_gpu_join(gpu, helper_vectors):
    """The product's synchroniser with its real join (cwipc_join_multi on the device) on the scripts the reference's
    `_Synchronizer.run` was driven with (reference python/cwipc/net/source_synchronizer.py:106-200): the same fused clouds --
    which inputs, in which order, timestamp, cellsize -- and the same late / desync / missing statistics."""
    from cwipc_util_amd.net.source_synchronizer import SyncCore
    from oracle.synchronizer import ScriptedSource
    _, meta = helper_vectors
    for case in meta["synchronizer"][:16]:
        sources = []
        for t, rows in enumerate(case["script"]):
            clouds = []
            for k, (ts, cs, n, _gate) in enumerate(rows):
                p = np.zeros(n, dtype=gpu.cwipc_point_numpy_dtype)
                p['x'] = np.arange(n)
                p['y'], p['z'], p['tile'] = t, k, 1 << t
                clouds.append(make_cloud(gpu, p, cs, ts))
            sources.append(ScriptedSource(clouds, [r[3] for r in rows]))
        core = SyncCore(sources, prefer_partial_over_unsynced=case["prefer_partial_over_unsynced"])
        produced = []
        for _ in range(100000):
            if any(s.eof() for s in core.sources):
                break
            r = core.poll()
            if r is not None:
                arr = r.get_numpy_array()
                parts = []
                for i in range(len(arr)):
                    key = [int(arr['y'][i]), int(arr['z'][i])]
                    if not parts or parts[-1] != key:
                        parts.append(key)
                produced.append({"timestamp": r.timestamp(), "cellsize": r.cellsize(), "count": r.count(), "parts": parts})
        assert produced == case["produced"]
        assert {"late": core.late_per_occurrence, "desync": core.desync_per_occurrence, "missing": core.missing_per_occurrence} == case["stats"]


@pytest.mark.parametrize("npoints,angle", [(0, 0.0), (100000, 0.7), (1000000, 2.5), (300000, 1.6), (10000000, 0.0)])
def test_synthetic_source_on_the_device(gpu, oracle, npoints, angle):
    """With a GPU the synthetic source writes its cloud straight into device planes (csrc/synthetic.cpp); the host generator
    (the oracle restates it, reference src/cwipc_synthetic.cpp:182-222) is its checker: every byte equal, the white patch
    ("eyes", :206-210) included -- angle 1.6 has fmod(angle, pi/2) < 0.08, the others not."""
    import struct
    src = gpu.cwipc_synthetic(0, npoints)
    src.start()
    out = bytearray(4)
    assert src.auxiliary_operation("amd-fixangle", struct.pack("f", angle), out)
    pc = src.get()
    src.stop()
    assert gpu.util.cwipc_util_dll_load().cwipc_hip_is_device_resident(pc.as_cwipc_p()) == 1
    want, cs = oracle.synthetic(npoints, angle)
    got = pc.get_numpy_array()
    assert pc.count() == len(want) and pc.cellsize() == pytest.approx(cs, rel=0, abs=0)
    differing = int((got.view(np.uint8).reshape(-1, 16) != want.view(np.uint8).reshape(-1, 16)).any(axis=1).sum())
    assert differing == 0, differing
    # and the cloud behaves like any other: the downsample anchors its octree at the first point without fetching it
    if npoints and npoints <= 1000000:
        if npoints <= 300000:
            check_downsample(gpu, oracle, want, cs, 0.01)
        assert same(gpu.cwipc_downsample(pc, 0.01).get_numpy_array(), gpu.cwipc_downsample(make_cloud(gpu, want, cs), 0.01).get_numpy_array())


def test_simulatecams_golden(gpu):
    """SimulatecamsFilter(hard=True) against vectors produced by the reference's own filter (tests/golden/make_simulatecams_vectors.py):
    the tile of every point bit for bit, everything else untouched."""
    from cwipc_util_amd.filters import factory
    data = np.load(os.path.join(GOLDEN, "simulatecams_vectors.npz"))
    for name in ("blob4", "blob8", "ring3", "ring8", "symmetric6"):
        m, ncam, want = data[name + "_in"], int(data[name + "_ncam"]), data[name + "_tile"]
        pc = gpu.cwipc_from_numpy_matrix(m, 77)
        pc._set_cellsize(0.125)
        out = factory("simulatecams(%d, True)" % ncam).filter(pc)
        got = out.get_numpy_array()
        assert (got['tile'] == want).all(), (name, int((got['tile'] != want).sum()))
        src = pc.get_numpy_array()
        for f in ('x', 'y', 'z', 'r', 'g', 'b'):
            assert (got[f] == src[f]).all(), (name, f)
        assert out.timestamp() == 77 and out.cellsize() == 0.125
        assert gpu.util.cwipc_util_dll_load().cwipc_hip_is_device_resident(out.as_cwipc_p()) == 1
    # the soft rule draws random numbers: every point goes to its best or second-best camera
    m, ncam = data["ring8_in"], 8
    soft = factory("simulatecams(8, False, 2.0)").filter(gpu.cwipc_from_numpy_matrix(m, 1)).get_numpy_array()
    hard = data["ring8_tile"]
    assert set(np.unique(soft['tile'])) <= {1 << c for c in range(8)}
    assert (soft['tile'] == hard).mean() > 0.5


def test_native_downsample_app(gpu, oracle, synth, tmp_path):
    """The reference's native tool and ctest (apps/cwipc_downsample: `cwipc_downsample 0.1 in.ply out.ply`) as a program of
    our own linked against this library (tests/abi/downsample_app.cpp): PLY in, filter through the C ABI and the C++ virtuals,
    PLY out (ASCII and binary), no objects left behind; the file it writes holds the oracle's result."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++ here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "cwipc_util_amd", "lib")
    exe = str(tmp_path / "downsample_app")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "abi", "downsample_app.cpp"),
                    "-o", exe, "-L" + libdir, "-lcwipc_util", "-Wl,-rpath," + libdir], check=True)
    pts, cs = synth(100000, 0.4)
    src = str(tmp_path / "in.ply")
    gpu.cwipc_write(src, make_cloud(gpu, pts, cs, 1), True)
    for cell, flavour in ((0.1, []), (0.02, ["binary"]), (-0.05, [])):
        dst = str(tmp_path / ("out_%s.ply" % cell))
        run = subprocess.run([exe, str(cell), src, dst] + flavour, capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, run.stderr
        got = gpu.cwipc_read(dst, 9).get_numpy_array()
        exp, _ = oracle.downsample(pts, 0.0, cell)   # (a PLY file carries no cellsize: the tool's input has none)
        assert len(got) == len(exp), (cell, len(got), len(exp))
        for f in ('r', 'g', 'b', 'tile'):
            assert (got[f] == exp[f]).all(), (cell, f)
        tol = XYZ_TOL if flavour else 2e-5   # ASCII files hold 8 significant digits
        for f in ('x', 'y', 'z'):
            assert np.abs(got[f].astype(np.float64) - exp[f]).max() <= tol, (cell, f)
    assert subprocess.run([exe, "0.1", str(tmp_path / "missing.ply"), str(tmp_path / "x.ply")], capture_output=True).returncode == 1
    assert subprocess.run([exe], capture_output=True).returncode == 2


def test_tiles_used(gpu, oracle, synth):
    pts, cs = synth(100000)
    assert gpu.get_tiles_used(make_cloud(gpu, pts, cs)) == oracle.tiles_used(pts) == [1, 2]
    pts = pts.copy()
    pts['tile'] = (np.arange(len(pts)) * 37 % 251).astype(np.uint8)
    pts['tile'][::1000] = 255
    assert gpu.get_tiles_used(make_cloud(gpu, pts, cs)) == oracle.tiles_used(pts)
    assert gpu.get_tiles_used(make_cloud(gpu, pts[:0], cs)) == []


def test_tile_sets_known_without_looking(gpu, oracle, synth):
    """What a device cloud knows about its tiles (a tilemap with one target, a tile filter's result, a census, the union in a
    join) lets cwipc_tilefilter answer without a kernel where the answer is "all" or "none" -- a camera's tile through its own
    mask is the first step of the reference's per-tile chain (python/cwipc/registration/util.py:170-182).  Results are the
    oracle's either way; the planes are shared where nothing is copied."""
    pts, cs = synth(100000)
    pc = make_cloud(gpu, pts, cs, 5)
    def planes(c):
        return gpu.cwipc_hip_device_planes(c)[:4]
    cam = gpu.cwipc_tilemap(pc, bytes([8]) * 256)                       # every point -> tile 8: the set is {8}
    exp_cam = oracle.tilemap(pts, np.full(256, 8, dtype=np.uint8))
    own = gpu.cwipc_tilefilter(cam, 8)
    assert same(own.get_numpy_array(), oracle.tilefilter(exp_cam, 8)) and planes(own) == planes(cam)
    other = gpu.cwipc_tilefilter(cam, 4)
    assert other.count() == 0 and other.timestamp() == 5 and other.cellsize() == cam.cellsize()
    # through filters that keep the tile words, and a colormap that rewrites them
    moved = gpu.cwipc_transform(cam, np.eye(4))
    assert planes(gpu.cwipc_tilefilter(moved, 8)) == planes(moved)
    recol = gpu.cwipc_colormap(cam, 0xff000000, 0x05000000)             # tile byte cleared, set to 5
    assert gpu.cwipc_tilefilter(recol, 8).count() == 0 and gpu.cwipc_tilefilter(recol, 5).count() == len(pts)
    assert same(gpu.cwipc_tilefilter(recol, 5).get_numpy_array(), oracle.tilefilter(oracle.colormap(exp_cam, 0xff000000, 0x05000000), 5))
    # a join knows the union: both filters really filter, a third value is empty
    cam2 = gpu.cwipc_tilemap(pc, bytes([16]) * 256)
    both = gpu.cwipc_join(cam, cam2)
    exp_both = oracle.join(exp_cam, oracle.tilemap(pts, np.full(256, 16, dtype=np.uint8)))
    for t in (8, 16, 1):
        assert same(gpu.cwipc_tilefilter(both, t).get_numpy_array(), oracle.tilefilter(exp_both, t)), t
    # an uploaded cloud knows nothing until somebody takes a census; afterwards the one-tile half is handed on as it is
    t1 = gpu.cwipc_tilefilter(pc, 1)                                     # (a real filter; its result's set is {1})
    assert same(t1.get_numpy_array(), oracle.tilefilter(pts, 1)) and planes(gpu.cwipc_tilefilter(t1, 1)) == planes(t1)
    up = make_cloud(gpu, oracle.tilefilter(pts, 2), cs, 6)
    assert gpu.get_tiles_used(up) == [2]
    again = gpu.cwipc_tilefilter(up, 2)
    assert planes(again) == planes(up) and gpu.cwipc_tilefilter(up, 1).count() == 0


# ---------------------------------------------------------------------------
# voxel downsample
# ---------------------------------------------------------------------------
def voxel_population(pts, leaf, out):
    """Points of `pts` in the voxel (global lattice, fp32 arithmetic of pcl::VoxelGrid) each point of `out` lies in.
    An upper bound for the octree path, where a voxel cut by a leaf face is emitted once per leaf."""
    inv = np.float32(1.0) / np.float32(leaf)
    def keys(a):
        ijk = [np.floor(a[f] * inv).astype(np.int64) + (1 << 20) for f in ('x', 'y', 'z')]
        return (ijk[2] << 42) | (ijk[1] << 21) | ijk[0]
    uniq, cnt = np.unique(keys(pts), return_counts=True)
    k = keys(out)
    pos = np.clip(np.searchsorted(uniq, k), 0, len(uniq) - 1)
    # a centroid that rounds onto a voxel face may land in the neighbour: treat it as crowded
    return np.where(uniq[pos] == k, cnt[pos], len(pts))


TOLERANCE_USED = {}   # what the downsample checks measured, by (points, cellsize): printed at the end of the session (conftest) and asserted below


def check_downsample(gpu, oracle, pts, pc_cellsize, cellsize, ordered=True):
    pc = make_cloud(gpu, pts, pc_cellsize, 4242)
    out = gpu.cwipc_downsample(pc, cellsize)
    exp, exp_cs, mean64, count = oracle.downsample_audit(pts, pc_cellsize, cellsize)
    got = out.get_numpy_array()
    assert len(got) == len(exp), (len(got), len(exp))
    assert out.timestamp() == 4242
    assert out.cellsize() == pytest.approx(exp_cs, rel=0, abs=0)
    if not ordered:
        order_g, order_e = np.argsort(got, order=['z', 'y', 'x']), np.argsort(exp, order=['z', 'y', 'x'])
        got, exp, mean64, count = got[order_g], exp[order_e], mean64[order_e], count[order_e]
    # The bar is north_star's 1e-5 wherever a voxel holds at most a few hundred points (the typical voxel of every
    # BASELINE configuration: 254 points on average at 10 M).  The reference algorithm keeps an fp32
    # running sum per voxel (pcl AccumulatorXYZ), whose own rounding error grows with the number of
    # points in the voxel; the HIP path sums exact integers and rounds once.  For crowded voxels (coarse cells, or the apex of
    # the synthetic shape where whole rows collapse into one voxel) the bound on |HIP - oracle| widens per voxel with
    # that error model -- and what is USED of it is measured: the oracle also hands out every output's mean in float64
    # (oracle_downsample_audit; not part of the reference), against which the HIP path must stay within
    # leaf * 2^-22 + 1.5 ulp whatever the population (the fp32 product p * inv_leaf costs half an ulp of the coordinate, the
    # offset inside the voxel 2^-23 of a voxel, the final rounding to fp32 half an ulp).
    if len(exp):
        fin = np.isfinite(pts['x']) & np.isfinite(pts['y']) & np.isfinite(pts['z'])   # the filter skips the others
        pop = count.astype(np.int64)
        maxabs = max(float(np.abs(pts[f][fin]).max()) for f in ('x', 'y', 'z')) if fin.any() else 0.0
        # (the same model covers clouds far from the origin: the oracle's running sums lose sqrt(pop) ulps of the
        # coordinate magnitude; the strict bar applies where BASELINE lives: |coordinates| <= 4, <= 300 points per voxel)
        model = np.maximum(XYZ_TOL, 4.0 * np.sqrt(pop) * float(np.spacing(np.float32(maxabs))))
        tol = np.where((pop <= 300) & (maxabs <= 4.0), XYZ_TOL, model)
        leaf = float(max(abs(cellsize), pc_cellsize))
        used = {"outputs": int(len(exp)), "outputs_above_300_points": int((pop > 300).sum()), "largest_population": int(pop.max())}
        for i, f in enumerate(('x', 'y', 'z')):
            g = got[f].astype(np.float64)
            err = np.abs(g - exp[f].astype(np.float64))
            worst = int(np.argmax(err - tol))
            assert (err <= tol).all(), (f, float(err[worst]), float(tol[worst]), int(pop[worst]))
            # against the exact mean: the HIP path everywhere, the oracle's fp32 sums for comparison
            e64 = np.abs(g - mean64[:, i])
            bound = leaf * 2.0 ** -22 + 1.5 * np.spacing(np.abs(mean64[:, i]).astype(np.float32)).astype(np.float64)
            w64 = int(np.argmax(e64 - bound))
            assert (e64 <= bound).all(), ("HIP vs float64 mean", f, float(e64[w64]), float(bound[w64]), int(pop[w64]))
            small = pop <= 300
            used.setdefault("hip_vs_oracle_max_pop_le_300", 0.0)
            used.setdefault("hip_vs_oracle_max_pop_gt_300", 0.0)
            used.setdefault("hip_vs_f64_mean_max", 0.0)
            used.setdefault("oracle_vs_f64_mean_max", 0.0)
            if small.any(): used["hip_vs_oracle_max_pop_le_300"] = max(used["hip_vs_oracle_max_pop_le_300"], float(err[small].max()))
            if (~small).any(): used["hip_vs_oracle_max_pop_gt_300"] = max(used["hip_vs_oracle_max_pop_gt_300"], float(err[~small].max()))
            used["hip_vs_f64_mean_max"] = max(used["hip_vs_f64_mean_max"], float(e64.max()))
            used["oracle_vs_f64_mean_max"] = max(used["oracle_vs_f64_mean_max"], float(np.abs(exp[f].astype(np.float64) - mean64[:, i]).max()))
        TOLERANCE_USED[(len(pts), float(cellsize))] = used
        print("downsample tolerance used: %d points, cellsize %g: %s" % (len(pts), cellsize, used))
    for f in ('r', 'g', 'b', 'tile'):
        assert (got[f] == exp[f]).all(), f
    return got, exp


@pytest.mark.parametrize("npoints", [1000, 100000, 300000, 2000000])
@pytest.mark.parametrize("cellsize", [0.01, -0.01])
def test_downsample_synthetic(gpu, oracle, synth, npoints, cellsize):
    pts, cs = synth(npoints)
    check_downsample(gpu, oracle, pts, cs, cellsize)


@pytest.mark.parametrize("cellsize", [0.003, 0.05, 0.3, 1.0, 7.5])
def test_downsample_cellsizes(gpu, oracle, synth, cellsize):
    pts, cs = synth(100000, 0.7)
    check_downsample(gpu, oracle, pts, cs, cellsize)
    check_downsample(gpu, oracle, pts, cs, -cellsize)


@pytest.mark.parametrize("tiles", ["zero", "high_bits", "mixed"])
@pytest.mark.parametrize("cellsize", [0.01, -0.01])
def test_downsample_tile_bits_of_every_kind(gpu, oracle, synth, tiles, cellsize):
    """A voxel's tile is the OR of its points' tiles.  The fast accumulate kernel's flush sends a record word only if it has
    something to add (csrc/voxel_k1_fast.inc): clouds whose tiles are all 0 (neither tile word is sent), all in bits 4-7 (only
    the second one), and a mixture inside voxels and across workgroups (2 M points: a voxel's points lie in several ranges)."""
    pts, cs = synth(2_000_000)
    pts = pts.copy()
    n = len(pts)
    if tiles == "zero":
        pts['tile'] = 0
    elif tiles == "high_bits":
        pts['tile'] = np.where(np.arange(n) < n // 2, 0x10, 0xa0).astype(np.uint8)
    else:
        pts['tile'] = np.array([0, 0x10, 3, 0x84, 1, 0, 0x40, 2], dtype=np.uint8)[(np.arange(n) // 1000) % 8]
    check_downsample(gpu, oracle, pts, cs, cellsize)


def test_downsample_reference_loop(gpu, synth):
    """reference test_downsample / test_downsample_voxelgrid: doubling cellsize ends with <= 8 points."""
    pts, cs = synth(0)
    for sign in (1, -1):
        pc = make_cloud(gpu, pts, cs, 99)
        cellsize = cs / 2
        count = len(pts)
        while cellsize < 16:
            out = gpu.cwipc_downsample(pc, sign * cellsize)
            count = out.count()
            assert 1 <= count <= len(pts)
            assert out.timestamp() == 99
            if count < 2:
                break
            cellsize *= 2
        assert count <= 8


def test_downsample_means_are_correctly_rounded(gpu, oracle, synth):
    """Independent of the oracle: voxel means against numpy float64 (plain grid, where grouping is a lexsort)."""
    for cell, npoints in ((0.01, 300000), (0.3, 100000), (2.0, 100000)):
        pts, cs = synth(npoints, 0.4)
        got = gpu.cwipc_downsample(make_cloud(gpu, pts, cs), -cell).get_numpy_array()
        leaf = np.float32(max(cell, cs))
        inv = np.float32(1.0) / leaf
        ijk = np.stack([np.floor(pts[f] * inv).astype(np.int64) for f in ('x', 'y', 'z')], axis=1)
        order = np.lexsort((ijk[:, 0], ijk[:, 1], ijk[:, 2]))
        sk = ijk[order]
        start = np.flatnonzero(np.r_[True, (sk[1:] != sk[:-1]).any(axis=1)])
        cnt = np.diff(np.r_[start, len(pts)])
        assert len(got) == len(start)
        for f in ('x', 'y', 'z'):
            mean = np.add.reduceat(pts[f][order].astype(np.float64), start) / cnt
            err = np.abs(got[f].astype(np.float64) - mean)
            # position inside the voxel = fract(fl(p / leaf)) to 2^-23 of a voxel: the fp32 product costs up to half an ulp of
            # the coordinate, the final rounding to fp32 another half
            assert err.max() <= float(leaf) * 2.0 ** -22 + 1.5 * np.spacing(np.float32(np.abs(mean).max())), (cell, f, err.max())
        assert (got['tile'] == np.bitwise_or.reduceat(pts['tile'][order], start)).all()


def test_downsample_permuted_input(gpu, oracle, synth):
    """Supplemental robustness input: the same points in random order (no spatial coherence)."""
    pts, cs = synth(300000)
    perm = np.random.default_rng(20260129).permutation(len(pts))
    check_downsample(gpu, oracle, pts[perm], cs, 0.01)
    check_downsample(gpu, oracle, pts[perm], cs, -0.01)


def test_downsample_stream_of_permuted_frames_goes_through_the_partition_pass(gpu, oracle, synth):
    """After a few calls on clouds in no spatial order the library moves the points into spatial buckets in front of the
    accumulate kernel (voxel_partition.inc).  The octree's shape depends on the order of the points AS THEY CAME, so every
    call of the stream must still equal the oracle's walk over the permuted cloud: octree variant (leaf lattice, growth
    history) and plain grid, with points that do not count in between; then the same cloud in scan order (the partition
    switches itself off again) -- results checked the same way throughout."""
    pts, cs = synth(300000)
    rng = np.random.default_rng(77)
    perm = pts[rng.permutation(len(pts))]
    for cellsize in (0.01, -0.01):
        if cellsize > 0:   # (pcl::VoxelGrid on a dense cloud does not skip them: the plain grid is tested without, as everywhere here)
            perm['x'][1000] = np.nan
            perm['z'][250000] = np.inf
        else:
            perm['x'][1000], perm['z'][250000] = 0.0, 0.0
        pc = make_cloud(gpu, perm, cs, 4242)
        exp, exp_cs = oracle.downsample(perm, cs, cellsize)
        for call in range(8):
            got = gpu.cwipc_downsample(pc, cellsize).get_numpy_array()
            assert len(got) == len(exp), (cellsize, call, len(got), len(exp))
            for f in ('r', 'g', 'b', 'tile'):
                assert (got[f] == exp[f]).all(), (cellsize, call, f)
            for f in ('x', 'y', 'z'):
                assert np.abs(got[f].astype(np.float64) - exp[f]).max() <= XYZ_TOL, (cellsize, call, f)
        check_downsample(gpu, oracle, pts, cs, cellsize)
        check_downsample(gpu, oracle, pts, cs, cellsize)
    if os.environ.get("CWIPC_VOXEL_PARTITION") != "0":
        with gpu.cwipc_hip_profile() as prof:
            pc = make_cloud(gpu, perm, cs, 1)
            for call in range(6):
                gpu.cwipc_downsample(pc, 0.01)
        assert "partition_scatter" in prof.kernels, sorted(prof.kernels)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_partition_pass_on_random_shuffled_clouds(gpu, oracle, seed):
    """The partition pass on clouds it was not tuned on: several points per voxel, voxels scattered along a line tens of metres
    long (the 16 x 16 x 16 window of coarse cells folds over many times: far-apart cells share a bucket), on a plane, or in
    clusters with far outliers; random order; non-finite points in between.  A stream of calls, every call equal to the
    oracle's walk over the same shuffled cloud, octree variant and plain grid."""
    rng = np.random.default_rng(seed)
    cell = float(rng.choice([0.01, 0.02, 0.005]))
    shape = ["line", "plane", "clusters", "line"][seed - 1]
    nvox = int(rng.integers(15000, 30000))
    if shape == "line":
        centres = np.stack([rng.random(nvox) * 6000 * cell, rng.random(nvox) * 20 * cell, rng.random(nvox) * 20 * cell], axis=1)
    elif shape == "plane":
        centres = np.stack([rng.random(nvox) * 600 * cell, rng.random(nvox) * 600 * cell, 3 * cell * np.sin(rng.random(nvox) * 6)], axis=1)
    else:
        blobs = rng.random((12, 3)) * 300 * cell
        centres = blobs[rng.integers(0, 12, nvox)] + rng.normal(0, 12 * cell, (nvox, 3))
        centres[::997] += 2000 * cell
    per = int(rng.integers(5, 9))
    # `per` points inside each chosen voxel (corners on the voxel lattice, so that they do share it)
    corners = np.floor(centres / cell) * cell + round(float(rng.choice([0.0, -40.0, 7.3])) / cell) * cell
    xyz = (np.repeat(corners, per, axis=0) + (0.05 + 0.9 * rng.random((nvox * per, 3))) * cell).astype(np.float32)
    n = len(xyz)
    pts = oracle.empty(n)
    pts['x'], pts['y'], pts['z'] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts['r'], pts['g'], pts['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    pts['tile'] = 1 << rng.integers(0, 8, n)
    pts = pts[rng.permutation(n)]
    for variant in (cell, -cell):
        p = pts.copy()
        if variant > 0:
            bad = rng.integers(1, n, 20)
            p['y'][bad] = np.nan
        if variant < 0 and shape == "line":
            continue   # (pcl::VoxelGrid refuses a grid of this extent at this cell size: the 2^31 rule; the octree path is the one for it)
        pc = make_cloud(gpu, p, 0.0, 5)
        try:
            exp, _ = oracle.downsample(p, 0.0, variant)
        except oracle.OracleError:   # the 2^31-cell rule of pcl::VoxelGrid: the reference returns NULL, and so does the library
            with pytest.raises(gpu.CwipcError):
                gpu.cwipc_downsample(pc, variant)
            continue
        assert n > 4 * len(exp)
        for call in range(6):
            got = gpu.cwipc_downsample(pc, variant).get_numpy_array()
            assert len(got) == len(exp), (shape, variant, call, len(got), len(exp))
            for f in ('r', 'g', 'b', 'tile'):
                assert (got[f] == exp[f]).all(), (shape, variant, call, f)
            scale = max(float(np.abs(p[f][np.isfinite(p[f])]).max()) for f in ('x', 'y', 'z'))
            tol = max(XYZ_TOL, 4.0 * float(np.spacing(np.float32(scale))))
            for f in ('x', 'y', 'z'):
                assert np.abs(got[f].astype(np.float64) - exp[f]).max() <= tol, (shape, variant, call, f)
    if os.environ.get("CWIPC_VOXEL_PARTITION") != "0":
        pc = make_cloud(gpu, pts, 0.0, 5)
        with gpu.cwipc_hip_profile() as prof:
            for call in range(5):
                gpu.cwipc_downsample(pc, cell)
        assert "partition_scatter" in prof.kernels, (shape, sorted(prof.kernels))


def test_downsample_shifted_and_rotated(gpu, oracle, synth):
    """Clouds away from the origin, negative coordinates, several octree growth steps in every direction."""
    pts, cs = synth(100000)
    rng = np.random.default_rng(5)
    for shift in ((10.0, -3.0, 7.5), (-25.0, 0.0, -0.125), (0.64, 0.64, 0.64)):
        p = pts.copy()
        p['x'] += np.float32(shift[0]); p['y'] += np.float32(shift[1]); p['z'] += np.float32(shift[2])
        check_downsample(gpu, oracle, p, cs, 0.01)
        check_downsample(gpu, oracle, p[::-1].copy(), cs, 0.01)           # reversed insertion order
        check_downsample(gpu, oracle, p[rng.permutation(len(p))], cs, 0.02)


def test_downsample_leaf_face_through_a_first_point_near_zero(gpu, oracle, synth):
    """The octree's first box puts a leaf face through the first point (PCL: min = p - resolution after getKeyBitSize).
    When that coordinate is a rounding residue like -5e-17 (a cloud rotated by 270 degrees), the double sum p - min
    swallows ~1e28 float steps around the face: the face threshold has to be searched over the whole float line
    (found by the BASELINE config 4 test: tile 6 came out 229 points short)."""
    pts = oracle.empty(3)
    pts['x'], pts['y'] = -0.27, 0.001
    pts['z'] = [-4.9759e-17, -0.005, 0.003]
    got, exp = check_downsample(gpu, oracle, pts, 0.0, 0.01)
    assert len(got) == 3          # voxel z = -1 is cut by the face at z ~ 0: one output on either side
    for first in (-4.9759e-17, 4.9759e-17, -1e-30, 1e-30, -0.0, -1.4e-45):
        for other in (-0.005, -2e-17, 2e-17, -1e-40, 0.0):
            pts['z'] = [first, other, 0.003]
            check_downsample(gpu, oracle, pts, 0.0, 0.01)
    from cwipc_util_amd.capture import rotation_about_y
    base, cs = synth(100000)
    for quarter in (1, 2, 3):
        check_downsample(gpu, oracle, oracle.transform(base, rotation_about_y(quarter * np.pi / 2)), cs, 0.01)


def test_downsample_points_on_voxel_faces(gpu, oracle):
    """Coordinates that are exact multiples of the cell size: fl(x * inv_leaf) puts them a hair
    below their voxel's origin, i.e. the offset inside the voxel is slightly negative."""
    rng = np.random.default_rng(21)
    n = 60000
    pts = oracle.empty(n)
    for f, base in (('x', -25.0), ('y', 3.0), ('z', 0.0)):
        pts[f] = (base + rng.integers(-40, 40, n) * 0.01).astype(np.float32)
    pts['r'], pts['g'], pts['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    pts['tile'] = 1 << rng.integers(0, 4, n)
    for cell in (0.01, -0.01, 0.02):
        check_downsample(gpu, oracle, pts, 0.0, cell)


def test_downsample_far_from_origin(gpu, oracle):
    """More than 4e6 voxels from the origin the fixed-point scale of the accumulators is widened (second pass)."""
    rng = np.random.default_rng(22)
    n = 50000
    pts = oracle.empty(n)
    pts['x'] = (50000.0 + rng.random(n) * 0.5).astype(np.float32)
    pts['y'] = (-70000.0 + rng.random(n) * 0.5).astype(np.float32)
    pts['z'] = (rng.random(n) * 0.5).astype(np.float32)
    pts['r'], pts['g'], pts['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    pts['tile'] = 1 << rng.integers(0, 4, n)
    for cell in (0.01, -0.01):
        pc = make_cloud(gpu, pts, 0.0, 1)
        got = gpu.cwipc_downsample(pc, cell).get_numpy_array()
        exp, _ = oracle.downsample(pts, 0.0, cell)
        assert len(got) == len(exp)
        for f in ('x', 'y', 'z'):
            # the oracle's fp32 running sums are good to a few ulp out here (ulp = 0.004 at 5e4)
            ulp = np.spacing(np.abs(exp[f]).astype(np.float32)).astype(np.float64)
            assert (np.abs(got[f].astype(np.float64) - exp[f]) <= np.maximum(4 * ulp, XYZ_TOL)).all(), f
        for f in ('r', 'g', 'b', 'tile'):
            assert (got[f] == exp[f]).all(), f


def test_downsample_random_volume(gpu, oracle):
    rng = np.random.default_rng(11)
    n = 200000
    pts = oracle.empty(n)
    pts['x'], pts['y'], pts['z'] = (rng.random(n) * 4 - 2), (rng.random(n) * 3), (rng.random(n) * 5 - 1)
    pts['r'], pts['g'], pts['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    pts['tile'] = 1 << rng.integers(0, 8, n)
    for cell in (0.05, -0.05, 0.25):
        check_downsample(gpu, oracle, pts, 0.0, cell)


def test_downsample_many_leaves_in_random_order(gpu, oracle):
    """Incoherent input over ~125 octree leaves: every workgroup meets more leaves than its local leaf table
    names (second pass with global leaf ids), and more than the initial number of leaf grids (workspace regrown)."""
    rng = np.random.default_rng(33)
    n = 150000
    pts = oracle.empty(n)
    pts['x'], pts['y'], pts['z'] = rng.random(n) * 3.0 - 1.5, rng.random(n) * 3.0, rng.random(n) * 3.0 - 1.0
    pts['r'], pts['g'], pts['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    pts['tile'] = 1 << rng.integers(0, 8, n)
    check_downsample(gpu, oracle, pts, 0.0, 0.01)
    check_downsample(gpu, oracle, pts, 0.0, -0.01)


def test_downsample_wide_scanlines(gpu, oracle):
    """A depth-camera like cloud: row-major scan lines 3 m wide, points about a voxel apart, so a wave's
    step of 256 points spans far more than the 128 voxels one pair of cached leaf faces covers."""
    rng = np.random.default_rng(44)
    w, h = 640, 360
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    n = w * h
    pts = oracle.empty(n)
    pts['x'] = (u.ravel() * (3.0 / w) - 1.5 + rng.normal(0, 0.001, n)).astype(np.float32)
    pts['y'] = (2.0 - v.ravel() * (2.0 / h) + rng.normal(0, 0.001, n)).astype(np.float32)
    pts['z'] = (1.5 + 0.4 * np.sin(u.ravel() * 0.01) + rng.normal(0, 0.002, n)).astype(np.float32)
    pts['r'], pts['g'], pts['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    pts['tile'] = 1 << rng.integers(0, 3, n)
    for cell in (0.01, 0.005, -0.01):
        check_downsample(gpu, oracle, pts, 0.0, cell)


def test_downsample_plain_grid_sort_path(gpu, oracle, synth, monkeypatch):
    """The plain grid orders its output through a bitmap over the VoxelGrid index space; index spaces
    beyond 2^28 cells are sorted instead.  Force that path (it needs tens of GB of bricks otherwise)."""
    pts, cs = synth(300000)
    a, _ = check_downsample(gpu, oracle, pts, cs, -0.01)
    monkeypatch.setenv("CWIPC_GRID_BITMAP_MAX", "0")
    b, _ = check_downsample(gpu, oracle, pts, cs, -0.01)
    assert same(a, b)


@pytest.mark.parametrize("seed", range(120))
def test_downsample_random_configurations(gpu, oracle, seed):
    """Differential test over random shapes, densities, orders, offsets and cell sizes (rare paths of the voxel
    kernel: face-cache refills, per-point face lookups, lanes with many runs, table overflow, workspace regrowth)."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 7, 64, 255, 257, 1000, 5000, 20000, 60000]))
    kind = rng.choice(["box", "clusters", "line", "sheet", "lattice"])
    scale = float(rng.choice([0.05, 0.3, 1.0, 3.0]))
    if kind == "box":
        xyz = rng.random((n, 3)) * scale
    elif kind == "clusters":
        centres = rng.random((8, 3)) * scale
        xyz = centres[rng.integers(0, 8, n)] + rng.normal(0, scale * 0.01, (n, 3))
    elif kind == "line":
        t = np.sort(rng.random(n))
        xyz = np.stack([t * scale, 0.3 * np.sin(t * 40) * scale, t * t * scale], axis=1) + rng.normal(0, 1e-4, (n, 3))
    elif kind == "sheet":
        u, v = rng.random(n), rng.random(n)
        xyz = np.stack([u * scale, v * scale, 0.1 * scale * np.sin(6 * u) * np.cos(5 * v)], axis=1)
    else:   # points exactly on a lattice that divides the cell size (voxel faces)
        xyz = rng.integers(0, 40, (n, 3)) * 0.005
    xyz += rng.choice([0.0, 0.0, 10.0, -37.5]) * rng.random(3)
    if rng.random() < 0.5:
        xyz = xyz[np.lexsort((xyz[:, 0], xyz[:, 1]))]   # scan order: rows along x
    pts = oracle.empty(n)
    pts['x'], pts['y'], pts['z'] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts['r'], pts['g'], pts['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    pts['tile'] = 1 << rng.integers(0, 8, n)
    if n > 10 and rng.random() < 0.3:
        bad = rng.integers(1, n, max(1, n // 50))   # not the first point: the octree's anchor must be finite
        pts['x'][bad] = np.nan
        pts['z'][bad[::2]] = np.inf
    cell = float(rng.choice([0.003, 0.01, 0.02, 0.05, 0.2]))
    cell = max(cell, scale / 400)   # keeps the leaf count (20 MB of records each) in bounds
    check_downsample(gpu, oracle, pts, 0.0, cell)
    if not np.isnan(pts['x']).any():
        check_downsample(gpu, oracle, pts, 0.0, -cell)


def test_downsample_non_finite_points(gpu, oracle, synth):
    """The octree skips non-finite points, so its lattice is anchored at the first FINITE point; a cloud
    without any gives an empty result (no leaves), not an error."""
    pts, cs = synth(20000)
    p = pts.copy()
    p['x'][0] = np.nan
    p['y'][1] = np.inf
    p['z'][2] = -np.inf
    p['x'][5000:5010] = np.nan
    got, exp = check_downsample(gpu, oracle, p, cs, 0.02)
    ref, _ = oracle.downsample(pts[3:][np.isfinite(p['x'][3:])], cs, 0.02)
    assert same(exp, ref)
    late = pts.copy()
    late['x'][:3000] = np.nan            # the first finite point lies beyond the first chunks
    check_downsample(gpu, oracle, late, cs, 0.02)
    none = pts[:100].copy()
    none['y'] = np.nan
    assert gpu.cwipc_downsample(make_cloud(gpu, none, cs), 0.02).count() == 0 == len(oracle.downsample(none, cs, 0.02)[0])


def test_downsample_single_point_and_duplicates(gpu, oracle):
    pts = oracle.empty(1)
    pts['x'], pts['y'], pts['z'], pts['r'], pts['tile'] = 0.5, -0.25, 3.0, 200, 4
    check_downsample(gpu, oracle, pts, 0.0, 0.1)
    check_downsample(gpu, oracle, pts, 0.0, -0.1)
    many = np.repeat(pts, 5000)
    got, _ = check_downsample(gpu, oracle, many, 0.0, 0.1)
    assert len(got) == 1 and got['r'][0] == 200 and got['tile'][0] == 4


def test_downsample_is_reproducible(gpu, synth):
    """Integer accumulation: two runs give bit-identical results, and the workspace is left clean."""
    pts, cs = synth(300000)
    pc = make_cloud(gpu, pts, cs)
    a = gpu.cwipc_downsample(pc, 0.01).get_numpy_array()
    b = gpu.cwipc_downsample(pc, 0.01).get_numpy_array()
    c = gpu.cwipc_downsample(pc, 0.02).get_numpy_array()
    d = gpu.cwipc_downsample(pc, 0.01).get_numpy_array()
    assert same(a, b) and same(a, d) and len(c) < len(a)


def test_downsample_stream_of_frames_returns_early(gpu, oracle, synth):
    """In a stream of frames (the same kind of cloud again and again) cwipc_downsample hands its result out while its kernels
    still run; the cloud settles when somebody asks for its points.  Every frame must be what a lone call gives, whatever
    is asked first, however many results are pending, and also when a frame does not look like the ones before (far more
    voxels than the result was sized for: the pass is run again behind the scenes)."""
    import gc
    pts, cs = synth(300000)
    want = gpu.cwipc_downsample(make_cloud(gpu, pts, cs, 5), 0.01).get_numpy_array()
    exp, _ = oracle.downsample(pts, cs, 0.01)
    assert len(want) == len(exp)
    pc = make_cloud(gpu, pts, cs, 5)
    gpu.cwipc_hip_upload(pc, drop_host_copy=True)
    frames = [gpu.cwipc_downsample(pc, 0.01) for _ in range(12)]          # nothing asked in between: up to two in flight
    assert same(frames[7].get_numpy_array(), want)                          # out of order
    assert [f.count() for f in frames] == [len(want)] * 12
    assert same(frames[11].get_numpy_array(), want) and frames[3].timestamp() == 5
    chained = gpu.cwipc_tilefilter(gpu.cwipc_downsample(pc, 0.01), 1)      # a pending result as the next filter's input
    assert same(chained.get_numpy_array(), want[want['tile'] == 1])
    for _ in range(6):                                                      # results dropped without ever being looked at
        gpu.cwipc_downsample(pc, 0.01)
    gc.collect()
    assert same(gpu.cwipc_downsample(pc, 0.01).get_numpy_array(), want)
    # the same number of points, spread over eight times the volume: the result sized from the frames before is too small
    rng = np.random.default_rng(11)
    wide = pts.copy()
    wide['x'] *= 2.0; wide['y'] *= 2.0; wide['z'] *= 2.0
    wide_exp, _ = oracle.downsample(wide, cs, 0.01)
    assert len(wide_exp) > 2 * len(want)
    stream = [gpu.cwipc_downsample(pc, 0.01) for _ in range(4)]
    odd = gpu.cwipc_downsample(make_cloud(gpu, wide, cs, 6), 0.01)
    after = [gpu.cwipc_downsample(pc, 0.01) for _ in range(4)]
    got = odd.get_numpy_array()
    assert len(got) == len(wide_exp) and np.abs(got['x'].astype(np.float64) - wide_exp['x']).max() <= XYZ_TOL
    assert all(same(f.get_numpy_array(), want) for f in stream + after)
    assert gpu.cwipc_dangling_allocations(False) >= 0


def test_pending_results_outlive_their_threads(gpu, oracle, synth):
    """A downsample in a stream of calls is handed out while its kernels run; the report it settles on lies in page-locked words
    of the calling thread's workspace.  Threads that have ended leave at most eight workspaces for the next threads -- the rest is
    given back to the device, words included (round 3).  A result that outlives its thread, and its thread's workspace, must still
    settle (round 3's review: the report was read from freed memory): twelve threads downsample once each -- their workspaces come
    from the pool with a streak of good passes, so the call returns early -- hand the result over and end; then the counts are asked for."""
    import threading
    pts, cs = synth(300000, 0.2)
    exp, _ = oracle.downsample(pts, cs, 0.01)
    pc = make_cloud(gpu, pts, cs, 3)
    gpu.cwipc_hip_upload(pc)

    def warm():
        for _ in range(5):
            gpu.cwipc_downsample(pc, 0.01).count()
    for _ in range(2):                       # two generations of threads: the pool holds workspaces that have seen this kind of call
        ws = [threading.Thread(target=warm) for _ in range(12)]
        for t in ws: t.start()
        for t in ws: t.join()
    results = [None] * 12

    def once(i):
        results[i] = gpu.cwipc_downsample(pc, 0.01)

    ts = [threading.Thread(target=once, args=(i,)) for i in range(12)]
    for t in ts: t.start()
    for t in ts: t.join()
    gpu.util.cwipc_util_dll_load().cwipc_hip_synchronize()
    for r in results:
        assert r.count() == len(exp)
        got = r.get_numpy_array()
        assert (got['tile'] == exp['tile']).all() and np.abs(got['x'].astype(np.float64) - exp['x']).max() <= XYZ_TOL


def test_fast_and_general_accumulate_kernels_agree(gpu, synth):
    """The lean accumulate kernel and the general one share one quantisation: bit-identical clouds (the general one is forced
    through CWIPC_VOXEL_GENERAL in a process of its own)."""
    import subprocess, sys, tempfile
    pts, cs = synth(300000, 0.3)
    fast = {c: gpu.cwipc_downsample(make_cloud(gpu, pts, cs), c).get_numpy_array() for c in (0.01, -0.01, 0.004)}
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "pts.npy"), pts)
        code = (
            "import sys, numpy as np, torch\n"
            "sys.path.insert(0, %r)\n"
            "import cwipc_util_amd as cw\n"
            "pts = np.load(%r)\n"
            "for c in (0.01, -0.01, 0.004):\n"
            "    pc = cw.cwipc_from_numpy_array(pts, 1); pc._set_cellsize(%r)\n"
            "    np.save(%r %% c, cw.cwipc_downsample(pc, c).get_numpy_array())\n"
        ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(tmp, "pts.npy"), cs, os.path.join(tmp, "out_%s.npy"))
        env = dict(os.environ, CWIPC_VOXEL_GENERAL="1")
        subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=600)
        for c, got in fast.items():
            assert same(got, np.load(os.path.join(tmp, "out_%s.npy" % c))), c


@pytest.mark.parametrize("knob", ["CWIPC_DEFER=0", "CWIPC_VOXEL_PARTITION=0", "CWIPC_SOR_HOST_GRID=1", "CWIPC_SYNTHETIC_HOST=1", "CWIPC_POLL_US=0",
                                  "CWIPC_K1_DUMP=1", "CWIPC_K1_DUMP=2", "CWIPC_K1_PAIR=1", "CWIPC_WORKSPACES=1", "CWIPC_WORKSPACES=4", "CWIPC_SOR_SMALL_CELLS=0", "CWIPC_SOR_PAIR=0", "CWIPC_SOR_STATS_FOLD=0"])   # (+ CWIPC_PINNED_UPLOAD=kernel: test_page_locked_buffers_both_ways)
def test_variant_knobs_change_no_result(gpu, synth, knob, tmp_path):
    """Every environment knob of the shipped library selects another way to the same result: a
    process with the knob set must produce, bit for bit, what this process produces -- a stream of downsample calls (the
    early return), a stream on a shuffled cloud (the partition pass), outlier removal (where the grid is decided), the
    synthetic source (where it is generated)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pts, cs = synth(200000, 0.1)
    np.save(str(tmp_path / "pts.npy"), pts)
    code = (
        "import sys, struct, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import cwipc_util_amd as cw\n"
        "pts = np.load(%r); out = {}\n"
        "pc = cw.cwipc_from_numpy_array(pts, 1); pc._set_cellsize(%r)\n"
        "outs = [cw.cwipc_downsample(pc, 0.01) for i in range(14)]\n"
        "for i, o in enumerate(outs): out['down%%d' %% i] = o.get_numpy_array()\n"
        "perm = pts[np.random.default_rng(3).permutation(len(pts))]\n"
        "pp = cw.cwipc_from_numpy_array(perm, 1); pp._set_cellsize(%r)\n"
        "for i in range(6): out['perm%%d' %% i] = cw.cwipc_downsample(pp, 0.01).get_numpy_array()\n"
        "out['sor'] = cw.cwipc_remove_outliers(cw.cwipc_from_numpy_array(pts[:50000], 1), 16, 1.0, False).get_numpy_array()\n"
        "src = cw.cwipc_synthetic(0, 100000); src.start(); b = bytearray(4)\n"
        "assert src.auxiliary_operation('amd-fixangle', struct.pack('f', 0.5), b)\n"
        "out['synth'] = src.get().get_numpy_array(); src.stop()\n"
        "np.savez(sys.argv[1], **out)\n"
    ) % (root, str(tmp_path / "pts.npy"), cs, cs)
    ours, theirs = str(tmp_path / "default.npz"), str(tmp_path / "knob.npz")
    name, value = knob.split("=")
    subprocess.run([sys.executable, "-c", code, ours], check=True, timeout=600, env={k: v for k, v in os.environ.items() if k != name})
    subprocess.run([sys.executable, "-c", code, theirs], check=True, timeout=600, env=dict(os.environ, **{name: value}))
    a, b = np.load(ours), np.load(theirs)
    assert sorted(a.keys()) == sorted(b.keys())
    for key in a.keys():
        assert same(a[key], b[key]), (knob, key)


@pytest.mark.parametrize("env", [{"CWIPC_K1_STAGGER": "0"}, {"CWIPC_K1_STAGGER": "40"}, {"CWIPC_K1_STAGGER": "25", "CWIPC_K1_STAGGER_REV": "1"}])
def test_staggered_ranges_change_no_result(gpu, synth, env, tmp_path):
    """The fast accumulate kernel's workgroups take ranges of growing length (r4: 0.75 to 1.25 of the mean, so that their flushes do
    not arrive together; clouds from 1.5 M points).  Equal ranges (rounds 1-3), a steeper slope and the reverse order give, bit for
    bit, the same clouds: integer sums do not care where a range ends, and the replay kernel reads the ranges the accumulate kernel
    took (range_first_step, kernels_voxel.hip)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pts, cs = synth(2_500_000, 0.3)
    np.save(str(tmp_path / "pts.npy"), pts)
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import cwipc_util_amd as cw\n"
        "pts = np.load(%r); out = {}\n"
        "pc = cw.cwipc_from_numpy_array(pts, 1); pc._set_cellsize(%r)\n"
        "for c in (0.01, -0.01, 0.003):\n"
        "    for i in range(4): out['c%%s_%%d' %% (c, i)] = cw.cwipc_downsample(pc, c).get_numpy_array()\n"
        "np.savez(sys.argv[1], **out)\n"
    ) % (root, str(tmp_path / "pts.npy"), cs)
    ours, theirs = str(tmp_path / "default.npz"), str(tmp_path / "knob.npz")
    subprocess.run([sys.executable, "-c", code, ours], check=True, timeout=600, env={k: v for k, v in os.environ.items() if not k.startswith("CWIPC_K1_STAGGER")})
    subprocess.run([sys.executable, "-c", code, theirs], check=True, timeout=600, env=dict(os.environ, **env))
    a, b = np.load(ours), np.load(theirs)
    assert sorted(a.keys()) == sorted(b.keys())
    for key in a.keys():
        assert len(a[key]) > 1000 and same(a[key], b[key]), (env, key)


def test_sor_sparse_and_dense_grid_layouts_agree(gpu, oracle, synth):
    """The k-NN grid has two layouts: dense (small clouds) and segments of 16 cells that exist only where points are (big
    clouds, kernels_sor.hip).  d_i is a property of the cloud, not of the search structure: both layouts, forced through
    CWIPC_SOR_SPARSE in processes of their own, must give the oracle's d_i bit for bit -- on the synthetic figure, on a thin
    wide cloud whose rows cross many empty segments, on a cloud with far outliers (empty space between), with non-finite
    points, and for k beyond 16 (the 33-slot variant)."""
    import subprocess, sys, tempfile
    rng = np.random.default_rng(11)
    clouds = {}
    pts, cs = synth(60000, 0.2)
    clouds["figure"] = pts
    wide = oracle.empty(20000)
    wide['x'] = rng.random(20000) * 40 - 20; wide['y'] = rng.random(20000) * 0.02; wide['z'] = rng.random(20000) * 0.5
    clouds["wide"] = wide
    far = pts[:15000].copy()
    far['x'][::500] += 50.0
    far['z'][250::500] -= 30.0
    clouds["far"] = far
    bad = pts[:15000].copy()
    bad['x'][100] = np.nan; bad['y'][5000] = np.inf
    clouds["bad"] = bad
    with tempfile.TemporaryDirectory() as tmp:
        for name, p in clouds.items():
            np.save(os.path.join(tmp, name + ".npy"), p)
        code = (
            "import sys, numpy as np, torch\n"
            "sys.path.insert(0, %r)\n"
            "import cwipc_util_amd as cw\n"
            "for name in %r:\n"
            "    pts = np.load(%r %% name)\n"
            "    for k in (16, 5, 24):\n"
            "        d, _ = cw.cwipc_hip_knn_mean_dist(cw.cwipc_from_numpy_array(pts, 1), k, 1.0)\n"
            "        np.save(%r %% (name, k, sys.argv[1]), d)\n"
        ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), sorted(clouds), os.path.join(tmp, "%s.npy"), os.path.join(tmp, "d_%s_%d_%s.npy"))
        for layout in ("0", "1"):
            subprocess.run([sys.executable, "-c", code, layout], check=True, env=dict(os.environ, CWIPC_SOR_SPARSE=layout), timeout=600)
        for name, p in clouds.items():
            for k in (16, 5, 24):
                dense, sparse = (np.load(os.path.join(tmp, "d_%s_%d_%s.npy" % (name, k, l))) for l in ("0", "1"))
                fin = np.isfinite(p['x']) & np.isfinite(p['y']) & np.isfinite(p['z'])
                assert (dense[fin].view(np.uint32) == sparse[fin].view(np.uint32)).all(), (name, k)
                if name != "bad":
                    want = oracle.knn_mean_dist(p, k)
                    assert (sparse.view(np.uint32) == want.view(np.uint32)).all(), (name, k)


def test_downsample_grid_overflow_is_an_error(gpu, oracle):
    """pcl::VoxelGrid refuses grids of more than 2^31 cells; the reference then returns NULL."""
    pts = oracle.empty(2)
    pts['x'] = [0, 100]; pts['y'] = [0, 100]; pts['z'] = [0, 100]
    with pytest.raises(oracle.OracleError):
        oracle.downsample(pts, 0.0, -0.01)
    with pytest.raises(gpu.CwipcError):
        gpu.cwipc_downsample(make_cloud(gpu, pts), -0.01)
    # the octree-split path exists precisely to handle this
    check_downsample(gpu, oracle, pts, 0.0, 0.01)


def test_downsample_pertile_chain(gpu, oracle, synth):
    """cwipc_downsample_pertile (reference python/cwipc/registration/util.py:170-182): tilefilter -> downsample -> join."""
    pts, cs = synth(300000)
    pc = make_cloud(gpu, pts, cs)
    result, expect = None, None
    for tile in (1, 2):
        t = gpu.cwipc_downsample(gpu.cwipc_tilefilter(pc, tile), 0.01)
        e, _ = oracle.downsample(oracle.tilefilter(pts, tile), cs, 0.01)
        result = t if result is None else gpu.cwipc_join(result, t)
        expect = e if expect is None else oracle.join(expect, e)
    got = result.get_numpy_array()
    assert len(got) == len(expect)
    assert (got['tile'] == expect['tile']).all()
    assert np.abs(got['x'].astype(np.float64) - expect['x']).max() <= XYZ_TOL


# ---------------------------------------------------------------------------
# statistical outlier removal
# ---------------------------------------------------------------------------
def check_sor(gpu, oracle, pts, cs, k, mul):
    pc = make_cloud(gpu, pts, cs, 31)
    exp, d_exp, thr_exp = oracle.remove_outliers(pts, k, mul, False, want_stats=True)
    d_got, thr_got = gpu.cwipc_hip_knn_mean_dist(pc, k, mul)
    assert (d_got == d_exp).all(), f"d_i differs at {np.flatnonzero(d_got != d_exp)[:5]}"
    assert thr_got == pytest.approx(thr_exp, rel=1e-12)
    out = gpu.cwipc_remove_outliers(pc, k, mul, False)
    got = out.get_numpy_array()
    assert out.timestamp() == 31 and out.cellsize() == pc.cellsize()
    # d_i is bit-identical, so the HIP mask is fully determined by its own threshold ...
    assert same(got, pts[~(d_got.astype(np.float64) > thr_got)])
    if not same(got, exp):
        # ... and may differ from the oracle's only for points sitting on the threshold
        # (the f64 mean/variance are summed in a different order)
        band = np.abs(d_exp.astype(np.float64) - thr_exp) <= 1e-6 * abs(thr_exp)
        assert abs(len(got) - len(exp)) <= band.sum()
    return got, exp


@pytest.mark.parametrize("npoints,k,mul", [(20000, 16, 1.0), (100000, 16, 1.0), (100000, 30, 1.5), (300000, 16, 1.0), (5000, 4, 0.5)])
def test_remove_outliers_synthetic(gpu, oracle, synth, npoints, k, mul):
    pts, cs = synth(npoints)
    got, exp = check_sor(gpu, oracle, pts, cs, k, mul)
    assert 0 < len(got) < len(pts)


@pytest.mark.parametrize("npoints,k", [(20000, 100), (20000, 200), (6000, 400)])
def test_remove_outliers_any_k(gpu, oracle, synth, npoints, k):
    """The reference takes any kNeighbors (src/cwipc_filters.cpp:197-201 hands it to pcl unchecked); round 2 stopped at 120.
    k = 100: lists in 64 KB of LDS; 200: in LDS above the default limit; 400: in a slab of device memory per workgroup."""
    pts, cs = synth(npoints, 0.3)
    got, exp = check_sor(gpu, oracle, pts, cs, k, 1.0)
    assert 0 < len(got) < len(pts)


def test_remove_outliers_random_cloud_with_outliers(gpu, oracle):
    rng = np.random.default_rng(9)
    n = 50000
    pts = oracle.empty(n)
    pts['x'], pts['y'], pts['z'] = rng.normal(0, 0.2, n), rng.normal(1, 0.3, n), rng.normal(0, 0.05, n)
    far = rng.integers(0, n, 200)
    pts['x'][far] += rng.normal(0, 3, 200).astype(np.float32)
    pts['tile'] = 1
    got, exp = check_sor(gpu, oracle, pts, 0.0, 16, 1.0)
    assert len(got) < n


def test_remove_outliers_pertile(gpu, oracle, synth):
    """reference test_remove_outliers: (30, 1.0, perTile=True) keeps 0 < n < N points."""
    pts, cs = synth(100000)
    out = gpu.cwipc_remove_outliers(make_cloud(gpu, pts, cs), 30, 1.0, True)
    exp = oracle.remove_outliers(pts, 30, 1.0, True)
    got = out.get_numpy_array()
    assert 0 < len(got) < len(pts)
    assert abs(len(got) - len(exp)) <= 2
    if len(got) == len(exp):
        assert same(got, exp)


def test_remove_outliers_tiny_clouds(gpu, oracle):
    # fewer points than neighbours: the oracle DEFINES the missing distances as 0 (upstream reads past its arrays)
    for n in (1, 2, 5, 17):
        pts = oracle.empty(n)
        pts['x'] = np.arange(n) * 0.1
        got = gpu.cwipc_remove_outliers(make_cloud(gpu, pts), 16, 1.0, False).get_numpy_array()
        assert same(got, oracle.remove_outliers(pts, 16, 1.0, False)), n


@pytest.mark.parametrize("n", [65535, 65536, 65537, 150000])
@pytest.mark.parametrize("kind", ["sheet", "box", "edge"])
def test_remove_outliers_either_side_of_the_small_flow(gpu, oracle, n, kind):
    """Clouds up to 65 536 points (8 cells per point = 2^19 cells) take the ten-launch flow of round 4 (kernels_sor.hip,
    sor_small_on_device: the grid derived by every workgroup of the count kernels, a one-workgroup scan), bigger ones the twelve-launch
    flow; both search with the k-NN kernel whose shells beyond the first are bounded row by row.  d_i bit for bit on both sides of the
    line: a sheet (the grid is coarsened after the census), a box (it is not: ~1 point per cell of the finest grid is already volume-like),
    and a thin strip whose every query is an edge query (second and third shells)."""
    if kind == "box" and n not in (65536, 65537):
        pytest.skip("the oracle's search through a volume takes ten seconds and more at these sizes: the two sizes at the line only")
    rng = np.random.default_rng(n + len(kind))
    if kind == "sheet":
        xyz = np.stack([rng.random(n) * 2.0, rng.random(n) * 1.0, 0.05 * np.sin(rng.random(n) * 6.0)], axis=1)
    elif kind == "box":
        xyz = rng.random((n, 3)) * 1.5
    else:
        xyz = np.stack([rng.random(n) * 40.0, rng.random(n) * 0.01, np.zeros(n)], axis=1)
    pts = oracle.empty(n)
    pts['x'], pts['y'], pts['z'] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    for k in ((16,) if kind == "box" else (16, 30)):
        d_got, _ = gpu.cwipc_hip_knn_mean_dist(make_cloud(gpu, pts), k, 1.0)
        d_exp = oracle.knn_mean_dist(pts, k)
        assert (d_got == d_exp).all(), (n, kind, k, np.flatnonzero(d_got != d_exp)[:5])


@pytest.mark.parametrize("kind", ["dupes", "line", "clusters", "few", "lattice"])
def test_remove_outliers_two_lanes_per_query_shapes(gpu, oracle, kind):
    """k = 16 on small clouds runs the k-NN kernel with two lanes per query (each lane every second candidate, the pair's nearest as a set):
    shapes that stress what is special about it -- more coincident points than neighbours (every distance zero, nothing to tell the smallest
    by), all candidates in one row of cells (a line), lists that never fill (fewer points than k + 1 in reach), ties at the (k + 1)-th
    distance (a lattice: rows and rings are turned away on equal bounds) -- against the oracle's d_i, bit for bit."""
    rng = np.random.default_rng(len(kind) * 7 + 1)
    if kind == "dupes":
        base = rng.random((40, 3))
        xyz = base[rng.integers(0, 40, 3000)]                     # ~75 copies of each of 40 places
    elif kind == "line":
        xyz = np.stack([rng.random(5000) * 3.0, np.full(5000, 0.5), np.full(5000, -0.25)], axis=1)
    elif kind == "clusters":
        centres = rng.random((7, 3)) * 4.0
        xyz = centres[rng.integers(0, 7, 6000)] + rng.normal(0, 0.002, (6000, 3))
    elif kind == "few":
        xyz = rng.random((19, 3))
    else:
        g = np.arange(18, dtype=np.float64) * 0.125               # exactly representable: many equal distances
        xyz = np.stack(np.meshgrid(g, g, g[:9], indexing="ij"), axis=-1).reshape(-1, 3)
    pts = oracle.empty(len(xyz))
    pts['x'], pts['y'], pts['z'] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    d_got, _ = gpu.cwipc_hip_knn_mean_dist(make_cloud(gpu, pts), 16, 1.0)
    d_exp = oracle.knn_mean_dist(pts, 16)
    assert (d_got == d_exp).all(), (kind, np.flatnonzero(d_got != d_exp)[:5])


def test_remove_outliers_distances_spread_over_many_binades(gpu, oracle):
    """The two-lanes-per-query k-NN kernel (small clouds, k = 16) sums a query's sixteen distances as a set, which is exact in f64 in any
    order while they lie within 2^23 of each other; a query whose neighbours are partly a nanometre and partly a centimetre away takes the
    sorted sum.  Both give the oracle's d_i bit for bit."""
    rng = np.random.default_rng(77)
    centres = np.array([[i, j, l] for i in range(2) for j in range(2) for l in range(3)], dtype=np.float64) * 0.01
    xyz = (centres[:, None, :] + rng.random((len(centres), 10, 3)) * 3e-9).reshape(-1, 3)
    xyz = np.concatenate([xyz, rng.random((400, 3)) * 0.02 + 0.05])   # ... next to ordinary queries in the same waves
    pts = oracle.empty(len(xyz))
    pts['x'], pts['y'], pts['z'] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    d_exp = oracle.knn_mean_dist(pts, 16)
    near = np.sort(np.linalg.norm(xyz[:1] - xyz[1:10], axis=1))
    assert near[0] > 0 and 0.009 / near[0] > 2 ** 23      # the case is what it says
    d_got, _ = gpu.cwipc_hip_knn_mean_dist(make_cloud(gpu, pts), 16, 1.0)
    assert (d_got == d_exp).all(), np.flatnonzero(d_got != d_exp)[:5]
    check_sor(gpu, oracle, pts, 0.001, 16, 1.0)


@pytest.mark.parametrize("seed", range(40))
def test_remove_outliers_random_configurations(gpu, oracle, seed):
    """Differential test of the k-NN statistics (d_i bit for bit) over random shapes, densities and k:
    exercises the register list (k <= 16, k <= 32), the LDS list (k > 32), searches that need many rings,
    coincident points, grids that degenerate to a line or a plane."""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([40, 200, 1000, 5000, 30000]))
    k = int(rng.choice([1, 4, 16, 17, 30, 32, 33, 50]))
    if n <= k:
        n = k + 7
    kind = rng.choice(["box", "clusters", "line", "sheet", "dupes"])
    scale = float(rng.choice([0.1, 1.0, 25.0]))
    if kind == "box":
        xyz = rng.random((n, 3)) * scale
    elif kind == "clusters":
        centres = rng.random((5, 3)) * scale
        xyz = centres[rng.integers(0, 5, n)] + rng.normal(0, scale * 0.003, (n, 3))
    elif kind == "line":
        t = rng.random(n)
        xyz = np.stack([t * scale, np.zeros(n), np.zeros(n)], axis=1)
    elif kind == "sheet":
        xyz = np.stack([rng.random(n) * scale, rng.random(n) * scale, np.full(n, 0.25)], axis=1)
    else:   # many coincident points
        base = rng.random((max(n // 10, 1), 3)) * scale
        xyz = base[rng.integers(0, len(base), n)]
    xyz += rng.choice([0.0, -3.0, 100.0])
    pts = oracle.empty(n)
    pts['x'], pts['y'], pts['z'] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts['tile'] = 1 << rng.integers(0, 3, n)
    check_sor(gpu, oracle, pts, 0.0, k, float(rng.choice([0.5, 1.0, 2.0])))


# ---------------------------------------------------------------------------
# chains and residency
# ---------------------------------------------------------------------------
def test_full_chain_stays_on_device(gpu, oracle, synth):
    """colorize -> downsample -> remove_outliers -> join (BASELINE config 5's per-frame chain)."""
    from cwipc_util_amd.filters import factory
    pts, cs = synth(300000, 0.3)
    pc = make_cloud(gpu, pts, cs)
    chain = [factory('colorize(0.8, "camera")'), factory('voxelize(0.01)'), factory('remove_outliers(16, 1.0, False)')]
    cur = pc
    for f in chain:
        cur = f.filter(cur)
        assert gpu.util.cwipc_util_dll_load().cwipc_hip_is_device_resident(cur.as_cwipc_p()) == 1
    from cwipc_util_amd.filters.colorize import ColorizeFilter
    lut, valid = ColorizeFilter(0.8, "camera").colorMap.tables()
    e = oracle.colorize(pts, 0.8, lut, valid)
    e, ecs = oracle.downsample(e, cs, 0.01)
    # the outlier filter runs on the oracle's voxel centroids, which differ from ours in the last bits: compare counts loosely
    e2 = oracle.remove_outliers(e, 16, 1.0, False)
    assert abs(cur.count() - len(e2)) <= max(3, len(e2) // 500)
    j = gpu.cwipc_join(cur, cur)
    assert j.count() == 2 * cur.count()
    for f in chain:
        f.statistics()


def test_profile_records_kernels(gpu, synth):
    pts, cs = synth(100000)
    pc = make_cloud(gpu, pts, cs)
    with gpu.cwipc_hip_profile() as prof:
        gpu.cwipc_downsample(pc, 0.01)
    # (the fast accumulate kernel, or the general one when the suite runs with CWIPC_VOXEL_GENERAL=1)
    name = "voxel_accumulate" if "voxel_accumulate" in prof.kernels else "voxel_accumulate_general"
    assert name in prof.kernels and prof.kernels[name][1] == 1
    assert prof.kernels[name][0] > 0


# ---------------------------------------------------------------------------
# BASELINE.json full sizes (10 M points): direct comparison (the oracle needs ~1 s for the voxel
# grid and ~40 s for the outlier filter) plus size-independent properties
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_cloud(oracle):
    return oracle.synthetic(10_000_000, 0.0)


def test_full_size_downsample(gpu, oracle, full_cloud):
    pts, cs = full_cloud
    assert len(pts) == 9998244
    got, exp = check_downsample(gpu, oracle, pts, cs, 0.01)
    assert len(got) == 39548
    assert np.bitwise_or.reduce(got['tile']) == np.bitwise_or.reduce(pts['tile'])
    # plain grid: besides the comparison, the result is a fixed point of the filter (up to the final
    # fp32 rounding of a centroid) and every centroid lies inside the cloud's box.  (Not so on the
    # octree path: a second pass anchors its leaves at another first point and cuts other voxels.)
    got, exp = check_downsample(gpu, oracle, pts, cs, -0.01)
    assert len(got) == 39312
    again = gpu.cwipc_downsample(make_cloud(gpu, got, 0.01), -0.01).get_numpy_array()
    assert len(again) == len(got)
    for f in ('x', 'y', 'z'):
        assert np.abs(again[f] - got[f]).max() <= 2.4e-7
        assert pts[f].min() <= got[f].min() and got[f].max() <= pts[f].max()
    assert (again['tile'] == got['tile']).all() and (again['r'] == got['r']).all()
    # what the widened bar uses at BASELINE configs[1] (VERDICT round 2, weak point 1): on record for DESIGN section 4
    used = {"%d points, cellsize %g" % k: v for k, v in TOLERANCE_USED.items() if k[0] == len(pts)}
    octree = TOLERANCE_USED[(len(pts), 0.01)]
    assert octree["hip_vs_oracle_max_pop_le_300"] <= XYZ_TOL and octree["hip_vs_f64_mean_max"] <= 3.0e-7
    import json, os
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(outdir):
        with open(os.path.join(outdir, "tolerance_used.json"), "w") as f:
            json.dump(used, f, indent=1)


def test_full_size_downsample_is_order_independent(gpu, full_cloud):
    """The plain grid emits voxels in index order, so any input order must give the same cloud;
    the sums are integers, so the equality is exact (the reference's fp32 sums are not)."""
    pts, cs = full_cloud
    rng = np.random.default_rng(20260129)   # the supplemental permuted input
    a = gpu.cwipc_downsample(make_cloud(gpu, pts, cs), -0.01).get_numpy_array()
    b = gpu.cwipc_downsample(make_cloud(gpu, pts[rng.permutation(len(pts))], cs), -0.01).get_numpy_array()
    assert same(a, b)


def test_full_size_tilefilter_join(gpu, oracle, full_cloud):
    pts, cs = full_cloud
    pc = make_cloud(gpu, pts, cs, 7)
    t1, t2 = gpu.cwipc_tilefilter(pc, 1), gpu.cwipc_tilefilter(pc, 2)
    assert t1.count() == 4999122 and t2.count() == 4999122
    assert same(t1.get_numpy_array(), oracle.tilefilter(pts, 1))
    j = gpu.cwipc_join(t1, t2).get_numpy_array()
    assert same(j, oracle.join(oracle.tilefilter(pts, 1), oracle.tilefilter(pts, 2)))


def test_full_size_remove_outliers(gpu, oracle, full_cloud):
    pts, cs = full_cloud
    got, exp = check_sor(gpu, oracle, pts, cs, 16, 1.0)
    assert 0 < len(got) < len(pts)


# ---------------------------------------------------------------------------
# threads: the reference calls filters from worker threads (net/source_synchronizer.py:17,184)
# ---------------------------------------------------------------------------
def test_filters_from_several_threads(gpu, oracle, synth):
    """Every thread has its own stream and workspace; a cloud produced on one thread (its last kernel may
    still be in flight when the call returns) is consumed on another."""
    import threading
    pts, cs = synth(300000)
    exp, _ = oracle.downsample(pts, cs, 0.01)
    exp_t1 = oracle.tilefilter(exp, 1)
    src = make_cloud(gpu, pts, cs, 9)
    produced, errors = [None] * 4, []

    def producer(i):
        try:
            for _ in range(5):
                produced[i] = gpu.cwipc_downsample(src, 0.01)
        except Exception as e:   # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=producer, args=(i,)) for i in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors
    results = []

    def consumer(i):
        try:
            results.append(gpu.cwipc_tilefilter(produced[i], 1).get_numpy_array())
        except Exception as e:   # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=consumer, args=((i + 1) % 4,)) for i in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors and len(results) == 4
    for r in results:
        assert len(r) == len(exp_t1)
        assert (r['tile'] == 1).all() and (r['r'] == exp_t1['r']).all()
        assert np.abs(r['x'] - exp_t1['x']).max() <= XYZ_TOL
    # the main thread reads what worker threads produced
    for pc in produced:
        got = pc.get_numpy_array()
        assert len(got) == len(exp) and (got['tile'] == exp['tile']).all()


def test_results_outlive_the_thread_that_made_them(gpu, oracle, synth):
    """Clouds made by threads that have ended (their streams retired, their `ready` events recycled) are
    consumed, freed and followed by many more calls on the main thread; none of those may fail with an
    error left over from an event of a stream that no longer exists."""
    import threading, gc
    pts, cs = synth(100000)
    exp, _ = oracle.downsample(pts, cs, 0.01)
    exp_t1 = oracle.tilefilter(exp, 1)
    src = make_cloud(gpu, pts, cs, 3)
    for generation in range(6):
        made, errors = [], []

        def worker():
            try:
                made.append(gpu.cwipc_tilefilter(gpu.cwipc_downsample(src, 0.01), 1))
            except Exception as e:   # pragma: no cover
                errors.append(e)

        threads = [threading.Thread(target=worker) for _ in range(3)]
        for t in threads: t.start()
        for t in threads: t.join()
        assert not errors
        for pc in made:
            assert gpu.cwipc_downsample(pc, 0.02).count() > 0    # device-side wait on the dead thread's event
            assert pc.count() == len(exp_t1) > 0
        del made
        gc.collect()
        for _ in range(10):                                      # recycled events, main thread's stream
            out = gpu.cwipc_remove_outliers(gpu.cwipc_downsample(src, 0.01), 8, 1.0, False)
            assert 0 < out.count() <= len(exp)


def test_input_freed_while_its_filter_result_is_pending(gpu, oracle, synth):
    """cwipc_tilefilter / cwipc_crop return with their last kernel still reading the input.  Freeing the input
    right away, and filling the memory it gives back with other clouds (here and on other threads), must
    not change the result."""
    import threading, gc
    pts, cs = synth(300000)
    exp = oracle.tilefilter(pts, 1)
    exp_crop = oracle.crop(pts, [-0.1, 0.1, 0.0, 1.0, -1, 1])
    other = pts.copy()
    other['x'] += 5.0
    other['tile'] = 1
    stop, errors = threading.Event(), []

    def churn():   # another thread grabbing and dirtying pool blocks of the same size class
        try:
            while not stop.is_set():
                gpu.cwipc_colormap(make_cloud(gpu, other, cs), 0xffffffff, 0x01020304).count()
        except Exception as e:   # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=churn)
    t.start()
    try:
        for i in range(40):
            pc = make_cloud(gpu, pts, cs)
            gpu.cwipc_hip_upload(pc, drop_host_copy=True)
            out = gpu.cwipc_tilefilter(pc, 1) if i % 2 == 0 else gpu.cwipc_crop(pc, [-0.1, 0.1, 0.0, 1.0, -1, 1])
            pc.free()
            del pc
            filler = make_cloud(gpu, other, cs)   # same size class: gets the block the input gave back
            gpu.cwipc_hip_upload(filler)
            assert same(out.get_numpy_array(), exp if i % 2 == 0 else exp_crop)
    finally:
        stop.set()
        t.join()
    assert not errors


def test_without_polling(gpu, oracle, synth, monkeypatch):
    """CWIPC_POLL_US=0: the calls that normally poll pinned memory for a kernel's report (downsample, tilefilter,
    crop) fall back to the stream wait at once; same results."""
    monkeypatch.setenv("CWIPC_POLL_US", "0")
    pts, cs = synth(300000)
    pc = make_cloud(gpu, pts, cs)
    exp, _ = oracle.downsample(pts, cs, 0.01)
    for _ in range(3):
        got = gpu.cwipc_downsample(pc, 0.01).get_numpy_array()
        assert len(got) == len(exp) and (got['tile'] == exp['tile']).all() and (got['r'] == exp['r']).all()
        assert np.abs(got['x'] - exp['x']).max() <= XYZ_TOL
        assert same(gpu.cwipc_tilefilter(pc, 1).get_numpy_array(), oracle.tilefilter(pts, 1))
        assert same(gpu.cwipc_crop(pc, [-0.1, 0.1, 0.0, 1.0, -1, 1]).get_numpy_array(), oracle.crop(pts, [-0.1, 0.1, 0.0, 1.0, -1, 1]))
    expg, _ = oracle.downsample(pts, cs, -0.01)
    gotg = gpu.cwipc_downsample(pc, -0.01).get_numpy_array()
    assert len(gotg) == len(expg) and (gotg['tile'] == expg['tile']).all()


def test_synchronizer_with_the_device_join(gpu, oracle, synth):
    """The tile synchroniser (cwipc_util_amd/net) on real clouds: its one n-ary GPU join against the reference's
    left fold of pairwise joins (oracle/synchronizer.py), timestamps and cellsizes included."""
    from cwipc_util_amd.net.source_synchronizer import SyncCore, cwipc_source_synchronizer
    from oracle.synchronizer import ScriptedSource, run_reference_loop

    class HostCloud:   # the oracle's side: numpy records
        def __init__(self, pts, ts, cs): self.pts, self.ts, self.cs = pts, ts, cs
        def timestamp(self): return self.ts
        def cellsize(self): return self.cs
        def payload(self): return self.pts

    rng = np.random.default_rng(5)
    base, _ = synth(20000)
    scripts = []
    for t in range(4):
        ts, frames = 100 + t % 2, []
        for f in range(5):
            pts = base[rng.integers(0, len(base), int(rng.integers(1, 3000)))].copy()
            pts['tile'] = 1 << t
            frames.append((pts, ts, 0.001 * (1 + (t + f) % 3)))
            ts += int(rng.integers(1, 3))
        scripts.append(frames)
    exp, exp_stats = run_reference_loop([ScriptedSource([HostCloud(*fr) for fr in frames]) for frames in scripts], oracle.join, True)
    assert len(exp) >= 3

    def device_sources():
        return [ScriptedSource([make_cloud(gpu, p, cs, ts) for p, ts, cs in frames]) for frames in scripts]

    core = SyncCore(device_sources())
    got = []
    while not any(s.eof() for s in core.sources):
        r = core.poll()
        if r is not None:
            got.append(r)
    assert len(got) == len(exp)
    for g, (ts, cs, pts) in zip(got, exp):
        assert g.timestamp() == ts and g.cellsize() == np.float32(cs)
        assert same(g.get_numpy_array(), pts)
    assert core.missing_per_occurrence == exp_stats["missing"] and core.late_per_occurrence == exp_stats["late"]
    # the same through the thread and its queue
    sync = cwipc_source_synchronizer(None, device_sources())
    assert sync.start()
    for ts, cs, pts in exp:
        pc = sync.output_queue.get(timeout=60)
        assert pc is not None and pc.timestamp() == ts and same(pc.get_numpy_array(), pts)
    sync.stop()
    assert sync.eof()


def test_from_device_slots(gpu, oracle, synth):
    """The receive buffer of the multi-GPU join (one slot per rank: header rows, then that rank's records) becomes
    the fused cloud in one pass; same bytes as joining the per-rank clouds one after the other."""
    import torch
    pts, _ = synth(50000)
    parts = [pts[:0], pts[:7], pts[100:100 + 4096], pts[5000:5001], pts[10000:10000 + 12345]]
    rows, header = 12345 + 2 + 13, 2
    buf = np.zeros((len(parts), rows, 4), dtype=np.int32)
    buf[:, :, :] = -7                                   # rows outside [header, header + count) must not matter
    for s, part in enumerate(parts):
        buf[s, header:header + len(part)] = part.view(np.int32).reshape(-1, 4)
    dev = torch.from_numpy(buf).to("cuda")
    torch.cuda.synchronize()
    out = gpu.cwipc_hip_from_device_slots(dev.data_ptr(), rows, header, [len(p) for p in parts], 4242, 0.25)
    exp = parts[0]
    for part in parts[1:]:
        exp = oracle.join(exp, part)
    assert out.timestamp() == 4242 and out.cellsize() == 0.25
    assert same(out.get_numpy_array(), exp)
    # nothing at all
    assert gpu.cwipc_hip_from_device_slots(dev.data_ptr(), rows, header, [0, 0], 1, 1.0).count() == 0
    # a count that does not fit its slot is refused
    with pytest.raises(gpu.CwipcError):
        gpu.cwipc_hip_from_device_slots(dev.data_ptr(), rows, header, [rows], 1, 1.0)


def test_path_vectors(gpu):
    """The HIP path against the committed vectors (tests/golden/path_vectors.npz), without the oracle in the loop."""
    d = np.load(os.path.join(GOLDEN, "path_vectors.npz"))
    pts, cs = d["input"], float(d["cellsize"])
    pc = make_cloud(gpu, pts, cs, 99)
    for name, cell in (("down_p05", 0.05), ("down_p20", 0.2), ("down_m05", -0.05), ("down_m20", -0.2)):
        out = gpu.cwipc_downsample(pc, cell)
        got, exp = out.get_numpy_array(), d[name]
        assert len(got) == len(exp) and np.float32(out.cellsize()) == d[name + "_cellsize"]
        for f in ('r', 'g', 'b', 'tile'):
            assert (got[f] == exp[f]).all()
        for f in ('x', 'y', 'z'):
            assert np.abs(got[f] - exp[f]).max() <= XYZ_TOL
    md, _ = gpu.cwipc_hip_knn_mean_dist(pc, 8, 1.0)
    assert md.tobytes() == d["knn8"].tobytes()
    assert same(gpu.cwipc_remove_outliers(pc, 8, 1.0, False).get_numpy_array(), d["sor_k8_s1"])
    assert same(gpu.cwipc_remove_outliers(pc, 8, 1.0, True).get_numpy_array(), d["sor_k8_s1_pertile"])
    assert same(gpu.cwipc_tilefilter(pc, 1).get_numpy_array(), d["tilefilter_1"])
    assert same(gpu.cwipc_crop(pc, [-0.1, 0.2, 0.5, 1.5, -0.3, 0.05]).get_numpy_array(), d["crop"])
    assert same(gpu.cwipc_colormap(pc, 0x00ff00ff, 0x05000007).get_numpy_array(), d["colormap"])
    assert same(gpu.cwipc_tilemap(pc, bytes((i * 7 + 3) % 256 for i in range(256))).get_numpy_array(), d["tilemap"])
    assert same(gpu.cwipc_join(make_cloud(gpu, pts[:100], cs), make_cloud(gpu, pts[300:], cs)).get_numpy_array(), d["join"])


def test_hundred_million_points(gpu, oracle, synth):
    """1.6 GB of points: more workgroups than compute units in the voxel kernel, index arithmetic well beyond 2^24,
    a tilefilter result of 800 MB.  Against the oracle (which needs a few seconds here)."""
    pts, cs = synth(100_000_000)
    assert len(pts) > 99_000_000
    pc = make_cloud(gpu, pts, cs)
    gpu.cwipc_hip_upload(pc, drop_host_copy=True)
    for cell in (0.01, -0.01):
        got = gpu.cwipc_downsample(pc, cell).get_numpy_array()
        exp, _ = oracle.downsample(pts, cs, cell)
        assert len(got) == len(exp)
        for f in ('r', 'g', 'b', 'tile'):
            assert (got[f] == exp[f]).all()
        # ~2500 points per voxel (and whole rows at the apex): the oracle's fp32 running sums, PCL's AccumulatorXYZ,
        # reach thousands, where one ulp is 5e-4 -- the bar here is the checker's own rounding, not the HIP path's
        # (which sums integers and rounds once: test_downsample_means_are_correctly_rounded)
        for f in ('x', 'y', 'z'):
            assert np.abs(got[f].astype(np.float64) - exp[f]).max() <= 2e-3
    got = gpu.cwipc_tilefilter(pc, 1).get_numpy_array()
    exp = oracle.tilefilter(pts, 1)
    assert len(got) == len(exp) and got.tobytes() == exp.tobytes()


def test_threads_mixed_chains(gpu, oracle, synth):
    """Four threads run chains of filters on shared inputs for a while, dropping intermediate clouds at once (results
    carry `ready` events, inputs `reader` events, downsamples alternate between two streams): every result must
    be what the oracle says for that chain."""
    import threading
    pts, cs = synth(200000)
    shared = make_cloud(gpu, pts, cs)
    gpu.cwipc_hip_upload(shared)
    bbox = [-0.2, 0.2, 0.0, 1.5, -0.3, 0.3]
    d_oct, _ = oracle.downsample(pts, cs, 0.02)
    d_grid, _ = oracle.downsample(pts, cs, -0.02)
    exp = {
        "tile_then_down": oracle.downsample(oracle.tilefilter(pts, 1), cs, 0.02)[0],
        "down_then_crop": oracle.crop(d_oct, bbox),
        "grid_then_tile": oracle.tilefilter(d_grid, 2),
        "crop_join": oracle.join(oracle.crop(pts, bbox), oracle.tilefilter(pts, 2)),
        "colormap_down": oracle.downsample(oracle.colormap(pts, 0x00ff0000, 0x00110000), cs, -0.02)[0],
    }
    errors = []

    def close(got, want):
        if len(got) != len(want):
            return False
        if not ((got['tile'] == want['tile']).all() and (got['r'] == want['r']).all() and (got['g'] == want['g']).all() and (got['b'] == want['b']).all()):
            return False
        return all(np.abs(got[f].astype(np.float64) - want[f]).max() <= 5e-5 for f in ('x', 'y', 'z')) if len(got) else True

    def worker(seed):
        rng = np.random.default_rng(seed)
        try:
            for _ in range(25):
                which = list(exp)[int(rng.integers(0, len(exp)))]
                if which == "tile_then_down":
                    out = gpu.cwipc_downsample(gpu.cwipc_tilefilter(shared, 1), 0.02)
                elif which == "down_then_crop":
                    out = gpu.cwipc_crop(gpu.cwipc_downsample(shared, 0.02), bbox)
                elif which == "grid_then_tile":
                    out = gpu.cwipc_tilefilter(gpu.cwipc_downsample(shared, -0.02), 2)
                elif which == "crop_join":
                    out = gpu.cwipc_join(gpu.cwipc_crop(shared, bbox), gpu.cwipc_tilefilter(shared, 2))
                else:
                    out = gpu.cwipc_downsample(gpu.cwipc_colormap(shared, 0x00ff0000, 0x00110000), -0.02)
                if not close(out.get_numpy_array(), exp[which]):
                    errors.append((seed, which))
        except Exception as e:   # pragma: no cover
            errors.append((seed, repr(e)))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors[:5]


def test_join_pipeline_on_the_device_path(gpu, oracle, synth):
    """join_across_ranks and JoinPipeline over RCCL with a one-rank group (the real device path: pack kernel into the
    send slot, all_gather_into_tensor, slots -> cloud), frames of changing size including ones that outgrow the slots."""
    import socket
    import torch
    import torch.distributed as dist
    from cwipc_util_amd.multigpu import JoinPipeline, join_across_ranks
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    torch.cuda.set_device(0)
    try:
        import datetime
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0),
                                timeout=datetime.timedelta(seconds=120))
    except Exception as e:   # an environment without a usable RCCL bootstrap is not a failure of the library
        pytest.skip(f"no one-rank RCCL group here: {e!r}")
    try:
        pts, cs = synth(100000)
        frames = [pts[:3000], pts[:3500], pts[:0], pts[5000:5000 + 40000], pts[:100], pts[:90000], pts[200:700]]
        clouds = [make_cloud(gpu, f, cs, 1000 + i) for i, f in enumerate(frames)]
        for exchange in ("torch", "library", None):   # None: the default, the library's own exchange on an RCCL group
            for pc, f in zip(clouds, frames):
                out = join_across_ranks(pc, exchange=exchange)
                assert same(out.get_numpy_array(), f) and out.timestamp() == pc.timestamp() and out.cellsize() == pc.cellsize(), exchange
        pipe = JoinPipeline()
        got = [pipe.submit(pc) for pc in clouds] + [pipe.flush()]
        assert got[0] is None
        for out, f, pc in zip(got[1:], frames, clouds):
            assert same(out.get_numpy_array(), f) and out.timestamp() == pc.timestamp() and out.cellsize() == pc.cellsize()
        assert pipe.flush() is None
    finally:
        dist.destroy_process_group()


def test_library_exchange_on_one_rank(gpu, oracle, synth):
    """The exchange inside the library (cwipc_hip_comm_join: RCCL linked into libcwipc_util.so) on a one-rank communicator, no
    torch.distributed anywhere: the plain call (one rank: the result holds the input's planes) and the loopback flavour, where the
    record all-gather and the four planes really travel through ncclAllGather / ncclSend / ncclRecv (to this rank itself) and
    land at their displacement in a result of their own.  Frames of changing size, an empty cloud, no cloud, and results that
    outlive their input."""
    import gc
    comm = gpu.cwipc_hip_comm(gpu.cwipc_hip_comm_unique_id(), 0, 1)
    try:
        pts, cs = synth(100000)
        frames = [pts[:3000], pts[:3501], pts[:0], pts[5000:5000 + 40000], pts[:1], pts[:90000], pts[200:703]]
        for loopback in (False, True):
            for i, f in enumerate(frames):
                pc = make_cloud(gpu, f, cs, 1000 + i)
                gpu.cwipc_hip_upload(pc, drop_host_copy=True)
                out = comm.join(pc, loopback=loopback)
                pc.free(force=True)          # the sends may still be reading it: the library keeps the planes until they have
                del pc
                gc.collect()
                filler = make_cloud(gpu, pts[:len(f)][::-1].copy(), cs, 1)   # takes what the freed input gave back to the pool
                gpu.cwipc_hip_upload(filler)
                assert same(out.get_numpy_array(), f), (loopback, i)
                assert out.timestamp() == 1000 + i and out.cellsize() == np.float32(cs)
            none = comm.join(None, loopback=loopback)
            assert none.count() == 0 and none.timestamp() == 0 and none.cellsize() == 0.0
        # a result of a filter that is still running when the join is called (deferred downsample results settle inside)
        pc = make_cloud(gpu, pts, cs, 5)
        want = check_downsample(gpu, oracle, pts, cs, 0.02)[0]
        for _ in range(4):
            out = comm.join(gpu.cwipc_downsample(pc, 0.02), loopback=True)
            assert same(out.get_numpy_array(), want) and out.cellsize() == np.float32(0.02)
    finally:
        comm.free()
    with pytest.raises(gpu.CwipcError):
        comm.join(None)


def test_library_exchange_stream_of_submitted_frames(gpu, oracle, synth):
    """cwipc_hip_comm_submit: the fused cloud is handed out at once and the exchange happens on a thread of the communicator,
    frame after frame in the order of the calls.  Inputs are freed right after the call (or are results of filters that are
    still running), some frames have no tile, a plain join() in between takes its place in the queue; every result -- points,
    count, and the timestamp and cellsize that only the exchange knows -- must be what join() gives."""
    import gc
    pts, cs = synth(100000)
    comm = gpu.cwipc_hip_comm(gpu.cwipc_hip_comm_unique_id(), 0, 1)
    try:
        for loopback in (False, True):
            frames = [pts[:3000], pts[:3501], pts[:0], None, pts[5000:5000 + 40000], pts[:1], None, pts[:90000], pts[200:703]]
            handed = []
            for i, f in enumerate(frames):
                pc = None
                if f is not None:
                    pc = make_cloud(gpu, f, cs + i, 1000 + i)
                    gpu.cwipc_hip_upload(pc, drop_host_copy=(i % 2 == 0))
                handed.append(comm.submit(pc, loopback=loopback))
                if pc is not None:
                    pc.free(force=True)
                    del pc
                if i == 4:   # a waiting call between submitted ones: exchanged in its turn
                    mid = comm.join(make_cloud(gpu, pts[:777], 3.0, 42), loopback=loopback)
                    assert same(mid.get_numpy_array(), pts[:777]) and mid.timestamp() == 42 and mid.cellsize() == 3.0
            gc.collect()
            for i, (f, out) in enumerate(zip(frames, handed)):
                if f is None:
                    assert out.count() == 0 and out.timestamp() == 0 and out.cellsize() == 0.0, (loopback, i)
                else:
                    assert out.timestamp() == 1000 + i and out.cellsize() == np.float32(cs + i), (loopback, i)   # (before the points are asked for)
                    assert same(out.get_numpy_array(), f), (loopback, i)
        # results of filters that are still running when they are submitted; the fused cloud goes straight into another filter
        pc = make_cloud(gpu, pts, cs, 5)
        want = gpu.cwipc_downsample(pc, 0.02).get_numpy_array()
        outs = [comm.submit(gpu.cwipc_downsample(pc, 0.02), loopback=True) for _ in range(6)]
        for out in outs:
            assert out.cellsize() == np.float32(0.02) and same(out.get_numpy_array(), want)
        again = gpu.cwipc_tilefilter(comm.submit(gpu.cwipc_downsample(pc, 0.02)), 1)
        assert same(again.get_numpy_array(), want[want['tile'] == 1])
    finally:
        comm.free()


def test_library_exchange_refuses_what_cannot_work(gpu, monkeypatch):
    """Between processes RCCL needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver: without it the communicator is refused at
    creation (before any rank can be left waiting inside a collective); bad ranks and ids are refused too."""
    uid = gpu.cwipc_hip_comm_unique_id()
    assert len(uid) == 128
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    with pytest.raises(gpu.CwipcError, match="HSA_ENABLE_IPC_MODE_LEGACY"):
        gpu.cwipc_hip_comm(uid, 0, 2)
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    with pytest.raises(gpu.CwipcError, match="HSA_ENABLE_IPC_MODE_LEGACY"):
        gpu.cwipc_hip_comm(uid, 1, 2)
    with pytest.raises(gpu.CwipcError):
        gpu.cwipc_hip_comm(uid, 3, 2)
    with pytest.raises(ValueError):
        gpu.cwipc_hip_comm(uid[:64], 0, 1)


def test_results_that_share_planes_outlive_their_input(gpu, oracle, synth):
    """Results of colormap / tilemap / transform / tilefilter(0) / a join with empty partners hold planes of their input
    instead of copies (clouds are immutable); freeing the input must leave them whole, and the input whole when they go."""
    import gc
    pts, cs = synth(50000)
    m = np.eye(4); m[0, 3] = 0.25

    def fresh():
        pc = make_cloud(gpu, pts, cs, 7)
        gpu.cwipc_hip_upload(pc, drop_host_copy=True)
        return pc

    makers = {
        "colormap": (lambda pc: gpu.cwipc_colormap(pc, 0x0000ff00, 0x00003300), oracle.colormap(pts, 0x0000ff00, 0x00003300)),
        "tilemap": (lambda pc: gpu.cwipc_tilemap(pc, {1: 4, 2: 8}), oracle.tilemap(pts, bytes([0, 4, 8] + [0] * 253))),
        "transform": (lambda pc: gpu.cwipc_transform(pc, m), oracle.transform(pts, m)),
        "tilefilter0": (lambda pc: gpu.cwipc_tilefilter(pc, 0), pts),
        "join_with_empty": (lambda pc: gpu.cwipc_join(gpu.cwipc_from_points([], 3), pc), pts),
    }
    for name, (make, exp) in makers.items():
        pc = fresh()
        out = make(pc)
        pc.free(force=True)
        del pc
        gc.collect()
        filler = fresh()                      # takes whatever the freed input gave back to the pool
        assert same(out.get_numpy_array(), exp), name
        # the other way round: the result goes, the input stays
        pc = fresh()
        out = make(pc)
        out.free(force=True)
        del out
        gc.collect()
        filler2 = make_cloud(gpu, exp, cs)
        gpu.cwipc_hip_upload(filler2)
        assert same(pc.get_numpy_array(), pts), name
        del filler, filler2


def test_no_leaks_over_many_calls(gpu, synth):
    """Device pool, pinned pool and object counters stay put over a few hundred filter calls."""
    import gc
    pts, cs = synth(100000)
    dll = gpu.cwipc_util_dll_load()

    def frame():
        pc = make_cloud(gpu, pts, cs)
        a = gpu.cwipc_downsample(pc, 0.01)
        b = gpu.cwipc_remove_outliers(a, 16, 1.0, False)
        c = gpu.cwipc_tilefilter(b, 1)
        d = gpu.cwipc_join(c, gpu.cwipc_downsample(pc, -0.02))
        return d.get_numpy_array().shape[0]

    for _ in range(20):
        frame()
    gc.collect()
    dll.cwipc_hip_synchronize()
    bytes0 = gpu.cwipc_hip_pool_bytes() if hasattr(gpu, 'cwipc_hip_pool_bytes') else dll.cwipc_hip_pool_bytes()
    dangling0 = gpu.cwipc_dangling_allocations(False)
    counts = {frame() for _ in range(200)}
    gc.collect()
    dll.cwipc_hip_synchronize()
    bytes1 = gpu.cwipc_hip_pool_bytes() if hasattr(gpu, 'cwipc_hip_pool_bytes') else dll.cwipc_hip_pool_bytes()
    assert len(counts) == 1                       # same input, same result, every time
    assert bytes1 == bytes0, (bytes0, bytes1)     # the pool recycles; nothing new was needed
    assert gpu.cwipc_dangling_allocations(False) == dangling0
