"""BASELINE configs 3 (perTile), 4 and 5 at their stated sizes, through the C-ABI, against the CPU oracle.

Config 4: 8 tiles x synthetic(2 000 000, angle = i pi/4), camera mask 1 << i, rotated i x 45 degrees about Y
(reference python/cwipc/scripts/cwipc_create_analysis_test.py:70-79); per tile tilefilter(1 << i) -> downsample(0.01)
(python/cwipc/registration/util.py:170-182); fused by the n-ary join and by the multi-GPU exchange (one-rank RCCL group:
the real device path, everything but the wire).
Config 5: per frame 8 x synthetic(300 000) tiles, colorize(0.8, "camera") -> downsample(0.01) -> remove_outliers(16, 1.0)
per tile, 8-way join; checked stage by stage (every stage's input is the HIP output of the stage before, handed to the oracle).
"""
import math

import numpy as np
import pytest

from conftest import make_cloud

pytestmark = pytest.mark.gpu

XYZ_TOL = 1e-5
NTILES = 8


def same(a, b):
    return len(a) == len(b) and a.tobytes() == b.tobytes()


def oracle_tile(oracle, npoints, tile):
    """The oracle's restatement of capture.capture_tile()."""
    from cwipc_util_amd.capture import rotation_about_y
    angle = tile * 2.0 * math.pi / NTILES
    pts, cs = oracle.synthetic(npoints, angle)
    pts = oracle.tilemap(pts, bytes([1 << tile]) * 256)
    if tile:
        pts = oracle.transform(pts, rotation_about_y(angle))
    return pts, cs


def close_clouds(got, exp, tol=XYZ_TOL):
    assert len(got) == len(exp), (len(got), len(exp))
    for f in ('x', 'y', 'z'):
        err = np.abs(got[f].astype(np.float64) - exp[f].astype(np.float64))
        assert err.max(initial=0.0) <= tol, (f, float(err.max()))
    for f in ('r', 'g', 'b', 'tile'):
        assert (got[f] == exp[f]).all(), f


@pytest.fixture(scope="module")
def one_rank_rccl(gpu):
    """A one-rank RCCL process group on device 0 (None where RCCL cannot bootstrap)."""
    import datetime
    import socket
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        yield None
        return
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0),
                                timeout=datetime.timedelta(seconds=120))
    except Exception:
        yield None
        return
    yield dist
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def config4(gpu, oracle):
    """The eight 2 M-point tiles on the device, next to the oracle's copies."""
    from cwipc_util_amd.capture import capture_tile
    tiles, expect = [], []
    for t in range(NTILES):
        pc = capture_tile(2_000_000, t, NTILES, timestamp=5000 + t)
        pts, cs = oracle_tile(oracle, 2_000_000, t)
        assert pc.count() == 1999396 == len(pts)
        assert pc.cellsize() == pytest.approx(cs, rel=0, abs=0)
        tiles.append(pc)
        expect.append((pts, cs))
    return tiles, expect


def test_config4_tiles_are_the_recipe(gpu, oracle, config4):
    """The product's capture (synthetic source -> tilemap -> transform, all on the GPU) equals the recipe bit for bit."""
    tiles, expect = config4
    for t, (pc, (pts, cs)) in enumerate(zip(tiles, expect)):
        assert same(pc.get_numpy_array(), pts), t
        assert (pts['tile'] == 1 << t).all()


def test_config4_per_tile_chain_and_join(gpu, oracle, config4, one_rank_rccl):
    from cwipc_util_amd.capture import per_tile_chain
    tiles, expect = config4
    outs, exps = [], []
    for t, (pc, (pts, cs)) in enumerate(zip(tiles, expect)):
        out = per_tile_chain(pc, t, 0.01)
        kept = oracle.tilefilter(pts, 1 << t)
        assert len(kept) == len(pts)                     # a camera's tile holds that camera's points only
        e, ecs = oracle.downsample(kept, cs, 0.01)
        got = out.get_numpy_array()
        close_clouds(got, e)
        assert out.timestamp() == 5000 + t and out.cellsize() == pytest.approx(ecs, rel=0, abs=0)
        assert 30000 < len(got) < 45000                  # SURVEY: about 38.5 k voxels per 2 M-point tile
        outs.append(out)
        exps.append(e)
    # n-ary join on one GPU: rank order = tile order = the reference's fold order; ts = min, cellsize = min
    fused = gpu.cwipc_join_multi(outs)
    fold = exps[0]
    for e in exps[1:]:
        fold = oracle.join(fold, e)
    got = fused.get_numpy_array()
    close_clouds(got, fold)
    assert same(got, np.concatenate([o.get_numpy_array() for o in outs]))   # the join itself is bit-exact
    assert fused.timestamp() == 5000
    assert fused.cellsize() == min(o.cellsize() for o in outs)
    # the multi-GPU exchange (one rank holds all eight tiles: tile t -> rank t mod 1), plain and pipelined
    if one_rank_rccl is None:
        pytest.skip("no one-rank RCCL group here: the device exchange was not exercised")
    import torch
    from cwipc_util_amd.multigpu import JoinPipeline, join_across_ranks, tiles_of_rank
    assert tiles_of_rank(NTILES, 0, 1) == list(range(NTILES))
    local = gpu.cwipc_join_multi(outs)
    across = join_across_ranks(local)
    assert same(across.get_numpy_array(), got) and across.timestamp() == 5000 and across.cellsize() == fused.cellsize()
    pipe = JoinPipeline()
    assert pipe.submit(local) is None
    assert same(pipe.flush().get_numpy_array(), got)
    # under a stream of the caller's: the pack and unpack kernels must follow torch's CURRENT stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        again = join_across_ranks(local)
        side.synchronize()
    assert same(again.get_numpy_array(), got)


def test_config5_frame_stage_by_stage(gpu, oracle):
    """One frame of config 5, every stage against the oracle on the SAME input (the HIP output of the stage before)."""
    from cwipc_util_amd.capture import capture_tile
    from cwipc_util_amd.filters import factory
    from cwipc_util_amd.filters.colorize import ColorizeFilter
    lut, valid = ColorizeFilter(0.8, "camera").colorMap.tables()
    colorize, voxelize, outliers = factory('colorize(0.8, "camera")'), factory('voxelize(0.01)'), factory('remove_outliers(16, 1.0, False)')
    outs = []
    for t in range(NTILES):
        pc = capture_tile(300_000, t, NTILES, timestamp=900 + t)
        pts, cs = oracle_tile(oracle, 300_000, t)
        assert pc.count() == 299209 and same(pc.get_numpy_array(), pts)
        # colorize: bit-exact
        c = colorize.filter(pc)
        c_np = c.get_numpy_array()
        assert same(c_np, oracle.colorize(pts, 0.8, lut, valid)), t
        # downsample of the HIP colorize output
        d = voxelize.filter(c)
        d_np = d.get_numpy_array()
        e, ecs = oracle.downsample(c_np, cs, 0.01)
        close_clouds(d_np, e)
        assert d.cellsize() == pytest.approx(ecs, rel=0, abs=0)
        # outlier removal of the HIP downsample output: d_i bit for bit, mask exact up to threshold ties
        exp, d_exp, thr_exp = oracle.remove_outliers(d_np, 16, 1.0, False, want_stats=True)
        d_got, thr_got = gpu.cwipc_hip_knn_mean_dist(d, 16, 1.0)
        assert (d_got == d_exp).all(), t
        assert thr_got == pytest.approx(thr_exp, rel=1e-12)
        o = outliers.filter(d)
        o_np = o.get_numpy_array()
        assert same(o_np, d_np[~(d_got.astype(np.float64) > thr_got)])
        if not same(o_np, exp):
            band = np.abs(d_exp.astype(np.float64) - thr_exp) <= 1e-6 * abs(thr_exp)
            assert abs(len(o_np) - len(exp)) <= band.sum()
        assert 0 < len(o_np) < len(d_np)
        assert gpu.util.cwipc_util_dll_load().cwipc_hip_is_device_resident(o.as_cwipc_p()) == 1
        outs.append(o)
    fused = gpu.cwipc_join_multi(outs)
    assert same(fused.get_numpy_array(), np.concatenate([o.get_numpy_array() for o in outs]))
    assert fused.timestamp() == 900 and fused.cellsize() == min(o.cellsize() for o in outs)


def test_config5_stream_through_the_server(gpu, oracle):
    """The stream form of config 5: one TileSource per camera (each with the per-tile chain) -> synchroniser (one GPU join per
    frame) -> SourceServer -> sink, as the reference's SourceServer.run() drives it (scripts/_scriptsupport.py:346-390)."""
    from cwipc_util_amd.capture import TileSource, capture_tile
    from cwipc_util_amd.filters import factory
    from cwipc_util_amd.net.source_synchronizer import cwipc_source_synchronizer
    from cwipc_util_amd.scripts._scriptsupport import CountingSink, SourceServer, server_args
    nframes = 12
    tiles = [capture_tile(300_000, t, NTILES) for t in range(NTILES)]
    chain = lambda: [factory('colorize(0.8, "camera")'), factory('voxelize(0.01)'), factory('remove_outliers(16, 1.0, False)')]
    # what one frame must be: the per-tile chain by hand, joined
    flt = chain()
    one = []
    for pc in tiles:
        cur = pc
        for f in flt:
            cur = f.filter(cur)
        one.append(cur)
    want = gpu.cwipc_join_multi(one).get_numpy_array()
    # (a few frames more than the server takes: the synchroniser, like the reference's, reports end of file as soon as a
    # source has run dry, whatever is still queued)
    sources = [TileSource(pc, nframes + 8, filters=chain(), first_timestamp=1000, timestamp_step=33, threaded=(i % 2 == 1)) for i, pc in enumerate(tiles)]
    sync = cwipc_source_synchronizer(None, sources)
    sink = CountingSink()
    frames = []
    feed = sink.feed
    sink.feed = lambda pc: (frames.append((pc.timestamp(), pc.count(), pc.get_numpy_array() if len(frames) in (0, nframes - 1) else None)), feed(pc))
    server = SourceServer(sync, sink, server_args(count=nframes))
    server.run()
    server.stop()
    for src in sources:
        src.free()
    assert sink.frames == nframes
    assert [ts for ts, _, _ in frames] == [1000 + 33 * i for i in range(nframes)]
    assert all(n == len(want) for _, n, _ in frames)
    assert same(frames[0][2], want) and same(frames[-1][2], want)
    assert len(server.times_grab) == nframes and sync.core.missing_per_occurrence == []


def test_config5_eight_threads_workspace_footprint(gpu):
    """Eight decoder-like threads, each running the per-tile chain of config 5 on its own 300 k-point tile frame after frame
    (reference net/source_synchronizer.py:128-149: one thread per tile).  A thread whose downsample calls are separated by
    other filters holds ONE voxel workspace (the second exists only for back-to-back calls); its leaf grids (20 MB each) are
    sized from what the thread's clouds need and shrink again after eight roomy passes.  A tile of this capture is the whole
    synthetic figure (12 to 16 octree leaves of 0.64 m at 1 cm), so a workspace is 16 grids = 0.32 GB: eight threads and this
    one hold ~3.2 GB.  Round 1 held 21 GB here (64 grids x 2 workspaces per thread)."""
    import gc
    import threading
    from cwipc_util_amd.capture import capture_tile
    from cwipc_util_amd.filters import factory
    dll = gpu.util.cwipc_util_dll_load()
    # what threads of earlier tests left in the pool of workspaces goes back to the device first (cwipc_hip_workspace_trim, r4), so that the
    # bound below is about THIS test's eight threads; what live threads of earlier tests hold (an executor's workers) is measured and taken off
    gc.collect()
    dll.cwipc_hip_synchronize()
    dll.cwipc_hip_workspace_trim()
    held_before = dll.cwipc_hip_workspace_bytes()
    tiles = [capture_tile(300_000, t, NTILES) for t in range(NTILES)]
    counts, errors = {}, []

    def work(t):
        try:
            flt = [factory('colorize(0.8, "camera")'), factory('voxelize(0.01)'), factory('remove_outliers(16, 1.0, False)')]
            seen = set()
            for _ in range(24):
                cur = tiles[t]
                for f in flt:
                    cur = f.filter(cur)
                seen.add(cur.count())
            counts[t] = seen
        except Exception as e:   # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(NTILES)]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors
    assert all(len(counts[t]) == 1 for t in range(NTILES)), counts      # every frame of a tile gives the same cloud
    # this thread's own workspaces may be sized for the 10 M-point clouds of earlier tests: the same rule shrinks them
    for _ in range(20):
        gpu.cwipc_downsample(tiles[0], 0.01).count()
    dll.cwipc_hip_synchronize()
    held = dll.cwipc_hip_workspace_bytes()
    # eight threads' workspaces of 16 grids (0.32 GB each) on top of what was there, or ten in all when nothing was
    assert held - held_before <= 2.7 * 10**9, (held, held_before)   # (round 3: max(3.5e9, held_before + 2.7e9), which forgave whatever was there before)


@pytest.fixture(scope="module")
def full_cloud(oracle):
    return oracle.synthetic(10_000_000, 0.0)


def test_config3_per_tile_at_full_size(gpu, oracle, full_cloud):
    """BASELINE config 3 with perTile = true at 10 M points (reference src/cwipc_filters.cpp:238-261: tiles in
    first-appearance order, each filtered on its own, results appended)."""
    pts, cs = full_cloud
    out = gpu.cwipc_remove_outliers(make_cloud(gpu, pts, cs, 77), 16, 1.0, True)
    got = out.get_numpy_array()
    assert out.timestamp() == 77 and out.cellsize() == pytest.approx(cs, rel=0, abs=0)
    # per tile: the statistic is bit-identical, so the mask is fixed by the HIP path's own threshold;
    # tile 2 appears first in the synthetic cloud (its first point has y >= 0)
    order = []
    for t in pts['tile']:
        if t not in order:
            order.append(int(t))
        if len(order) == 2:
            break
    parts = []
    for t in order:
        sub = pts[pts['tile'] == t]
        d_got, thr_got = gpu.cwipc_hip_knn_mean_dist(make_cloud(gpu, sub, cs), 16, 1.0)
        d_exp = oracle.knn_mean_dist(sub, 16)
        assert (d_got == d_exp).all(), t
        parts.append(sub[~(d_got.astype(np.float64) > thr_got)])
    assert same(got, np.concatenate(parts))
    assert 0 < len(got) < len(pts)
