"""tests/golden/helper_vectors.npz -- outputs of the REFERENCE's own Python (made by tests/golden/make_helper_vectors.py in the
build container: `_Synchronizer.run`, `cwipc_tilefilter_masked`, `get_tiles_used`, `cwipc_transform`, `cwipc_downsample_pertile`,
`TransformFilter.filter`, bodies unmodified) -- against the checkers that restate them (oracle/synchronizer.py, oracle/oracle.py)
and against the product's host logic (SyncCore).  CPU only; the GPU paths meet the same fixture in tests/test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

from cwipc_util_amd.net.source_synchronizer import SyncCore
from oracle import oracle
from oracle.synchronizer import ScriptedSource, run_reference_loop

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "helper_vectors.npz")


@pytest.fixture(scope="module")
def vectors():
    d = np.load(FIXTURE)
    return d, json.loads(bytes(d["meta_json"]).decode())


class FakeCloud:
    def __init__(self, pts, ts, cellsize):
        self.pts, self.ts, self.cs = pts, ts, cellsize

    def timestamp(self): return self.ts
    def cellsize(self): return self.cs
    def count(self): return len(self.pts)
    def payload(self): return self.pts
    def _set_timestamp(self, ts): self.ts = ts
    def _set_cellsize(self, cs): self.cs = cs
    def free(self): pass


def scripted_sources(script):
    """The generator's clouds: n points, x = 0 .. n-1, y = tile, z = index of the cloud in its tile's script, tile = 1 << tile."""
    sources = []
    for t, rows in enumerate(script):
        clouds = []
        for k, (ts, cs, n, _gate) in enumerate(rows):
            p = oracle.empty(n)
            p['x'] = np.arange(n)
            p['y'], p['z'], p['tile'] = t, k, 1 << t
            clouds.append(FakeCloud(p, ts, float(np.float32(cs))))   # (a cloud's cellsize is a C float)
        sources.append(ScriptedSource(clouds, [r[3] for r in rows]))
    return sources


def parts_of(arr):
    parts = []
    for i in range(len(arr)):
        key = [int(arr['y'][i]), int(arr['z'][i])]
        if not parts or parts[-1] != key:
            parts.append(key)
    return parts


def nary_join(clouds):
    """What cwipc_join_multi does: one cloud -> that cloud; else one concatenation in the given order."""
    if len(clouds) == 1:
        return clouds[0]
    return FakeCloud(np.concatenate([c.pts for c in clouds]), min(c.ts for c in clouds), min(c.cs for c in clouds))


def test_fixture_is_there(vectors):
    d, meta = vectors
    assert len(meta["synchronizer"]) == 48 and len(meta["pertile"]) == 5
    assert sum(len(c["stats"]["late"]) for c in meta["synchronizer"]) > 0      # every branch of the policy is in the fixture
    assert sum(len(c["stats"]["missing"]) for c in meta["synchronizer"]) > 0
    assert sum(len(c["stats"]["desync"]) for c in meta["synchronizer"]) > 0


def test_oracle_synchronizer_reproduces_the_reference(vectors):
    """oracle/synchronizer.py (the restatement the product is checked against step by step) against the reference's own loop."""
    _, meta = vectors
    for case in meta["synchronizer"]:
        produced, stats = run_reference_loop(scripted_sources(case["script"]), oracle.join, case["prefer_partial_over_unsynced"])
        assert stats == case["stats"]
        assert len(produced) == len(case["produced"])
        for (ts, cs, payload), exp in zip(produced, case["produced"]):
            assert ts == exp["timestamp"] and cs == exp["cellsize"] and len(payload) == exp["count"]
            assert parts_of(payload) == exp["parts"]


def test_product_synchronizer_reproduces_the_reference(vectors):
    """cwipc_util_amd.net.source_synchronizer.SyncCore (the product's policy object; the join injected, CPU) against the same."""
    _, meta = vectors
    for case in meta["synchronizer"]:
        core = SyncCore(scripted_sources(case["script"]), join=nary_join, prefer_partial_over_unsynced=case["prefer_partial_over_unsynced"])
        produced = []
        for _ in range(100000):
            if any(s.eof() for s in core.sources):
                break
            r = core.poll()
            if r is not None:
                produced.append((r.timestamp(), r.cellsize(), r.payload().copy()))
        assert {"late": core.late_per_occurrence, "desync": core.desync_per_occurrence, "missing": core.missing_per_occurrence} == case["stats"]
        assert len(produced) == len(case["produced"])
        for (ts, cs, payload), exp in zip(produced, case["produced"]):
            assert ts == exp["timestamp"] and cs == exp["cellsize"] and len(payload) == exp["count"]
            assert parts_of(payload) == exp["parts"]


def test_oracle_masked_tilefilter_and_tiles_used(vectors):
    d, _ = vectors
    for i in range(6):
        pts = d["masked%d_in" % i]
        assert oracle.tiles_used(pts) == d["masked%d_tiles_used" % i].tolist()
        for m in (0, 1, 2, 3, 4, 8, 15, 128, 255):
            exp = d["masked%d_mask%d_out" % (i, m)]
            got = oracle.tilefilter_masked(pts, m)
            assert got.tobytes() == exp.tobytes(), (i, m)
            ts, cs = d["masked%d_mask%d_meta" % (i, m)]
            assert ts == 100 + i
            # an empty result comes from cwipc_from_points([], ts): cellsize 0 (reference registration/util.py:108-109)
            assert cs == (0.0 if len(exp) == 0 else np.float32(0.003 + 0.001 * i))


def test_oracle_transform(vectors):
    d, _ = vectors
    pts = d["transform_in"]
    for name in ("identity", "rot_y_45", "translate", "general", "scale_shear"):
        got = oracle.transform(pts, d["transform_%s_matrix" % name])
        assert got.tobytes() == d["transform_%s_out" % name].tobytes(), name
        assert d["transform_%s_meta" % name].tolist() == [7.0, float(np.float32(0.004))]


def test_oracle_offset_scale(vectors):
    d, _ = vectors
    pts = d["offsetscale_in"]
    for i in range(4):
        x, y, z, scale = d["offsetscale%d_params" % i]
        got = oracle.offset_scale(pts, x, y, z, scale)
        assert got.tobytes() == d["offsetscale%d_out" % i].tobytes(), i
        ts, cs = d["offsetscale%d_meta" % i]
        assert ts == 9 and cs == np.float32(np.float32(0.005) * scale)      # (reference transform.py:46: cellsize * scale, stored as a C float)


def test_oracle_downsample_pertile_order(vectors):
    """The calls `cwipc_downsample_pertile` makes (reference registration/util.py:170-182): tiles in ascending order of
    get_tiles_used, tilefilter -> downsample per tile, a left fold of joins."""
    _, meta = vectors
    for case in meta["pertile"]:
        calls = []
        oracle.downsample_pertile_plan(case["tiles_in_cloud"], 0.0125, calls)
        assert calls == case["calls"]
