"""The tile synchroniser (cwipc_util_amd/net/source_synchronizer.py) against the restatement of the reference's
loop (oracle/synchronizer.py): which clouds are combined, in which order, with which timestamp and cellsize, and
the late / desync / missing statistics -- on scripted sources, without a clock.  CPU only: the join is injected
(an n-ary concatenation, as cwipc_join_multi does it on the GPU); tests/test_gpu_parity.py runs the real one."""
import numpy as np
import pytest

from cwipc_util_amd.net.source_synchronizer import SyncCore, cwipc_source_synchronizer
from oracle import oracle
from oracle.synchronizer import ScriptedSource, run_reference_loop


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


def nary_join(clouds):
    """What cwipc_join_multi does: one cloud -> that cloud; else one concatenation in the given order."""
    if len(clouds) == 1:
        return clouds[0]
    return FakeCloud(np.concatenate([c.pts for c in clouds]), min(c.ts for c in clouds), min(c.cs for c in clouds))


def make_scripts(rng, n_tile, n_frames):
    scripts = []
    for t in range(n_tile):
        ts, clouds, gates = int(rng.integers(0, 3)), [], []
        for _ in range(n_frames):
            ts += int(rng.integers(0, 3))            # equal timestamps, gaps, tiles that skip frames
            n = int(rng.integers(0, 6))
            pts = oracle.empty(n)
            pts['x'] = rng.random(n)
            pts['tile'] = 1 << t
            clouds.append((pts, ts, float(rng.choice([0.001, 0.002, 0.004]))))
            gates.append(int(rng.integers(0, 4)))
            ts += 1
        scripts.append((clouds, gates))
    return scripts


def sources_of(scripts):
    return [ScriptedSource([FakeCloud(p.copy(), ts, cs) for p, ts, cs in clouds], gates) for clouds, gates in scripts]


def drive_core(core, max_iterations=100000):
    produced = []
    for _ in range(max_iterations):
        if any(s.eof() for s in core.sources):
            break
        r = core.poll()
        if r is not None:
            produced.append((r.timestamp(), r.cellsize(), r.payload().copy()))
    return produced


@pytest.mark.parametrize("prefer_partial", [True, False])
@pytest.mark.parametrize("seed", range(25))
def test_policy_matches_reference_loop(seed, prefer_partial):
    rng = np.random.default_rng(seed)
    n_tile, n_frames = int(rng.integers(1, 6)), int(rng.integers(1, 12))
    scripts = make_scripts(rng, n_tile, n_frames)
    exp, exp_stats = run_reference_loop(sources_of(scripts), oracle.join, prefer_partial)
    core = SyncCore(sources_of(scripts), join=nary_join, prefer_partial_over_unsynced=prefer_partial)
    got = drive_core(core)
    assert len(got) == len(exp)
    for (gts, gcs, gp), (ets, ecs, ep) in zip(got, exp):
        assert gts == ets and gcs == ecs
        assert len(gp) == len(ep) and gp.tobytes() == ep.tobytes()
    assert core.late_per_occurrence == exp_stats["late"]
    assert core.desync_per_occurrence == exp_stats["desync"]
    assert core.missing_per_occurrence == exp_stats["missing"]


def test_single_cloud_is_passed_on_as_it_is():
    """One tile: the reference's fold returns the input object itself (and re-stamps it)."""
    pts = oracle.empty(3)
    src = ScriptedSource([FakeCloud(pts, 7, 0.5), FakeCloud(pts, 9, 0.5)])
    core = SyncCore([src], join=nary_join)
    first = src.clouds[0]
    assert core.poll() is first and first.timestamp() == 7
    assert core.earliest_timestamp == 8


def test_late_tile_is_dropped_and_counted():
    a = [FakeCloud(oracle.empty(1), 10, 1.0), FakeCloud(oracle.empty(1), 20, 1.0), FakeCloud(oracle.empty(1), 30, 1.0)]
    b = [FakeCloud(oracle.empty(2), 20, 1.0), FakeCloud(oracle.empty(2), 5, 1.0), FakeCloud(oracle.empty(2), 30, 1.0), FakeCloud(oracle.empty(2), 40, 1.0)]
    core = SyncCore([ScriptedSource(a), ScriptedSource(b)], join=nary_join)
    got = drive_core(core)
    # ts 10 (tile a only), ts 20 (both), then b's stale ts-5 cloud is too late by 21 - 5; tile a has handed out
    # its last cloud by then, and the reference's loop stops as soon as a source reports end of file (:113-116)
    assert [(ts, len(p)) for ts, _, p in got] == [(10, 1), (20, 3)]
    assert core.late_per_occurrence == [16]
    assert core.missing_per_occurrence == [1]


def test_thread_produces_the_same_stream():
    rng = np.random.default_rng(99)
    scripts = make_scripts(rng, 3, 4)
    exp, _ = run_reference_loop(sources_of(scripts), oracle.join, True)
    assert 0 < len(exp) <= 5           # fits the output queue (6) together with the end marker
    sync = cwipc_source_synchronizer(None, sources_of(scripts))
    sync.core.join = nary_join
    assert sync.start()
    sync.join(timeout=30)              # the thread ends when a source reports end of file
    assert not sync.is_alive()
    got = []
    while True:
        pc = sync.output_queue.get_nowait()
        if pc is None:
            break
        got.append(pc)
    assert [(g.timestamp(), g.cellsize(), g.payload().tobytes()) for g in got] == [(ts, cs, p.tobytes()) for ts, cs, p in exp]
    assert sync.eof() and sync.get() is None and not sync.available(False)
    sync.statistics()
