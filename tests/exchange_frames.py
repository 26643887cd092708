"""The frames the multi-rank exchange tests run (tests/test_gpu_exchange_ranks.py and its child, tests/standin/ranks_child.py):
for every frame and rank (points, timestamp, cellsize, has_cloud).  Deterministic in (scenario, world, frame, rank): parent and
child build the same clouds.  The scenarios are those of tests/test_multigpu_gloo.py -- ragged, empty and missing tiles, regrowth --
plus the one round 2 deadlocked on (all points on one rank) and the faults of CWIPC_TEST_EXCHANGE_FAULTS."""
import numpy as np

POINT_DTYPE = np.dtype([('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('r', 'u1'), ('g', 'u1'), ('b', 'u1'), ('tile', 'u1')])


def cloud(n, frame, rank):
    rng = np.random.default_rng(1000003 * frame + 7919 * rank + n)
    p = np.zeros(n, dtype=POINT_DTYPE)
    p['x'], p['y'], p['z'] = rng.random(n), rng.random(n), rng.random(n)
    p['r'], p['g'], p['b'] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    p['tile'] = 1 << (rank % 8)
    return p


def frames_of(scenario, world):
    frames = []

    def frame(spec):   # spec[rank] = (n, has) ; timestamps and cell sizes differ per rank and frame so that the minima are told apart
        f = len(frames)
        frames.append([(cloud(n, f, r), 1000 + 10 * f - r, float(np.float32(0.001 * (1 + (r + f) % 3))), has) if has else (cloud(0, f, r), 0, 0.0, False)
                       for r, (n, has) in enumerate(spec)])

    if scenario == "stream":
        # a stream of frames of changing size: steady, growing by more than the 25 % of head room (a second gather round in the
        # middle of the stream), shrinking, ragged, an empty tile, a missing tile, everything on one rank, nothing anywhere
        frame([(3000 + 500 * r, True) for r in range(world)])
        frame([(3000 + 500 * r, True) for r in range(world)])
        frame([(3100 + 400 * r, True) for r in range(world)])
        frame([(9000 + 1000 * r, True) for r in range(world)])           # regrowth
        frame([(200 + 10 * r, True) for r in range(world)])              # much smaller: the room of the frame before serves
        frame([(0, True) if r == 1 else (1000 * r + 3, True) for r in range(world)])      # rank 1 holds an empty cloud
        frame([(0, False) if r == 0 else (700 + r, True) for r in range(world)])          # rank 0 has no tile this frame
        frame([(5000, True) if r == world - 1 else (0, False) for r in range(world)])     # all points on the last rank (round 2's deadlock)
        frame([(4000, True) if r == 0 else (0, True) for r in range(world)])              # all points on rank 0, the others hold empty clouds
        frame([(0, False) for r in range(world)])                                          # no tile anywhere
        frame([(0, True) for r in range(world)])                                           # empty tiles everywhere
        frame([(2500 + 300 * r, True) for r in range(world)])
    elif scenario == "faults":
        # what CWIPC_TEST_EXCHANGE_FAULTS (set by the test) strikes: see tests/test_gpu_exchange_ranks.py
        for f in range(8):
            frame([(2000 + 250 * r + 100 * f, True) for r in range(world)])
    else:
        raise ValueError(scenario)
    return frames


def expected(frames, world, absent=(), no_result=()):
    """The fold of cwipc_join over the tiles of every frame in rank order (reference src/cwipc_filters.cpp:388-418 folded by
    python/cwipc/net/source_synchronizer.py:175-188): points concatenated, timestamp and cellsize the minima over the tiles that
    took part.  absent: {(frame, rank)} ranks left out of a frame by a fault; returns per frame (points, ts, cellsize, any)."""
    out = []
    for f, per_rank in enumerate(frames):
        parts, ts, cs, any_cloud = [], None, None, False
        for r in range(world):
            pts, t, c, has = per_rank[r]
            if not has or (f, r) in absent:
                continue
            parts.append(pts)
            ts = t if ts is None else min(ts, t)
            cs = c if cs is None else min(cs, c)
            any_cloud = True
        fused = np.concatenate(parts) if parts else np.zeros(0, dtype=POINT_DTYPE)
        out.append((fused, ts if any_cloud else 0, cs if any_cloud else 0.0, any_cloud))
    return out
