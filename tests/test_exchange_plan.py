"""The library's multi-GPU exchange (csrc/exchange.cpp) at W >= 2, without hardware.

cwipc_hip_comm_join issues its ncclSend / ncclRecv from a plan that is a pure function of the records the ranks gathered
(csrc/exchange_plan.hpp, exported as cwipc_hip_exchange_plan).  Here that very function is run for ALL ranks of a frame and
the plans are checked against each other -- every receive has exactly one send of the same length, per pair of ranks in the
same order, nothing is sent to a rank that does not receive -- and then carried out on host arrays: every rank that is
planned to have a fused cloud must end up with the left fold of cwipc_join over the tiles that took part (reference
src/cwipc_filters.cpp:388-418 folded by python/cwipc/net/source_synchronizer.py:175-188; tiles that are late are not part of
the frame, :163-171), timestamp and cellsize the minimum (:411-414).

Round 2's plan deadlocked when one rank held all of a frame's points (it left before the group while the others posted
receives from it): test_whole_frame_on_one_rank_is_still_sent is that case.
"""
import ctypes
import struct

import numpy as np
import pytest

ST_OK, ST_ABSENT, ST_NO_RECV = 0, 1, 2
F_TOO_BIG, F_NO_RESULT, F_SHARE, F_OWN_COPY, F_ANY, F_SECOND, F_NEEDS_BUF = 1, 2, 4, 8, 16, 32, 64


def _f32_bits(v):
    return struct.unpack("<I", struct.pack("<f", v))[0]


def _bits_f32(b):
    return struct.unpack("<f", struct.pack("<I", b & 0xffffffff))[0]


class Rank:
    """What one rank brings to a frame."""

    def __init__(self, tile=None, ts=0, cellsize=0.0, status=ST_OK, capacity=0, alloc_fails=False):
        self.tile = tile                # numpy array of point ids (stand-ins for the four planes), or None: no tile this frame
        self.ts, self.cellsize, self.status, self.capacity, self.alloc_fails = ts, cellsize, status, capacity, alloc_fails

    def record(self):
        has = self.tile is not None
        n = len(self.tile) if has else 0
        return [n, 1 if has else 0, _f32_bits(self.cellsize) if has else 0, self.status, self.ts & 0xffffffff if has else 0,
                (self.ts >> 32) if has else 0, self.capacity, 0]


def plan(dll, rank, metas, loopback=False, cap=64):
    W = len(metas)
    flat = (ctypes.c_uint32 * (8 * W))(*[w for m in metas for w in m])
    summary = (ctypes.c_uint64 * 8)()
    sends = (ctypes.c_uint64 * (3 * cap))()
    recvs = (ctypes.c_uint64 * (3 * cap))()
    dll.cwipc_hip_exchange_plan.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    dll.cwipc_hip_exchange_plan.restype = ctypes.c_int
    rc = dll.cwipc_hip_exchange_plan(rank, W, flat, 1 if loopback else 0, summary, sends, recvs, cap)
    assert rc == 0, rc
    s = list(summary)
    return {
        "total": s[0], "flags": s[1], "ts": s[2], "cs": _bits_f32(s[3]), "disp": s[4],
        "sends": [(sends[3 * i], sends[3 * i + 1], sends[3 * i + 2]) for i in range(s[5])],
        "recvs": [(recvs[3 * i], recvs[3 * i + 1], recvs[3 * i + 2]) for i in range(s[6])],
    }


def run_frame(dll, ranks, loopback=False):
    """The protocol of join_frame (csrc/exchange.cpp) for all ranks of one frame, carried out on host arrays.
    Returns per rank: the fused cloud (array) or None, plus (ts, cellsize)."""
    W = len(ranks)
    metas = [r.record() for r in ranks]
    # 1. first gather done: every rank holds `metas`.  Whether the ranks meet again is the same answer on every rank
    first = [plan(dll, r, metas, loopback) for r in range(W)]
    second = {bool(p["flags"] & F_SECOND) for p in first}
    assert len(second) == 1, "the ranks disagree on whether there is a second round"
    if second.pop():
        total = first[0]["total"]
        words = []
        for r in range(W):
            word = ranks[r].status
            if (first[r]["flags"] & F_NEEDS_BUF) and ranks[r].capacity < total:
                if ranks[r].alloc_fails:
                    word = ST_NO_RECV
                else:
                    ranks[r].capacity = total + total // 4 + 1024
            words.append(word)
        for r in range(W):
            if metas[r][3] == ST_OK:
                metas[r][3] = words[r]
    else:
        for r in range(W):   # nobody may be asked to allocate without the others hearing of it
            if first[r]["flags"] & F_NEEDS_BUF:
                assert ranks[r].capacity >= first[r]["total"]
    # 2. the plans
    plans = [plan(dll, r, metas, loopback) for r in range(W)]
    totals = {p["total"] for p in plans}
    assert len(totals) == 1
    total = totals.pop()
    if plans[0]["flags"] & F_TOO_BIG:
        assert all(p["flags"] & F_TOO_BIG and not p["sends"] and not p["recvs"] for p in plans)
        return [None] * W, plans
    # 3. the plans fit together: the messages rank a sends to rank b are, in order, what b expects from a
    for a in range(W):
        for b in range(W):
            sent = [n for (peer, n, _o) in plans[a]["sends"] if peer == b]
            expected = [n for (peer, n, _o) in plans[b]["recvs"] if peer == a]
            assert sent == expected, (a, b, sent, expected, metas)
            assert len(sent) <= 1
            if a == b and not loopback:
                assert not sent
    # 4. carried out
    results = []
    for r in range(W):
        p = plans[r]
        if p["flags"] & F_NO_RESULT:
            assert not p["recvs"], "a rank without a result must not be sent anything"
            results.append(None)
            continue
        if total == 0:
            assert not p["sends"] and not p["recvs"]
            results.append(np.zeros(0, dtype=np.int64))
            continue
        if p["flags"] & F_SHARE:
            assert not p["recvs"] and not (p["flags"] & F_OWN_COPY)
            results.append(ranks[r].tile)
            continue
        assert ranks[r].capacity >= total, "planned to receive into a buffer that is too small"
        buf = np.full(ranks[r].capacity, -1, dtype=np.int64)
        written = np.zeros(ranks[r].capacity, dtype=np.int32)
        if p["flags"] & F_OWN_COPY:
            n = len(ranks[r].tile)
            buf[p["disp"]:p["disp"] + n] = ranks[r].tile
            written[p["disp"]:p["disp"] + n] += 1
        for (peer, n, off) in p["recvs"]:
            assert off + n <= total
            assert n == len(ranks[peer].tile)
            buf[off:off + n] = ranks[peer].tile
            written[off:off + n] += 1
        assert (written[:total] == 1).all() and (written[total:] == 0).all(), "every point of the fused cloud is written exactly once"
        results.append(buf[:total])
    return results, plans


def expected_frame(ranks):
    parts = [r.tile for r in ranks if r.tile is not None and r.status != ST_ABSENT]
    took_part = [r for r in ranks if r.tile is not None and r.status != ST_ABSENT]
    fused = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
    ts = min((r.ts for r in took_part), default=0)
    cs = min((np.float32(r.cellsize) for r in took_part), default=np.float32(0))
    return fused, ts, float(cs)


def check_frame(dll, ranks, loopback=False):
    want, ts, cs = expected_frame(ranks)
    results, plans = run_frame(dll, ranks, loopback)
    for r, got in enumerate(results):
        if ranks[r].status != ST_OK or (ranks[r].alloc_fails and plans[r]["flags"] & F_NO_RESULT):
            assert got is None or ranks[r].status == ST_OK
            continue
        assert got is not None, r
        assert np.array_equal(got, want), (r, got, want)
        if plans[r]["flags"] & F_ANY:
            assert plans[r]["ts"] == ts and plans[r]["cs"] == cs
    return results, plans


def tiles(counts, start=0):
    out, at = [], start
    for n in counts:
        out.append(None if n is None else np.arange(at, at + n, dtype=np.int64))
        at += n or 0
    return out


@pytest.fixture(scope="module")
def dll(cwipc):
    return cwipc.cwipc_util_dll_load()


@pytest.mark.parametrize("W", [2, 3, 8])
def test_whole_frame_on_one_rank_is_still_sent(dll, W):
    """Round 2's deadlock: one rank holds every point of the frame.  It keeps its own planes as the result AND sends them."""
    for holder in range(W):
        counts = [0] * W
        counts[holder] = 1000
        for missing_as_none in (False, True):
            ts_ = tiles([c if (c or not missing_as_none) else None for c in counts])
            ranks = [Rank(t, ts=100 + i, cellsize=0.01 * (i + 1)) for i, t in enumerate(ts_)]
            results, plans = check_frame(dll, ranks)
            assert plans[holder]["flags"] & F_SHARE
            assert sorted(peer for (peer, _n, _o) in plans[holder]["sends"]) == [r for r in range(W) if r != holder]
            for r in range(W):
                if r != holder:
                    assert plans[r]["recvs"] == [(holder, 1000, 0)]


@pytest.mark.parametrize("W", [1, 2, 3, 8])
def test_empty_and_missing_frames(dll, W):
    # all empty clouds: an empty fused cloud with the minimum timestamp, nothing moves
    ranks = [Rank(np.zeros(0, dtype=np.int64), ts=50 + i, cellsize=0.5) for i in range(W)]
    results, plans = check_frame(dll, ranks)
    assert all(len(r) == 0 for r in results) and all(not p["sends"] and not p["recvs"] for p in plans)
    assert plans[0]["ts"] == 50
    # no tile anywhere
    results, plans = check_frame(dll, [Rank(None) for _ in range(W)])
    assert all(len(r) == 0 for r in results) and not (plans[0]["flags"] & F_ANY)
    # a cloud missing here and there
    for miss in range(W):
        counts = [None if r == miss else 10 + r for r in range(W)]
        check_frame(dll, [Rank(t, ts=9, cellsize=1.0) for t in tiles(counts)])


@pytest.mark.parametrize("W", [2, 3, 8])
def test_random_frames(dll, W):
    rng = np.random.default_rng(20260129 + W)
    for _ in range(300):
        counts = []
        for r in range(W):
            u = rng.random()
            counts.append(None if u < 0.15 else 0 if u < 0.35 else int(rng.integers(1, 5000)))
        if rng.random() < 0.1:      # everything on one rank
            keep = int(rng.integers(W))
            counts = [c if r == keep else (0 if rng.random() < 0.5 else None) for r, c in enumerate(counts)]
        total = sum(c or 0 for c in counts)
        ranks = []
        for r, t in enumerate(tiles(counts)):
            status = ST_ABSENT if rng.random() < 0.1 else ST_OK
            capacity = int(rng.choice([0, total // 2, total, total + 100, 10 ** 6]))
            ranks.append(Rank(t, ts=int(rng.integers(1, 1 << 40)), cellsize=float(rng.random()), status=status, capacity=capacity,
                              alloc_fails=rng.random() < 0.15))
        check_frame(dll, ranks)


@pytest.mark.parametrize("W", [2, 3, 8])
def test_a_rank_that_cannot_allocate_is_left_out_by_everybody(dll, W):
    """Round 2: a rank without memory for the fused cloud left after the gather while the others sent to it.  Now the others
    know before payload moves: it still sends its tile, nobody sends to it, everyone else gets the whole frame."""
    counts = [100 * (r + 1) for r in range(W)]
    for broke in range(W):
        ranks = [Rank(t, ts=7, cellsize=0.1, capacity=0, alloc_fails=(r == broke)) for r, t in enumerate(tiles(counts))]
        results, plans = check_frame(dll, ranks)
        assert results[broke] is None and plans[broke]["flags"] & F_NO_RESULT
        assert len(plans[broke]["sends"]) == W - 1 and not plans[broke]["recvs"]
        for r in range(W):
            if r != broke:
                assert all(peer != broke for (peer, _n, _o) in plans[r]["sends"])
                assert len(results[r]) == sum(counts)


@pytest.mark.parametrize("W", [2, 8])
def test_second_round_only_when_somebody_has_to_allocate(dll, W):
    counts = [500] * W
    total = sum(counts)
    ranks = [Rank(t, ts=1, cellsize=1.0, capacity=total) for t in tiles(counts)]
    metas = [r.record() for r in ranks]
    assert not any(plan(dll, r, metas)["flags"] & F_SECOND for r in range(W))
    ranks[W - 1].capacity = total - 1
    metas = [r.record() for r in ranks]
    assert all(plan(dll, r, metas)["flags"] & F_SECOND for r in range(W))
    # a rank that hands its own input on needs no buffer, whatever it holds
    counts = [0] * W
    counts[0] = 77
    ranks = [Rank(t, ts=1, cellsize=1.0, capacity=(0 if r == 0 else 77)) for r, t in enumerate(tiles(counts))]
    metas = [r.record() for r in ranks]
    assert not any(plan(dll, r, metas)["flags"] & F_SECOND for r in range(W))
    check_frame(dll, ranks)


def test_an_absent_rank_changes_nothing_for_the_others(dll):
    """A rank that cannot use its device says so in its record: its tile is not part of the frame (as if it were late),
    it is sent nothing and the displacements close the gap."""
    counts = [10, 20, 30, 40]
    ranks = [Rank(t, ts=5 + r, cellsize=1.0 + r, status=(ST_ABSENT if r == 1 else ST_OK), capacity=1000) for r, t in enumerate(tiles(counts))]
    results, plans = check_frame(dll, ranks)
    assert results[1] is None and not plans[1]["sends"] and not plans[1]["recvs"]
    assert len(results[0]) == 80 and plans[2]["disp"] == 10 and plans[3]["disp"] == 40


def test_too_many_points_fail_alike(dll):
    metas = [[0xffffffff, 1, 0, 0, 1, 0, 0, 0], [1, 1, 0, 0, 1, 0, 0, 0]]
    for r in range(2):
        p = plan(dll, r, metas)
        assert p["flags"] & F_TOO_BIG and not p["sends"] and not p["recvs"]


def test_loopback_sends_to_itself(dll):
    ranks = [Rank(np.arange(50, dtype=np.int64), ts=3, cellsize=0.25, capacity=0)]
    results, plans = check_frame(dll, ranks, loopback=True)
    assert plans[0]["sends"] == [(0, 50, 0)] and plans[0]["recvs"] == [(0, 50, 0)] and not (plans[0]["flags"] & (F_SHARE | F_OWN_COPY))
