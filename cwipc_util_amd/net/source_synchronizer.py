"""Combine the point clouds of several (tiled) sources into one stream of fused clouds.

Counterpart of reference python/cwipc/net/source_synchronizer.py (the `_Synchronizer` thread and the
`cwipc_source_synchronizer` factory): same interface, same timestamp-matching policy, same statistics.
Two things are done differently:

  * the fused cloud is ONE n-ary join on the GPU (`cwipc_join_multi`), where the reference folds
    `cwipc_join` pairwise (source_synchronizer.py:175-184), copying O(n_tile^2) bytes;
  * the policy is a plain object (`SyncCore`, one call = one pass of the reference's loop body,
    source_synchronizer.py:113-192) that the thread drives, so that it can be checked step by step
    against the restatement in oracle/synchronizer.py without a clock.
"""
from __future__ import annotations

import queue
import threading
import time
from typing import Any, Callable, List, Optional, Sequence, Union

from ..abstract import cwipc_activesource_abstract, cwipc_pointcloud_abstract, cwipc_source_abstract, cwipc_tileinfo_dict

__all__ = ["SyncCore", "cwipc_source_synchronizer"]


def _join_on_gpu(clouds: Sequence[cwipc_pointcloud_abstract]) -> cwipc_pointcloud_abstract:
    from ..util import cwipc_join_multi
    return cwipc_join_multi(clouds)   # one cloud: that cloud itself, as the reference's fold returns it


class SyncCore:
    """The synchroniser's policy, without thread or clock.

    `poll()` is one pass of the reference's loop (source_synchronizer.py:113-192) after its end-of-file
    test: it refreshes the per-tile buffer heads from the sources and, when every tile has a head,
    returns the fused cloud; None means "nothing to produce yet" (the thread sleeps 1 ms and polls again).
    """

    def __init__(self, sources: Sequence[cwipc_source_abstract], join: Optional[Callable[[Sequence[Any]], Any]] = None,
                 prefer_partial_over_unsynced: bool = True, verbose: bool = False):
        self.sources = list(sources)
        self.n_tile = len(self.sources)
        self.input_buffers: List[Optional[cwipc_pointcloud_abstract]] = [None] * self.n_tile
        self.prefer_partial_over_unsynced = prefer_partial_over_unsynced
        self.join = join if join is not None else _join_on_gpu
        self.verbose = verbose
        self.earliest_timestamp = 0
        self.latest_timestamp = 0
        self.combine_times: List[float] = []
        self.late_per_occurrence: List[int] = []
        self.desync_per_occurrence: List[int] = []
        self.missing_per_occurrence: List[int] = []

    def poll(self) -> Optional[cwipc_pointcloud_abstract]:
        # the latest of the buffer heads (reference :120-124; kept for the statistics it feeds upstream)
        for head in self.input_buffers:
            if head:
                self.latest_timestamp = max(self.latest_timestamp, head.timestamp())
        # outdated heads go (:125-130)
        for i in range(self.n_tile):
            buf = self.input_buffers[i]
            if buf and buf.timestamp() < self.earliest_timestamp:
                self.input_buffers[i] = None
        # empty slots are filled from their sources (:131-154)
        any_empty = False
        for i in range(self.n_tile):
            if self.input_buffers[i] is None:
                if self.sources[i].available(False):
                    pc = self.sources[i].get()
                    if not pc:
                        print(f"synchronizer: source {i} returned no point cloud")
                        any_empty = True
                        break
                    if self.verbose:
                        print(f"synchronizer: got ts={pc.timestamp()} from {i}")
                    if pc.timestamp() >= self.earliest_timestamp:
                        self.input_buffers[i] = pc
                    else:
                        too_late = self.earliest_timestamp - pc.timestamp()
                        if self.verbose:
                            print(f"synchronizer: tile {i}: too late by {too_late}")
                        self.late_per_occurrence.append(too_late)
                        any_empty = True
                else:
                    any_empty = True
        if any_empty:
            return None
        # every tile has a head: which of them make the next cloud (:160-173)
        heads = [pc for pc in self.input_buffers if pc]
        current_timestamps = [pc.timestamp() for pc in heads]
        current_earliest = min(current_timestamps)
        current_latest = max(current_timestamps)
        if self.prefer_partial_over_unsynced:
            to_combine = [pc for pc in heads if pc.timestamp() == current_earliest]
            desync = 0
        else:
            to_combine = heads
            desync = current_latest - current_earliest
        if len(to_combine) < self.n_tile:
            self.missing_per_occurrence.append(self.n_tile - len(to_combine))
        if desync > 0:
            self.desync_per_occurrence.append(desync)
        # one join, rank order = tile order = the reference's fold order (:175-188)
        t0 = time.time()
        current_cellsize = min(pc.cellsize() for pc in to_combine)
        result = self.join(to_combine)
        self.combine_times.append(time.time() - t0)
        result._set_timestamp(current_earliest)
        result._set_cellsize(current_cellsize)
        self.earliest_timestamp = current_earliest + 1
        return result


class _Synchronizer(threading.Thread, cwipc_activesource_abstract):
    """A source that combines point clouds gotten from multiple (tiled) sources into one stream
    (reference `_Synchronizer`, source_synchronizer.py:16-98, 194-232)."""

    QUEUE_WAIT_TIMEOUT = 1

    def __init__(self, reader: Any, sources: List[cwipc_source_abstract], verbose: bool = False):
        threading.Thread.__init__(self)
        self.name = 'cwipc_util._NetDecoder'
        self.reader = reader
        self.sources = sources
        self.n_tile = len(sources)
        self.running = False
        self.verbose = verbose
        self.output_queue: "queue.Queue[Optional[cwipc_pointcloud_abstract]]" = queue.Queue(maxsize=6)
        self.core = SyncCore(sources, verbose=verbose)

    # the reference exposes these as attributes of the thread object
    @property
    def prefer_partial_over_unsynced(self) -> bool:
        return self.core.prefer_partial_over_unsynced

    @prefer_partial_over_unsynced.setter
    def prefer_partial_over_unsynced(self, value: bool) -> None:
        self.core.prefer_partial_over_unsynced = value

    @property
    def input_buffers(self) -> List[Optional[cwipc_pointcloud_abstract]]:
        return self.core.input_buffers

    def free(self) -> None:
        pass

    def start(self) -> bool:
        assert not self.running
        if self.verbose: print('synchronizer: start', flush=True)
        self.running = True
        # (the sources are not started here: they are not active sources, they may be decoders -- reference :50-53)
        if self.reader is not None and not self.reader.start():
            return False
        threading.Thread.start(self)
        return True

    def stop(self) -> None:
        if self.verbose: print('synchronizer: stop', flush=True)
        self.running = False
        if self.reader is not None:
            self.reader.stop()
        try:
            self.output_queue.put(None, block=False)
        except queue.Full:
            pass
        self.join()

    def eof(self) -> bool:
        if not self.running:
            return True
        if not self.output_queue.empty():
            return False
        return self._any_source_eof()

    def _any_source_eof(self) -> bool:
        return any(s.eof() for s in self.sources)

    def available(self, wait: bool = False) -> bool:
        if not self.running:
            return False
        if not self.output_queue.empty():
            return True
        return all(s.available(wait) for s in self.sources)

    def get(self) -> Optional[cwipc_pointcloud_abstract]:
        if self.eof():
            return None
        return self.output_queue.get()

    def run(self) -> None:
        if self.verbose: print("synchronizer: thread started", flush=True)
        while self.running:
            if self._any_source_eof():
                if self.verbose: print("synchronizer: end of file")
                break
            result = self.core.poll()
            if result is None:
                time.sleep(0.001)
                continue
            if self.verbose:
                latency = int(time.time() * 1000) - result.timestamp()
                print(f'synchronizer: produced pointcloud ts={result.timestamp()} with {result.count()} points, latency={latency} ms, '
                      f'qlen={self.output_queue.qsize()}', flush=True)
            while self.running:   # (a plain put() would keep stop() waiting for ever behind a full queue)
                try:
                    self.output_queue.put(result, timeout=0.05)
                    break
                except queue.Full:
                    continue
        if self.verbose: print("synchronizer: thread exiting", flush=True)
        self.running = False
        try:
            self.output_queue.put(None, block=False)
        except queue.Full:
            pass

    def statistics(self) -> None:
        self.print1stat('combine_time', self.core.combine_times)
        self.print1stat('late', self.core.late_per_occurrence)
        self.print1stat('desync', self.core.desync_per_occurrence)
        self.print1stat('missing', self.core.missing_per_occurrence, isInt=True)
        for s in self.sources:
            if hasattr(s, 'statistics'):
                s.statistics()

    def print1stat(self, name: str, values: Sequence[Union[int, float]], isInt: bool = False) -> None:
        count = len(values)
        if count == 0:
            print('netdecoder: {}: count=0'.format(name))
            return
        fmt = 'netdecoder: {}: count={}, average={:.3f}, min={:d}, max={:d}' if isInt else 'netdecoder: {}: count={}, average={:.3f}, min={:.3f}, max={:.3f}'
        print(fmt.format(name, count, sum(values) / count, min(values), max(values)))

    def request_metadata(self, name: str) -> None:
        assert False

    def is_metadata_requested(self, name: str) -> bool:
        return False

    # the rest of the active-source interface is not meaningful for a synchroniser (reference :251-277)
    def reload_config(self, config: Union[str, bytes, None]) -> Any:
        raise NotImplementedError

    def get_config(self) -> bytes:
        raise NotImplementedError

    def seek(self, timestamp: int) -> bool:
        raise NotImplementedError

    def auxiliary_operation(self, op: str, inbuf: bytes, outbuf: bytearray) -> bool:
        raise NotImplementedError

    def maxtile(self) -> int:
        raise NotImplementedError

    def get_tileinfo_dict(self, tilenum: int) -> cwipc_tileinfo_dict:
        raise NotImplementedError


def cwipc_source_synchronizer(reader: Any, sources: List[cwipc_source_abstract], verbose: bool = False) -> cwipc_activesource_abstract:
    """A cwipc_source-like object that combines and synchronises point clouds from several sources
    (reference source_synchronizer.py:279-282).  `reader` may be None when nothing has to be started."""
    return _Synchronizer(reader, sources, verbose=verbose)
