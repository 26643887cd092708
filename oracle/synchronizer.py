"""CPU restatement of the reference's tile synchroniser policy -- TEST INFRASTRUCTURE ONLY.

Follows python/cwipc/net/source_synchronizer.py:106-200 (`_Synchronizer.run`) line by line, as one function
over scripted sources: no thread, no clock, and the fused cloud is the reference's left fold of
pairwise joins (:175-184) over numpy records (oracle.join).  Only tests may import this module.

Pinned?  The reference holds no test or fixture for its synchroniser; this restatement is checked against
the loop's text only ("parity unpinned" for the policy), the join it folds is pinned (oracle.join).
"""
from typing import Callable, List, Optional, Tuple


class ScriptedSource:
    """A source that hands out `clouds` in order; `gates[k]` polls must pass before cloud k is available."""

    def __init__(self, clouds: list, gates: Optional[List[int]] = None):
        self.clouds = list(clouds)
        self.gates = list(gates) if gates is not None else [0] * len(self.clouds)
        self.next = 0
        self.polls = 0

    def free(self) -> None:
        pass

    def eof(self) -> bool:
        return self.next >= len(self.clouds)

    def available(self, wait: bool) -> bool:
        if self.eof():
            return False
        self.polls += 1
        return self.polls > self.gates[self.next]

    def get(self):
        pc = self.clouds[self.next]
        self.next += 1
        self.polls = 0
        return pc

    def statistics(self) -> None:
        pass


def run_reference_loop(sources: list, join2: Callable, prefer_partial_over_unsynced: bool = True, max_iterations: int = 100000
                       ) -> Tuple[list, dict]:
    """The body of `_Synchronizer.run` until a source reports end of file.  Returns the produced clouds
    (as (timestamp, cellsize, payload) with payload = fold of join2 over the combined clouds' payloads,
    or the single cloud's payload itself) and the statistics lists."""
    n_tile = len(sources)
    input_buffers = [None] * n_tile
    earliest_timestamp = 0
    latest_timestamp = 0
    late, desyncs, missing = [], [], []
    produced = []
    for _ in range(max_iterations):
        if any(s.eof() for s in sources):                                   # :113-116
            break
        for head in input_buffers:                                         # :119-123
            if head:
                latest_timestamp = max(latest_timestamp, head.timestamp())
        for i in range(n_tile):                                            # :124-129
            buf = input_buffers[i]
            if buf:
                if buf.timestamp() < earliest_timestamp:
                    input_buffers[i] = None
        any_empty_input_buffers = False                                    # :130-154
        for i in range(n_tile):
            if input_buffers[i] == None:   # noqa: E711 (as upstream)
                if sources[i].available(False):
                    pc = sources[i].get()
                    if not pc:
                        any_empty_input_buffers = True
                        break
                    if pc.timestamp() >= earliest_timestamp:
                        input_buffers[i] = pc
                    else:
                        late.append(earliest_timestamp - pc.timestamp())
                        any_empty_input_buffers = True
                else:
                    any_empty_input_buffers = True
        if any_empty_input_buffers:                                        # :155-159
            continue
        current_timestamps = [pc.timestamp() for pc in input_buffers if pc]
        current_earliest_timestamp = min(current_timestamps)
        current_latest_timestamp = max(current_timestamps)
        if prefer_partial_over_unsynced:                                    # :166-171
            to_combine = [pc for pc in input_buffers if pc and pc.timestamp() == current_earliest_timestamp]
            desync = 0
        else:
            to_combine = [pc for pc in input_buffers]
            desync = current_latest_timestamp - current_earliest_timestamp
        if len(to_combine) < n_tile:
            missing.append(n_tile - len(to_combine))
        if desync > 0:
            desyncs.append(desync)
        current_cellsize = min([pc.cellsize() for pc in to_combine if pc])  # :177
        result = None
        for pc in to_combine:                                              # :178-184
            if result is None:
                result = pc.payload()
            else:
                result = join2(result, pc.payload())
        produced.append((current_earliest_timestamp, current_cellsize, result))
        earliest_timestamp = current_earliest_timestamp + 1                 # :190
    return produced, {"late": late, "desync": desyncs, "missing": missing}
