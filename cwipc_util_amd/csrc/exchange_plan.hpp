// exchange_plan.hpp -- the multi-GPU join's plan for one frame, as a pure function of what the ranks told each other.
//
// No HIP, no RCCL, no device: exchange.cpp feeds it the gathered records and issues what it says; the CPU tests run it for
// every rank of a frame (cwipc_hip_exchange_plan, tests/test_exchange_plan.py) and check that the ranks' plans fit together --
// every receive has exactly one send of the same length, in the same order per pair of ranks -- and that carrying them out
// yields the left fold of cwipc_join over the tiles that arrived (reference src/cwipc_filters.cpp:388-418 folded by
// python/cwipc/net/source_synchronizer.py:175-188; tiles that are late are simply not part of the frame, :163-171).
//
// The rule that makes the plans fit: EVERY decision is a function of the gathered records alone (which are the same on all
// ranks), never of anything only one rank knows.  What a single rank finds out on its own -- it cannot use its device, it
// cannot get memory for the fused cloud -- goes into its record (`status`) and is thereby known to everybody before any
// payload moves.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

namespace cwipc_amd {
namespace xplan {

// What every rank tells the others about its part of the frame: 8 words (one ncclAllGather of 32 bytes per rank).
struct FrameMeta {
    uint32_t count;          // points of this rank's tile (0: empty tile or none)
    uint32_t has_cloud;      // 1: there is a tile (its timestamp and cellsize take part in the minimum), 0: none this frame
    uint32_t cellsize_bits;  // float
    uint32_t status;         // ST_*: what this rank can do this frame
    uint32_t ts_lo, ts_hi;
    uint32_t capacity;       // points the result buffer this rank already holds has room for (allocated BEFORE the gather)
    uint32_t pad;
};
static_assert(sizeof(FrameMeta) == 32, "FrameMeta travels as 8 uint32");

enum : uint32_t {
    ST_OK = 0,
    ST_ABSENT = 1,    // the rank takes no part in this frame's payload: it sends nothing and receives nothing, the others see
                      //   a frame without its tile (its own call fails, logged there)
    ST_NO_RECV = 2,   // the rank's tile is part of the frame and it sends it, but it has no memory for the fused cloud:
                      //   nobody sends to it (its own call fails, logged there)
};

struct Transfer {
    int peer;
    size_t n;        // points (each transfer is four plane messages of n elements: x, y, z, rgbt -- in that order)
    size_t offset;   // receives: where in the result's planes the peer's points go; sends: 0
};

struct FramePlan {
    std::vector<size_t> disp;   // W + 1 prefix sums of the counts that take part
    size_t total = 0;
    bool any = false;           // some tile arrived: ts_min / cs_min are meaningful
    uint64_t ts_min = 0;
    float cs_min = 0;
    bool too_big = false;       // 2^32 points or more: every rank fails alike, nothing moves
    bool no_result = false;     // this rank's status says it has no fused cloud this frame
    bool share_input = false;   // the fused cloud IS this rank's input (all points are its own): no result buffer needed
    bool own_copy = false;      // this rank's own part is copied into the result by a kernel (not through the wire)
    std::vector<Transfer> sends, recvs;
};

// A rank's count as the others use it: a rank that is out of the frame contributes nothing.
inline uint32_t effective_count(const FrameMeta &m) { return m.status == ST_ABSENT ? 0u : m.count; }
inline bool effective_cloud(const FrameMeta &m) { return m.status != ST_ABSENT && m.has_cloud != 0; }

inline size_t frame_total(int W, const FrameMeta *all) {
    size_t t = 0;
    for (int r = 0; r < W; r++) t += effective_count(all[r]);
    return t;
}

// Does rank r need a result buffer of its own for this frame?  Not if it has no result, not if the frame is empty (an empty
// cloud needs no room), not if all points are its own (it hands its input on) -- unless `loopback` (a one-GPU exercise of the
// wire: everything travels, the rank's own part too).
inline bool needs_buffer(int r, int W, const FrameMeta *all, bool loopback) {
    if (all[r].status != ST_OK) return false;
    const size_t total = frame_total(W, all);
    if (total == 0 || total >= ((size_t)1 << 32)) return false;
    if (!loopback && effective_count(all[r]) == total) return false;
    return true;
}

// After the first gather: must the ranks meet again (one more 4-byte gather of status words) before payload moves?  Only if
// some rank that needs a result buffer does not hold one that is big enough yet: it then has to allocate, and whether that
// worked is news to the others.  A function of the records alone, so all ranks agree on whether there is a second round.
inline bool needs_second_round(int W, const FrameMeta *all, bool loopback) {
    const size_t total = frame_total(W, all);
    for (int r = 0; r < W; r++)
        if (needs_buffer(r, W, all, loopback) && all[r].capacity < total) return true;
    return false;
}

// The plan of rank `rank`.  `all` must be the records as they stand when payload is about to move (status words of a second
// round merged in).
inline FramePlan plan_frame(int rank, int W, const FrameMeta *all, bool loopback) {
    FramePlan p;
    p.disp.assign(W + 1, 0);
    for (int r = 0; r < W; r++) {
        p.disp[r + 1] = p.disp[r] + effective_count(all[r]);
        if (!effective_cloud(all[r])) continue;
        const uint64_t ts = ((uint64_t)all[r].ts_hi << 32) | all[r].ts_lo;
        float cs;
        memcpy(&cs, &all[r].cellsize_bits, 4);
        if (!p.any || ts < p.ts_min) p.ts_min = ts;
        if (!p.any || cs < p.cs_min) p.cs_min = cs;     // (std::min's rule: a NaN in front stays, as in the reference's fold)
        p.any = true;
    }
    p.total = p.disp[W];
    p.no_result = all[rank].status != ST_OK;
    if (p.total >= ((size_t)1 << 32)) { p.too_big = true; return p; }
    if (p.total == 0) return p;
    const size_t n_me = effective_count(all[rank]);
    const bool i_send = all[rank].status != ST_ABSENT && n_me > 0;
    const bool i_recv = all[rank].status == ST_OK;
    p.share_input = i_recv && !loopback && n_me == p.total;
    p.own_copy = i_recv && !loopback && n_me > 0 && !p.share_input;
    for (int peer = 0; peer < W; peer++) {
        if (peer == rank && !loopback) continue;
        // my points to every rank that builds a fused cloud -- whether or not I keep a copy of my own (a rank whose cloud
        // is the whole frame still owes it to the others)
        if (i_send && all[peer].status == ST_OK) p.sends.push_back(Transfer{peer, n_me, 0});
        // their points to me
        const size_t n = effective_count(all[peer]);
        if (i_recv && n > 0) p.recvs.push_back(Transfer{peer, n, p.disp[peer]});
    }
    return p;
}

}  // namespace xplan
}  // namespace cwipc_amd
