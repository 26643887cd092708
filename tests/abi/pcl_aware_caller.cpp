// A caller built the way a PCL-aware sibling library of the reference is built (reference
// include/cwipc_util/api_pcl.h:74: `typedef pcl::shared_ptr<...> cwipc_pcl_pointcloud;`, which with current PCL is
// std::shared_ptr): it sees cwipc_pointcloud::access_pcl_pointcloud() as a virtual that returns a shared_ptr BY VALUE,
// i.e. through a hidden return slot.  The library, built without PCL, must fill that slot with the image of an
// empty shared_ptr ("access_pcl_pointcloud() hazard"), and every other virtual must sit in the slot the
// reference's class layout gives it (reference include/cwipc_util/api.h:184-284).
#include <cstdio>
#include <cstring>
#include <memory>

struct FakePclCloud { int dummy; };
typedef std::shared_ptr<FakePclCloud> cwipc_pcl_pointcloud;
#define _CWIPC_PCL_POINTCLOUD_DEFINED
#include "cwipc_util/api.h"

int main() {
    cwipc_point pts[3] = {{1.f, 2.f, 3.f, 10, 20, 30, 1}, {4.f, 5.f, 6.f, 40, 50, 60, 2}, {7.f, 8.f, 9.f, 70, 80, 90, 4}};
    char *err = nullptr;
    cwipc_pointcloud *pc = cwipc_from_points(pts, sizeof(pts), 3, 4242, &err, CWIPC_API_VERSION);
    if (!pc) { std::printf("FAIL from_points: %s\n", err ? err : "?"); return 1; }
    int bad = 0;
    // neighbours of the slot under test, through the vtable
    if (pc->timestamp() != 4242) { std::printf("FAIL timestamp slot\n"); bad++; }
    if (pc->count() != 3) { std::printf("FAIL count slot\n"); bad++; }
    if (pc->get_uncompressed_size() != sizeof(pts)) { std::printf("FAIL get_uncompressed_size slot\n"); bad++; }
    {
        // poison the return slot's surroundings: a callee that treated the slot as a plain pointer return would leave
        // the second word untouched or write past it
        struct { unsigned long long before; cwipc_pcl_pointcloud cloud; unsigned long long after; } frame;
        frame.before = 0x1111111111111111ull;
        frame.after = 0x2222222222222222ull;
        frame.cloud = pc->access_pcl_pointcloud();
        if (frame.cloud != nullptr || frame.cloud.use_count() != 0) { std::printf("FAIL access_pcl_pointcloud: not an empty shared_ptr\n"); bad++; }
        if (frame.before != 0x1111111111111111ull || frame.after != 0x2222222222222222ull) { std::printf("FAIL return slot overrun\n"); bad++; }
        cwipc_pcl_pointcloud direct = pc->access_pcl_pointcloud();   // copy elision: the callee constructs in place
        unsigned char zero[sizeof(direct)] = {0};
        if (sizeof(direct) != 16 || std::memcmp(&direct, zero, sizeof(direct)) != 0) { std::printf("FAIL slot image is not {nullptr, nullptr}\n"); bad++; }
    }   // the shared_ptr destructors run here: an image with a control block pointer would crash
    // the slot after it
    if (pc->access_metadata() == nullptr) { std::printf("FAIL access_metadata slot\n"); bad++; }
    cwipc_point back[3];
    if (pc->copy_uncompressed(back, sizeof(back)) != 3 || std::memcmp(back, pts, sizeof(pts)) != 0) { std::printf("FAIL copy_uncompressed slot\n"); bad++; }
    pc->free();
    if (cwipc_dangling_allocations(false) != 0) { std::printf("FAIL dangling allocations\n"); bad++; }
    if (!bad) std::printf("OK\n");
    return bad;
}
