// A native caller of the library with the command line of the reference's cwipc_downsample tool (apps/cwipc_downsample,
// ctest `cwipc_downsample 0.1 in.ply out.ply`): PLY in -> cwipc_downsample -> PLY out, through the C entry points and the C++
// virtuals of include/cwipc_util/api.h exactly as a program linked against the reference library would call them.
// Exit codes: 0 ok, 1 I/O or filter failure, 2 usage, 3 objects left alive.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "cwipc_util/api.h"

static int fail(const char *what, const char *detail) {
    std::fprintf(stderr, "downsample_app: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
    return 1;
}

int main(int argc, char **argv) {
    if (argc != 4 && argc != 5) {
        std::fprintf(stderr, "usage: %s voxelsize in.ply out.ply [binary]\n", argv[0]);
        return 2;
    }
    const float voxelsize = (float)std::atof(argv[1]);
    const bool binary = argc == 5 && std::strcmp(argv[4], "binary") == 0;
    char *message = nullptr;
    cwipc_pointcloud *in = cwipc_read(argv[2], 4711, &message, CWIPC_API_VERSION);
    if (in == nullptr) return fail("cannot read the input cloud", message);
    std::fprintf(stderr, "downsample_app: %d points in, %zu bytes\n", in->count(), in->get_uncompressed_size());
    cwipc_pointcloud *out = cwipc_downsample(in, voxelsize);
    int rc = 0;
    if (out == nullptr) {
        rc = fail("the filter returned no cloud", nullptr);
    } else {
        std::fprintf(stderr, "downsample_app: %d points out, cellsize %g, timestamp %llu\n", out->count(), out->cellsize(),
                     (unsigned long long)out->timestamp());
        message = nullptr;
        const int wrote = binary ? cwipc_write_ext(argv[3], out, CWIPC_FLAG_BINARY, &message) : cwipc_write(argv[3], out, &message);
        if (wrote < 0) rc = fail("cannot write the result", message);
        out->free();
    }
    in->free();
    if (rc == 0 && cwipc_dangling_allocations(true) != 0) rc = 3;
    return rc;
}
