// stubs.cpp -- exported constructors of subsystems that are OUT OF SCOPE for the
// MI355X filter path (PLY I/O needs PCL's reader, capture
// needs camera plugins, the window needs glfw/OpenGL, the proxy is TCP transport).
// The reference's ctypes wrapper binds these symbols eagerly at load time
// (python/cwipc/util.py:398-400, 516-528), so they must exist; each one fails
// loudly through the reference's own error convention (errorMessage + NULL/-1).
#include "internal.hpp"

#include <cstdlib>
#include <cstring>

using namespace cwipc_amd;

static void fail(const char *who, const char *why, char **errorMessage) {
    cwipc_log_set_errorbuf(errorMessage);
    cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, why);
    cwipc_log_set_errorbuf(nullptr);
}

extern "C" cwipc_pointcloud *cwipc_read(const char *filename, uint64_t, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_read", apiVersion, errorMessage)) return nullptr;
    fail("cwipc_read", (std::string("Loading of PLY file failed: ") + (filename ? filename : "(null)") +
                        " (PLY I/O is not part of the MI355X filter-path build; use cwipc_read_debugdump)").c_str(), errorMessage);
    return nullptr;
}

extern "C" int cwipc_write(const char *filename, cwipc_pointcloud *, char **errorMessage) {
    fail("cwipc_write", (std::string("Saving of PLY file failed: ") + (filename ? filename : "(null)") +
                         " (PLY I/O is not part of the MI355X filter-path build; use cwipc_write_debugdump)").c_str(), errorMessage);
    return -1;
}

extern "C" int cwipc_write_ext(const char *filename, cwipc_pointcloud *, int, char **errorMessage) {
    fail("cwipc_write_ext", (std::string("Saving of PLY file failed: ") + (filename ? filename : "(null)") +
                             " (PLY I/O is not part of the MI355X filter-path build; use cwipc_write_debugdump)").c_str(), errorMessage);
    return -1;
}

extern "C" cwipc_activesource *cwipc_capturer(const char *, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_capturer", apiVersion, errorMessage)) return nullptr;
    fail("cwipc_capturer", "no capturer plugins in the MI355X filter-path build", errorMessage);
    return nullptr;
}

extern "C" cwipc_sink *cwipc_window(const char *, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_window", apiVersion, errorMessage)) return nullptr;
    fail("cwipc_window", "no GUI support in the MI355X filter-path build", errorMessage);
    return nullptr;
}

extern "C" cwipc_activesource *cwipc_proxy(const char *, int, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_proxy", apiVersion, errorMessage)) return nullptr;
    fail("cwipc_proxy", "no TCP proxy in the MI355X filter-path build", errorMessage);
    return nullptr;
}
