// logging.cpp -- cwipc_log and the error-buffer convention.
//
// Behaviour follows reference src/logging.cpp:
//   - default level WARNING, default sink stderr                    (:19, :25)
//   - CWIPC_LOGGING=LEVEL[:file] switches to a file/stderr stream   (:48-72)
//   - cwipc_log_configure(level, callback): NONE keeps the level, a
//     callback replaces stderr                                       (:74-92)
//   - message format "module: Level: text", "t=<sec>: " prefix on
//     stream sinks only                                              (:98-129)
//   - the first ERROR during a call with an error buffer installed is
//     strdup'd into it                                               (:113-116)
// Unlike the reference the state is guarded by a mutex and the error buffer is
// per thread: ctypes releases the GIL, so filters run concurrently
// (reference python/cwipc/net/source_synchronizer.py:17,184).
#include "internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <mutex>

namespace {

std::mutex g_log_mutex;
cwipc_log_level g_level = CWIPC_LOG_LEVEL_WARNING;
bool g_initialized = false;
std::ostream *g_stream = nullptr;
cwipc_log_callback_t g_callback = nullptr;
bool g_to_callback = false, g_to_stderr = true, g_to_file = false;
time_t g_start = 0;
thread_local char **t_errorbuf = nullptr;

const char *level_name(cwipc_log_level level) {
    switch (level) {
    case CWIPC_LOG_LEVEL_ERROR: return "Error";
    case CWIPC_LOG_LEVEL_WARNING: return "Warning";
    case CWIPC_LOG_LEVEL_TRACE: return "Trace";
    case CWIPC_LOG_LEVEL_DEBUG: return "Debug";
    default: return "Unknown-level";
    }
}

cwipc_log_level parse_level(const std::string &s) {
    if (s == "NONE") return CWIPC_LOG_LEVEL_NONE;
    if (s == "ERROR") return CWIPC_LOG_LEVEL_ERROR;
    if (s == "WARNING") return CWIPC_LOG_LEVEL_WARNING;
    if (s == "TRACE") return CWIPC_LOG_LEVEL_TRACE;
    if (s == "DEBUG") return CWIPC_LOG_LEVEL_DEBUG;
    return CWIPC_LOG_LEVEL_WARNING;
}

// caller holds g_log_mutex
void init_locked() {
    if (g_initialized) return;
    g_initialized = true;
    const char *env = getenv("CWIPC_LOGGING");
    if (!env) return;
    std::string spec(env), file;
    size_t colon = spec.find(':');
    if (colon != std::string::npos) {
        file = spec.substr(colon + 1);
        spec = spec.substr(0, colon);
    }
    g_level = parse_level(spec);
    g_stream = file.empty() ? &std::cerr : new std::ofstream(file, std::ios::out | std::ios::app);
    g_to_stderr = false;
    g_to_file = true;
}

}  // namespace

extern "C" void cwipc_log_configure(int level, cwipc_log_callback_t callback) {
    {
        std::lock_guard<std::mutex> lock(g_log_mutex);
        init_locked();
        if (level != CWIPC_LOG_LEVEL_NONE) g_level = static_cast<cwipc_log_level>(level);
        g_callback = callback;
        g_to_callback = callback != nullptr;
        g_to_stderr = callback ? false : !g_to_file;
    }
    if (cwipc_log_get_level() >= CWIPC_LOG_LEVEL_DEBUG) {
        cwipc_log(CWIPC_LOG_LEVEL_DEBUG, "logging", "Logging configured, (int)callback=" + std::to_string((intptr_t)callback));
    }
}

extern "C" void cwipc_log(cwipc_log_level level, std::string module, std::string message) {
    cwipc_log_callback_t cb = nullptr;
    std::string full;
    {
        std::lock_guard<std::mutex> lock(g_log_mutex);
        init_locked();
        if (level > g_level) return;
        full = module + ": " + level_name(level) + ": " + message;
        if (g_start == 0) g_start = time(nullptr);
        std::string stamp = "t=" + std::to_string((long)(time(nullptr) - g_start)) + ": ";
        if (t_errorbuf && level == CWIPC_LOG_LEVEL_ERROR && *t_errorbuf == nullptr) {
            *t_errorbuf = strdup(full.c_str());   // handed to the caller, as in the reference
        }
        if (g_to_stderr) std::cerr << stamp << full << std::endl;
        if (g_to_file && g_stream) {
            (*g_stream) << stamp << full << std::endl;
            g_stream->flush();
        }
        if (g_to_callback) cb = g_callback;
    }
    if (cb) cb(level, full.c_str());   // outside the lock: the callback may log
}

extern "C" void _cwipc_log_emit(int level, const char *module, const char *message) {
    cwipc_log(static_cast<cwipc_log_level>(level), module ? module : "", message ? message : "");
}

extern "C" void cwipc_log_set_errorbuf(char **errorbuf) {
    t_errorbuf = errorbuf;
}

extern "C" cwipc_log_level cwipc_log_get_level() {
    std::lock_guard<std::mutex> lock(g_log_mutex);
    init_locked();
    return g_level;
}

namespace cwipc_amd {

bool api_version_rejected(const char *fname, uint64_t apiVersion, char **errorMessage) {
    if (apiVersion >= CWIPC_API_VERSION_OLD && apiVersion <= CWIPC_API_VERSION) return false;
    if (errorMessage) {
        char *msg = (char *)malloc(1024);
        snprintf(msg, 1024, "%s: incorrect apiVersion 0x%08llx expected 0x%08llx..0x%08llx", fname,
                 (unsigned long long)apiVersion, (unsigned long long)CWIPC_API_VERSION_OLD,
                 (unsigned long long)CWIPC_API_VERSION);
        *errorMessage = msg;
    }
    return true;
}

}  // namespace cwipc_amd
