// stubs.cpp -- exported constructors of subsystems that are OUT OF SCOPE for the
// MI355X filter path (PLY I/O needs PCL's reader, capture
// needs camera plugins, the window needs glfw/OpenGL, the proxy is TCP transport).
// The reference's ctypes wrapper binds these symbols eagerly at load time
// (python/cwipc/util.py:398-400, 516-528), so they must exist; each one fails
// loudly through the reference's own error convention (errorMessage + NULL/-1).
#include "internal.hpp"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace cwipc_amd;

static void fail(const char *who, const char *why, char **errorMessage) {
    cwipc_log_set_errorbuf(errorMessage);
    cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, why);
    cwipc_log_set_errorbuf(nullptr);
}

// ---------------------------------------------------------------------------
// capturer registry (reference src/cwipc_capturer.cpp:23-160)
// ---------------------------------------------------------------------------
// Camera plugins (cwipc_realsense2, cwipc_kinect ...) are sibling libraries that register a factory
// here when they are loaded; cwipc_capturer() dispatches on the "type" of the camera configuration.
// No capturer lives in this library, but the registry does, so those plugins keep working against it.
namespace {
struct Capturer {
    std::string name;
    _cwipc_functype_count_devices *countFunc;
    _cwipc_func_capturer_factory *factoryFunc;
};
std::mutex g_capturer_mutex;
std::vector<Capturer> g_capturers;

// The value of the top-level "type" member of a JSON object, without a JSON library: the first
// "type" key at nesting depth 1 whose value is a string.  Empty if there is none.
std::string json_type_of(const std::string &text) {
    int depth = 0;
    bool in_string = false;
    std::string token;
    for (size_t i = 0; i < text.size(); i++) {
        const char ch = text[i];
        if (in_string) {
            if (ch == '\\') { i++; continue; }
            if (ch == '"') {
                in_string = false;
                if (depth == 1 && token == "type") {
                    size_t j = i + 1;
                    while (j < text.size() && isspace((unsigned char)text[j])) j++;
                    if (j < text.size() && text[j] == ':') {
                        j++;
                        while (j < text.size() && isspace((unsigned char)text[j])) j++;
                        if (j < text.size() && text[j] == '"') {
                            const size_t end = text.find('"', j + 1);
                            if (end != std::string::npos) return text.substr(j + 1, end - j - 1);
                        }
                    }
                }
            } else {
                token.push_back(ch);
            }
        } else if (ch == '"') {
            in_string = true;
            token.clear();
        } else if (ch == '{' || ch == '[') {
            depth++;
        } else if (ch == '}' || ch == ']') {
            depth--;
        }
    }
    return "";
}
}  // namespace

extern "C" int _cwipc_register_capturer(const char *name, _cwipc_functype_count_devices *countFunc, _cwipc_func_capturer_factory *factoryFunc) {
    if (name == nullptr || factoryFunc == nullptr) return 0;
    std::lock_guard<std::mutex> lock(g_capturer_mutex);
    g_capturers.push_back(Capturer{name, countFunc, factoryFunc});
    return 1;
}

extern "C" cwipc_activesource *cwipc_capturer(const char *configFilename, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_capturer", apiVersion, errorMessage)) return nullptr;
    if (configFilename == nullptr || *configFilename == '\0') configFilename = "cameraconfig.json";
    std::vector<Capturer> capturers;
    {
        std::lock_guard<std::mutex> lock(g_capturer_mutex);
        capturers = g_capturers;
    }
    if (strcmp(configFilename, "auto") == 0) {
        // the one registered capturer whose hardware is present
        const Capturer *candidate = nullptr;
        for (const auto &c : capturers) {
            if (c.countFunc != nullptr && c.countFunc()) {
                if (candidate != nullptr) { fail("cwipc_capturer", "auto: multiple supported cameras found", errorMessage); return nullptr; }
                candidate = &c;
            }
        }
        if (candidate == nullptr) { fail("cwipc_capturer", "auto: no supported cameras found", errorMessage); return nullptr; }
        return candidate->factoryFunc(configFilename, errorMessage, apiVersion);
    }
    std::string json;
    if (configFilename[0] == '{') {
        json = configFilename;   // a string starting with { is a JSON literal
    } else {
        const char *extension = strrchr(configFilename, '.');
        if (extension == nullptr || strcmp(extension, ".json") != 0) {
            fail("cwipc_capturer", (std::string("auto: unknown config file type: ") + configFilename).c_str(), errorMessage);
            return nullptr;
        }
        FILE *f = fopen(configFilename, "rb");
        if (f == nullptr) {
            fail("cwipc_capturer", (std::string("auto: cannot open \"") + configFilename + "\"").c_str(), errorMessage);
            return nullptr;
        }
        char buf[4096];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), f)) > 0) json.append(buf, got);
        fclose(f);
    }
    const std::string type = json_type_of(json);
    if (type.empty()) {
        fail("cwipc_capturer", (std::string("auto: cannot determine camera type from \"") + configFilename + "\"").c_str(), errorMessage);
        return nullptr;
    }
    for (const auto &c : capturers) {
        if (c.name == type) return c.factoryFunc(configFilename, errorMessage, apiVersion);
    }
    fail("cwipc_capturer", (std::string("auto: camera type \"") + type + "\" not supported").c_str(), errorMessage);
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
