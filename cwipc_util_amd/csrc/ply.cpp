// ply.cpp -- cwipc_read / cwipc_write / cwipc_write_ext: PLY files, the wire format at either end of the filter path
// (SURVEY section 8f rank 4).  Reference: src/cwipc_util.cpp:432-497, which hands the work to pcl::PLYReader / pcl::PLYWriter
// on the point type of include/cwipc_util/api_pcl.h:20-70 (fields x, y, z, rgba; `a` is the tile).  PCL is not here, so the
// two ends are restated from the format PCL reads and writes for that point type:
//   writer  "ply / format ascii 1.0 | binary_little_endian 1.0 / comment PCL generated / element vertex N /
//           property float x, y, z / property uchar red, green, blue, alpha / element camera 1 / 21 camera properties /
//           end_header", then one vertex per line "x y z r g b a" (floats with 8 significant digits, as an ostream with
//           precision(8) prints them) or 16 bytes per vertex -- which IS a cwipc_point --, then the camera record (origin 0,
//           identity axes, viewport N x 1);
//   reader  any PLY whose vertex element has x, y, z and (optionally) red, green, blue, alpha / tile: ascii, binary little or
//           big endian, scalar properties of any PLY type, list properties and other elements skipped.
// Host code: files are the slow end of any pipeline, and the points come from / go to the host representation of the
// cloud (page-locked, so the next filter's upload reads it directly).
#include "internal.hpp"

#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

using namespace cwipc_amd;

namespace {

struct PlyProperty {
    std::string name;
    int type = 0;          // index into kTypes
    bool is_list = false;
    int count_type = 0;
};
struct PlyElement {
    std::string name;
    size_t count = 0;
    std::vector<PlyProperty> props;
};

const struct { const char *a, *b; int size; } kTypes[] = {
    {"char", "int8", 1}, {"uchar", "uint8", 1}, {"short", "int16", 2}, {"ushort", "uint16", 2},
    {"int", "int32", 4}, {"uint", "uint32", 4}, {"float", "float32", 4}, {"double", "float64", 8},
};

int type_of(const std::string &s) {
    for (int i = 0; i < 8; i++)
        if (s == kTypes[i].a || s == kTypes[i].b) return i;
    return -1;
}

double load_scalar(const unsigned char *p, int type, bool swap) {
    unsigned char b[8];
    const int n = kTypes[type].size;
    for (int i = 0; i < n; i++) b[i] = swap ? p[n - 1 - i] : p[i];
    switch (type) {
    case 0: { int8_t v; memcpy(&v, b, 1); return v; }
    case 1: { uint8_t v; memcpy(&v, b, 1); return v; }
    case 2: { int16_t v; memcpy(&v, b, 2); return v; }
    case 3: { uint16_t v; memcpy(&v, b, 2); return v; }
    case 4: { int32_t v; memcpy(&v, b, 4); return v; }
    case 5: { uint32_t v; memcpy(&v, b, 4); return v; }
    case 6: { float v; memcpy(&v, b, 4); return v; }
    default: { double v; memcpy(&v, b, 8); return v; }
    }
}

bool host_is_little_endian() {
    const uint16_t one = 1;
    return *reinterpret_cast<const unsigned char *>(&one) == 1;
}

void fail(const char *who, const std::string &why, char **errorMessage) {
    cwipc_log_set_errorbuf(errorMessage);
    cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, why);
    cwipc_log_set_errorbuf(nullptr);
}

// The vertices of a PLY file as cwipc_points.  false + reason on failure.
bool read_ply(const char *filename, std::vector<cwipc_point> &out, std::string &why) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) { why = std::string("cannot open: ") + strerror(errno); return false; }
    std::string line;
    if (!std::getline(f, line) || line.substr(0, 3) != "ply") { why = "not a PLY file"; return false; }
    int format = -1;   // 0 ascii, 1 little endian, 2 big endian
    std::vector<PlyElement> elements;
    bool header_done = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream is(line);
        std::string word;
        is >> word;
        if (word == "format") {
            std::string fmt;
            is >> fmt;
            format = fmt == "ascii" ? 0 : fmt == "binary_little_endian" ? 1 : fmt == "binary_big_endian" ? 2 : -1;
        } else if (word == "element") {
            PlyElement e;
            long long count = -1;
            is >> e.name >> count;
            if (is.fail() || count < 0) { why = "bad element line: " + line; return false; }   // (a count that does not parse must not read as an empty element)
            e.count = (size_t)count;
            elements.push_back(e);
        } else if (word == "property" && !elements.empty()) {
            PlyProperty p;
            std::string t;
            is >> t;
            if (t == "list") {
                std::string ct, it;
                is >> ct >> it >> p.name;
                p.is_list = true;
                p.count_type = type_of(ct);
                p.type = type_of(it);
                if (p.count_type < 0) { why = "unknown list count type " + ct; return false; }
            } else {
                p.type = type_of(t);
                is >> p.name;
            }
            if (p.type < 0) { why = "unknown property type in: " + line; return false; }
            elements.back().props.push_back(p);
        } else if (word == "end_header") {
            header_done = true;
            break;
        }
    }
    if (!header_done || format < 0) { why = "malformed PLY header"; return false; }
    const bool swap = (format == 1) != host_is_little_endian() && format != 0;
    // what is left of the file bounds every count the header claims (a bogus "element vertex 10^18" must not become a reserve())
    const std::streamoff data_at = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streamoff file_end = f.tellg();
    f.seekg(data_at);
    const size_t bytes_left = file_end > data_at ? (size_t)(file_end - data_at) : 0;
    // a colour or tile value as a byte: clamped (converting a double that is out of range, or a NaN, to uint8_t is undefined)
    const auto to_byte = [](double v) -> uint8_t { return !(v > 0.0) ? (uint8_t)0 : v >= 255.0 ? (uint8_t)255 : (uint8_t)v; };
    for (const PlyElement &e : elements) {
        const bool vertex = e.name == "vertex";
        int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1, it = -1;
        for (size_t i = 0; i < e.props.size(); i++) {
            const std::string &n = e.props[i].name;
            if (e.props[i].is_list) continue;
            if (n == "x") ix = (int)i; else if (n == "y") iy = (int)i; else if (n == "z") iz = (int)i;
            else if (n == "red" || n == "r") ir = (int)i; else if (n == "green" || n == "g") ig = (int)i; else if (n == "blue" || n == "b") ib = (int)i;
            else if (n == "alpha" || n == "tile" || n == "a") it = (int)i;
        }
        if (vertex) {
            if (ix < 0 || iy < 0 || iz < 0) { why = "vertex element without x, y, z"; return false; }
            size_t row_bytes = 0;   // the least a row can take: one byte per value in ASCII (digit + separator), the scalars' sizes in binary
            for (const PlyProperty &pr : e.props) row_bytes += format == 0 ? 2 : (pr.is_list ? kTypes[pr.count_type].size : kTypes[pr.type].size);
            // (ASCII: the last value of the file need not be followed by a newline: one byte of grace)
            if (row_bytes == 0 || e.count > (bytes_left + (format == 0 ? 1 : 0)) / row_bytes) { why = "element vertex claims more rows than the file can hold"; return false; }
            out.reserve(e.count);
        }
        std::vector<double> vals(e.props.size());
        for (size_t row = 0; row < e.count; row++) {
            if (format == 0) {
                if (!std::getline(f, line)) { why = "file ends inside element " + e.name; return false; }
                if (!vertex) continue;
                const char *p = line.c_str();
                for (size_t i = 0; i < e.props.size(); i++) {
                    char *end = nullptr;
                    if (e.props[i].is_list) {
                        const long cnt = strtol(p, &end, 10);
                        if (cnt < 0) { why = "bad list length in: " + line; return false; }
                        p = end;
                        for (long c = 0; c < cnt && *p; c++) { (void)strtod(p, &end); if (end == p) break; p = end; }
                        continue;
                    }
                    vals[i] = strtod(p, &end);
                    if (end == p) { why = "short vertex line: " + line; return false; }
                    p = end;
                }
            } else {
                for (size_t i = 0; i < e.props.size(); i++) {
                    unsigned char buf[8];
                    if (e.props[i].is_list) {
                        if (!f.read((char *)buf, kTypes[e.props[i].count_type].size)) { why = "file ends inside element " + e.name; return false; }
                        const double cnt = load_scalar(buf, e.props[i].count_type, swap);
                        if (!(cnt >= 0.0) || cnt * kTypes[e.props[i].type].size > (double)bytes_left) { why = "bad list length in element " + e.name; return false; }
                        f.seekg((std::streamoff)((size_t)cnt * kTypes[e.props[i].type].size), std::ios::cur);
                        continue;
                    }
                    if (!f.read((char *)buf, kTypes[e.props[i].type].size)) { why = "file ends inside element " + e.name; return false; }
                    if (vertex) vals[i] = load_scalar(buf, e.props[i].type, swap);
                }
            }
            if (vertex) {
                cwipc_point pt;
                pt.x = (float)vals[ix]; pt.y = (float)vals[iy]; pt.z = (float)vals[iz];
                pt.r = ir >= 0 ? to_byte(vals[ir]) : 0;
                pt.g = ig >= 0 ? to_byte(vals[ig]) : 0;
                pt.b = ib >= 0 ? to_byte(vals[ib]) : 0;
                pt.tile = it >= 0 ? to_byte(vals[it]) : 0;
                out.push_back(pt);
            }
        }
        if (vertex) return true;   // nothing after the vertices is of interest
    }
    why = "no vertex element";
    return false;
}

int write_ply(const char *who, const char *filename, cwipc_pointcloud *pc, bool binary, char **errorMessage) {
    if (filename == nullptr || pc == nullptr) { fail(who, "Saving NULL pointcloud not implemented", errorMessage); return -1; }
    const size_t bytes = pc->get_uncompressed_size();
    std::vector<cwipc_point> pts(bytes / sizeof(cwipc_point));
    if (bytes && pc->copy_uncompressed(pts.data(), bytes) < 0) { fail(who, "cannot read the point data of the argument", errorMessage); return -1; }
    FILE *f = fopen(filename, "wb");
    if (!f) { fail(who, std::string("Saving of PLY file failed: ") + filename, errorMessage); return -1; }
    const size_t n = pts.size();
    fprintf(f, "ply\nformat %s 1.0\ncomment PCL generated\nelement vertex %zu\n", binary ? "binary_little_endian" : "ascii", n);
    fputs("property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\n", f);
    fputs("element camera 1\nproperty float view_px\nproperty float view_py\nproperty float view_pz\nproperty float x_axisx\nproperty float x_axisy\n"
          "property float x_axisz\nproperty float y_axisx\nproperty float y_axisy\nproperty float y_axisz\nproperty float z_axisx\nproperty float z_axisy\n"
          "property float z_axisz\nproperty float focal\nproperty float scalex\nproperty float scaley\nproperty float centerx\nproperty float centery\n"
          "property int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n", f);
    bool ok = true;
    if (binary) {
        // x, y, z as little-endian floats and four bytes: a cwipc_point as it stands (on a little-endian host)
        if (host_is_little_endian()) {
            ok = n == 0 || fwrite(pts.data(), sizeof(cwipc_point), n, f) == n;
        } else {
            for (const cwipc_point &p : pts) {
                unsigned char b[16];
                const float c[3] = {p.x, p.y, p.z};
                for (int a = 0; a < 3; a++) { unsigned char t[4]; memcpy(t, &c[a], 4); for (int i = 0; i < 4; i++) b[4 * a + i] = t[3 - i]; }
                b[12] = p.r; b[13] = p.g; b[14] = p.b; b[15] = p.tile;
                ok = ok && fwrite(b, 16, 1, f) == 1;
            }
        }
        const float cam[17] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0};
        const int32_t viewport[2] = {(int32_t)n, 1};
        const float k12[2] = {0, 0};
        ok = ok && fwrite(cam, sizeof(cam), 1, f) == 1 && fwrite(viewport, sizeof(viewport), 1, f) == 1 && fwrite(k12, sizeof(k12), 1, f) == 1;
    } else {
        for (const cwipc_point &p : pts) {
            if (fprintf(f, "%.8g %.8g %.8g %d %d %d %d\n", (double)p.x, (double)p.y, (double)p.z, (int)p.r, (int)p.g, (int)p.b, (int)p.tile) < 0) { ok = false; break; }
        }
        ok = ok && fprintf(f, "0 0 0 1 0 0 0 1 0 0 0 1 0 0 0 0 0 %zu 1 0 0\n", n) > 0;
    }
    ok = fclose(f) == 0 && ok;
    if (!ok) { fail(who, std::string("Saving of PLY file failed: ") + filename, errorMessage); return -1; }
    return 0;
}

}  // namespace

// reference src/cwipc_util.cpp:432-461
extern "C" cwipc_pointcloud *cwipc_read(const char *filename, uint64_t timestamp, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_read", apiVersion, errorMessage)) return nullptr;
    std::vector<cwipc_point> pts;
    std::string why;
    bool read_ok = false;
    try {   // (no exception crosses the C boundary: an allocation failure inside the reader is a failed load like any other)
        read_ok = filename != nullptr && read_ply(filename, pts, why);
    } catch (const std::exception &e) {
        why = e.what();
    }
    if (!read_ok) {
        fail("cwipc_read", std::string("Loading of PLY file failed: ") + (filename ? filename : "(null)") + (why.empty() ? "" : " (" + why + ")"), errorMessage);
        return nullptr;
    }
    cwipc_log_set_errorbuf(errorMessage);
    auto *rv = new cwipc_hip_pointcloud();
    if (rv->from_points(pts.data(), pts.size() * sizeof(cwipc_point), (int)pts.size(), timestamp, /* exact_size */ false) < 0) {
        delete rv;
        rv = nullptr;
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_read", "unspecified error creating point cloud");
    }
    cwipc_log_set_errorbuf(nullptr);
    return rv;
}

// reference src/cwipc_util.cpp:463-479
extern "C" int cwipc_write(const char *filename, cwipc_pointcloud *pointcloud, char **errorMessage) {
    return write_ply("cwipc_write", filename, pointcloud, false, errorMessage);
}

// reference src/cwipc_util.cpp:481-497 (flag & CWIPC_FLAG_BINARY)
extern "C" int cwipc_write_ext(const char *filename, cwipc_pointcloud *pointcloud, int flag, char **errorMessage) {
    return write_ply("cwipc_write_ext", filename, pointcloud, (flag & 1) != 0, errorMessage);
}
