// synthetic.cpp -- the synthetic point cloud source, the input of every benchmark
// configuration.  Restates reference src/cwipc_synthetic.cpp.  With a GPU the points are
// generated on the device, straight into the planes the filters read (k::synthetic_fill: the
// per-row and per-column libm values still come from the host, made as the reference makes
// them); without one -- or with CWIPC_SYNTHETIC_HOST=1 -- on the host, which is also the
// pinned checker of the device generator (tests compare the two byte for byte).
//
// Two additions for reproducibility, both behind hooks the reference already has:
//   - auxiliary_operation("test-setangle")      reference :169-179 (sets m_angle; the
//     next get() overwrites it with wall-clock time there, and here too), and
//   - auxiliary_operation("amd-fixangle"): pins m_angle so get() stops following
//     the wall clock.  The reference's colours depend on elapsed time (:120,126,
//     198-200), which no parity test could pin.
#include "internal.hpp"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

using namespace cwipc_amd;

class synthetic_source : public cwipc_activesource {
    float m_angle = 0;
    bool m_angle_fixed = false;
    std::chrono::system_clock::time_point m_start;
    std::chrono::system_clock::time_point m_earliest_next;
    int m_hsteps = 0, m_asteps = 0;
    int m_fps = 0;
    cwipc_point *m_points = nullptr;
    size_t m_points_size = 0;
    bool m_started = false;
    // device generator: per-row / per-column tables (radius, height | angle, sin, cos), one pool block, made once
    void *m_tables = nullptr;
    int m_tables_device = -1;
    float m_first[3] = {0, 0, 0};   // the cloud's first point (coordinates do not depend on the angle)

public:
    // reference :32-49 -- npoints 0 means 160000; the cloud is int(sqrt(n))^2 points.
    synthetic_source(int fps, int npoints) : m_fps(fps) {
        if (npoints == 0) npoints = 160000;
        m_hsteps = m_asteps = int(sqrt(npoints));
        m_points_size = (size_t)m_hsteps * m_asteps * sizeof(cwipc_point);
        m_points = (cwipc_point *)malloc(m_points_size ? m_points_size : 1);
    }
    ~synthetic_source() override { free(); }

    void free() override {
        ::free(m_points);
        m_points = nullptr;
        if (m_tables) pool_free(m_tables);
        m_tables = nullptr;
    }
    bool reload_config(const char *) override {
        cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_synthetic", "reload_config() not implemented (nor needed)");
        return false;
    }
    size_t get_config(char *, size_t) override { return 0; }
    bool start() override {
        if (m_started) {
            cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_synthetic", "start() called when already started");
            return true;
        }
        m_start = std::chrono::system_clock::now();
        m_earliest_next = m_start;
        m_started = true;
        return true;
    }
    void stop() override { m_started = false; }
    bool eof() override { return false; }
    bool seek(uint64_t) override { return false; }

    // reference :95-108
    bool available(bool wait) override {
        if (!m_started) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_synthetic", "available() called before start()");
            return false;
        }
        if (!wait && m_fps != 0 && m_earliest_next.time_since_epoch() != std::chrono::milliseconds(0) &&
            std::chrono::system_clock::now() < m_earliest_next) {
            return false;
        }
        return true;
    }

    // reference :110-143
    cwipc_pointcloud *get() override {
        if (!m_started) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_synthetic", "get() called before start()");
            return nullptr;
        }
        if (m_fps != 0 && m_earliest_next.time_since_epoch() != std::chrono::milliseconds(0)) {
            std::this_thread::sleep_until(m_earliest_next);
        }
        auto now = std::chrono::system_clock::now();
        uint64_t timestamp = std::chrono::duration_cast<std::chrono::milliseconds>(now.time_since_epoch()).count();
        std::chrono::duration<float, std::ratio<1>> runtime = now - m_start;
        if (m_fps != 0) m_earliest_next = now + std::chrono::milliseconds(1000 / m_fps);
        if (!m_angle_fixed) m_angle = runtime.count();
        cwipc_pointcloud *rv = generate_on_device(timestamp);
        if (rv == nullptr) {
            generate_points();
            rv = cwipc_from_points(m_points, m_points_size, m_hsteps * m_asteps, timestamp, nullptr, CWIPC_API_VERSION);
        }
        if (rv) {
            rv->_set_cellsize(2.0 / m_hsteps);
            if (is_metadata_requested("test-angle")) {
                void *mem = malloc(sizeof(m_angle));
                memcpy(mem, &m_angle, sizeof(m_angle));
                rv->access_metadata()->_add("test-angle", "", mem, sizeof(m_angle), ::free);
            }
        }
        return rv;
    }

    int maxtile() override { return 3; }

    // reference :149-167
    bool get_tileinfo(int tilenum, struct cwipc_tileinfo *tileinfo) override {
        static cwipc_tileinfo info[3] = {
            {{0, 0, 0}, (char *)"synthetic", 2, 0},
            {{0, 0, 1}, (char *)"synthetic-right", 1, 1},
            {{0, 0, -1}, (char *)"synthetic-left", 1, 2},
        };
        if (tilenum < 0 || tilenum > 2) return false;
        if (tileinfo) *tileinfo = info[tilenum];
        return true;
    }

    bool auxiliary_operation(const std::string op, const void *inbuf, size_t insize, void *outbuf, size_t outsize) override {
        bool fix = op == "amd-fixangle";
        if (op != "test-setangle" && !fix) return false;
        if (inbuf == nullptr || insize != sizeof(float)) return false;
        if (outbuf == nullptr || outsize != sizeof(float)) return false;
        memcpy(&m_angle, inbuf, sizeof(float));
        memcpy(outbuf, &m_angle, sizeof(float));
        if (fix) m_angle_fixed = true;
        return true;
    }

private:
    // The same cloud, generated on the device (nullptr: no GPU, or switched off -- the caller takes the host path).
    cwipc_pointcloud *generate_on_device(uint64_t timestamp) {
        static const bool host_only = []() { const char *e = getenv("CWIPC_SYNTHETIC_HOST"); return e && atoi(e) != 0; }();
        const size_t n = (size_t)m_hsteps * m_asteps;
        if (host_only || n == 0 || cwipc_hip_device_count() < 1 || current_device() >= cwipc_hip_device_count()) return nullptr;
        ThreadCtx &c = tctx();
        if (!c.ensure()) return nullptr;
        const size_t nh = (size_t)m_hsteps, na = (size_t)m_asteps;
        // layout: double sin[na] cos[na] | float radius[nh] height[nh] angle[na]
        const size_t bytes = 2 * na * sizeof(double) + (2 * nh + na) * sizeof(float);
        if (m_tables == nullptr || m_tables_device != current_device()) {
            if (m_tables) pool_free(m_tables);
            m_tables = pool_alloc(bytes);
            if (!m_tables) return nullptr;
            char *h = (char *)c.staging(bytes);
            if (!h) return nullptr;
            double *sin_a = (double *)h, *cos_a = sin_a + na;
            float *radius = (float *)(cos_a + na), *height = radius + nh, *angle = height + nh;
            // reference :183-196, the values that do not depend on the other loop index
            const float pi = 3.14159265358979f;
            const float max_height = 2.0;
            const float delta_h = max_height / m_hsteps;
            const float delta_a = 2 * pi / m_asteps;
            for (int hi = 0; hi < m_hsteps; hi++) {
                height[hi] = hi * delta_h;
                radius[hi] = 0.3 * pow(cos((double)(height[hi] * pi / 3 - pi / 6)), 0.71);
            }
            for (int ai = 0; ai < m_asteps; ai++) {
                angle[ai] = ai * delta_a;
                sin_a[ai] = sin((double)angle[ai]);
                cos_a[ai] = cos((double)angle[ai]);
            }
            const float x0 = radius[0] * sin_a[0], y0 = radius[0] * cos_a[0];
            m_first[0] = -x0; m_first[1] = height[0]; m_first[2] = y0;
            bool ok = hipMemcpyAsync(m_tables, h, bytes, hipMemcpyHostToDevice, c.stream) == hipSuccess;
            ok = c.sync() && ok;
            if (!ok) { pool_free(m_tables); m_tables = nullptr; return nullptr; }
            m_tables_device = current_device();
        }
        auto dst = soa_alloc(n);
        if (!dst) return nullptr;
        const double *sin_a = (const double *)m_tables, *cos_a = sin_a + na;
        const float *radius = (const float *)(cos_a + na), *height = radius + nh, *angle = height + nh;
        const float pi = 3.14159265358979f;
        const bool eyes_white = fmod((double)m_angle, (double)(pi / 2)) > 0.08;   // reference :209
        k::synthetic_fill(*dst, m_hsteps, m_asteps, m_angle, eyes_white, radius, height, angle, sin_a, cos_a, c.stream);
        const uint32_t both[8] = {6u, 0, 0, 0, 0, 0, 0, 0};   // tile = (y < 0) ? 1 : 2 (reference src/cwipc_synthetic.cpp:218)
        dst->set_tiles(both);
        if (!c.sync()) return nullptr;
        dst->first[0] = m_first[0]; dst->first[1] = m_first[1]; dst->first[2] = m_first[2];
        dst->has_first = true;
        auto *rv = new cwipc_hip_pointcloud();
        rv->adopt_device(dst, timestamp, 0.0f, /* exact_size */ true);   // (cwipc_from_points clouds want their exact size in copy_uncompressed)
        return rv;
    }

    // reference :182-222.  Mixed float/double arithmetic exactly as written there
    // (float locals, double libm calls and literals).
    void generate_points() {
        const float pi = 3.14159265358979f;
        const float max_height = 2.0;
        const float delta_h = max_height / m_hsteps;
        const float delta_a = 2 * pi / m_asteps;
        cwipc_point *p = m_points;
        for (int hi = 0; hi < m_hsteps; hi++) {
            float height = hi * delta_h;
            for (int ai = 0; ai < m_asteps; ai++) {
                float angle = ai * delta_a;
                float radius = 0.3 * pow(cos((double)(height * pi / 3 - pi / 6)), 0.71);
                float x = radius * sin((double)angle);
                float y = radius * cos((double)angle);
                float r = (1 + sin((double)(2 * pi * height + m_angle + angle))) / 2;
                float g = (1 + sin((double)(3 * pi * height + m_angle + angle))) / 2;
                float b = (1 + sin((double)(4 * pi * height + m_angle + angle))) / 2;
                int rr = (int)(r * 255.0), gg = (int)(g * 255.0), bb = (int)(b * 255.0);
                if (height > 1.7 && height < 1.8 &&
                    ((angle > pi * 0.083 && angle < pi * 0.1667) || (angle > pi * 1.833 && angle < pi * 1.917))) {
                    if (fmod((double)m_angle, (double)(pi / 2)) > 0.08) rr = gg = bb = 255;
                }
                p->x = -x;
                p->y = height;
                p->z = y;
                p->r = (uint8_t)rr;
                p->g = (uint8_t)gg;
                p->b = (uint8_t)bb;
                p->tile = y < 0 ? 1 : 2;
                p++;
            }
        }
    }
};

}  // namespace

// reference src/cwipc_synthetic.cpp:225-242
extern "C" cwipc_activesource *cwipc_synthetic(int fps, int npoints, char **errorMessage, uint64_t apiVersion) {
    if (cwipc_amd::api_version_rejected("cwipc_synthetic", apiVersion, errorMessage)) return nullptr;
    cwipc_log_set_errorbuf(errorMessage);
    cwipc_activesource *rv = new synthetic_source(fps, npoints);
    cwipc_log_set_errorbuf(nullptr);
    return rv;
}
