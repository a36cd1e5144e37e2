// .txt scene loader of the hw1-hw5 snapshots -> flat host scene (analytic primitives).
//
// Replaces sceneio::loadScene(std::istream&) of hw1 (hw1/src/sceneio.cpp:8-97), hw2 (hw2/src/sceneio.cpp:8-147),
// hw3/hw4 (hw3/src/sceneio.cpp:8-107) and hw5 (hw5/src/sceneio.cpp:8-101); `flavor` picks the grammar.
// The format is line oriented: the first word of a line is a command, the rest are numbers read with
// `stream >> float`.  Two behaviours of that extraction are part of the format:
//   * a field that does not parse sets failbit: the field becomes 0 and every later field of the line keeps
//     its default (hw3/practice3_5.txt:46 "ROTATION 0 0.3826834, 0 0.9238795" -> (0, 0.3826834, 0, 1));
//   * inside a NEW_PRIMITIVE block the first unrecognised line ends the block and is re-dispatched as a
//     header command.
#include "host_scene.h"
#include "png.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace rtamd {
namespace {

// One line being consumed the way `std::stringstream >> value` would.
struct LineReader {
    const std::string &s;
    size_t pos = 0;
    bool failed = false;
    explicit LineReader(const std::string &line) : s(line) {}
    void skip_ws() { while (pos < s.size() && isspace((unsigned char)s[pos])) pos++; }
    std::string word() {
        skip_ws();
        size_t b = pos;
        while (pos < s.size() && !isspace((unsigned char)s[pos])) pos++;
        return s.substr(b, pos - b);
    }
    // num_get semantics: accumulate the characters that can belong to a number, convert, demand that the
    // whole accumulation converted; on failure the target becomes 0 and the stream stays failed.
    void read(float &v) {
        if (failed) return;
        skip_ws();
        size_t b = pos;
        if (pos < s.size() && (s[pos] == '+' || s[pos] == '-')) pos++;
        bool digits = false;
        while (pos < s.size() && isdigit((unsigned char)s[pos])) { pos++; digits = true; }
        if (pos < s.size() && s[pos] == '.') { pos++; while (pos < s.size() && isdigit((unsigned char)s[pos])) { pos++; digits = true; } }
        bool bad_exp = false;
        if (digits && pos < s.size() && (s[pos] == 'e' || s[pos] == 'E')) {
            pos++;
            if (pos < s.size() && (s[pos] == '+' || s[pos] == '-')) pos++;
            bool ed = false;
            while (pos < s.size() && isdigit((unsigned char)s[pos])) { pos++; ed = true; }
            bad_exp = !ed;
        }
        if (!digits || bad_exp) { v = 0; failed = true; return; }
        v = strtof(s.substr(b, pos - b).c_str(), nullptr);
    }
    void read(int32_t &v) {
        if (failed) return;
        skip_ws();
        size_t b = pos;
        if (pos < s.size() && (s[pos] == '+' || s[pos] == '-')) pos++;
        bool digits = false;
        while (pos < s.size() && isdigit((unsigned char)s[pos])) { pos++; digits = true; }
        if (!digits) { v = 0; failed = true; return; }
        v = (int32_t)strtol(s.substr(b, pos - b).c_str(), nullptr, 10);
    }
    void read3(float *v) { read(v[0]); read(v[1]); read(v[2]); }
};

bool get_line(std::istringstream &in, std::string &line) { return (bool)std::getline(in, line); }

} // namespace

rt_host_scene *load_txt(const std::string &path, int flavor, int32_t *w, int32_t *h, int32_t *samples, int32_t *depth) {
    if (flavor < RT_INTEGRATOR_HW1 || flavor > RT_INTEGRATOR_HW5) throw std::runtime_error("rt_load_txt: flavor must be RT_INTEGRATOR_HW1..HW5");
    // What each snapshot's sceneio.cpp understands.
    const bool has_depth = flavor >= RT_INTEGRATOR_HW2;       // RAY_DEPTH, METALLIC, DIELECTRIC, IOR
    const bool has_samples = flavor >= RT_INTEGRATOR_HW3;     // SAMPLES, EMISSION
    const bool has_lights = flavor == RT_INTEGRATOR_HW2;      // NEW_LIGHT, LIGHT_*, AMBIENT_LIGHT
    const bool shape_in_block = flavor == RT_INTEGRATOR_HW5;  // hw5: the shape is one more property line, and TRIANGLE exists
    // hw2 normalises in Plane::Plane (hw2/src/primitives.cpp:78), hw3/hw4 at load (hw3/src/sceneio.cpp:25); hw1 and hw5 keep it raw.
    const bool normalise_plane = flavor >= RT_INTEGRATOR_HW2 && flavor <= RT_INTEGRATOR_HW4;
    std::vector<uint8_t> bytes = read_file(path);
    std::istringstream in(std::string(bytes.begin(), bytes.end()));
    std::unique_ptr<rt_host_scene> hs(new rt_host_scene());
    int32_t width = 0, height = 0, spp = 1, ray_depth = 1;
    std::string line;
    auto read_shape = [&](LineReader &sr, const std::string &shape, rt_primitive &p) -> bool {
        if (shape == "ELLIPSOID") { p.type = RT_PRIM_ELLIPSOID; sr.read3(p.data); }
        else if (shape == "PLANE") {
            p.type = RT_PRIM_PLANE; sr.read3(p.data);
            if (normalise_plane) { // n.normalize(): 1./len() in double, narrowed (vec3.h:78-80)
                float len = (float)std::sqrt((double)(p.data[0] * p.data[0] + p.data[1] * p.data[1] + p.data[2] * p.data[2]));
                float k = (float)(1. / (double)len);
                for (int i = 0; i < 3; i++) p.data[i] = k * p.data[i];
            }
        }
        else if (shape == "BOX") { p.type = RT_PRIM_BOX; sr.read3(p.data); }
        else if (shape == "TRIANGLE" && shape_in_block) { p.type = RT_PRIM_TRIANGLE; sr.read3(p.data3); sr.read3(p.data2); sr.read3(p.data); } // hw5/src/sceneio.cpp:27-29
        else return false;
        return true;
    };
    bool have = get_line(in, line);
    while (have) {
        LineReader lr(line);
        std::string cmd = lr.word();
        bool redispatch = false;
        if (cmd == "DIMENSIONS") { lr.read(width); lr.read(height); }
        else if (cmd == "BG_COLOR") lr.read3(hs->bg);
        else if (cmd == "CAMERA_POSITION") lr.read3(hs->camera.position);
        else if (cmd == "CAMERA_RIGHT") lr.read3(hs->camera.right);
        else if (cmd == "CAMERA_UP") lr.read3(hs->camera.up);
        else if (cmd == "CAMERA_FORWARD") lr.read3(hs->camera.forward);
        else if (cmd == "CAMERA_FOV_X") lr.read(hs->camera.fov_x);
        else if (cmd == "RAY_DEPTH" && has_depth) lr.read(ray_depth);
        else if (cmd == "SAMPLES" && has_samples) lr.read(spp);
        else if (cmd == "AMBIENT_LIGHT" && has_lights) lr.read3(hs->ambient);
        else if (cmd == "NEW_LIGHT" && has_lights) {
            // loadLightSource (hw2/src/sceneio.cpp:58-92): any LIGHT_DIRECTION line makes it directional.
            rt_light L;
            memset(&L, 0, sizeof L); // the reference leaves unset fields uninitialised; defined here as 0
            L.type = RT_LIGHT_POINT;
            bool ended = true;
            while (get_line(in, line)) {
                LineReader pr(line);
                std::string c = pr.word();
                if (c == "LIGHT_INTENSITY") pr.read3(L.intensity);
                else if (c == "LIGHT_POSITION") pr.read3(L.position);
                else if (c == "LIGHT_DIRECTION") { pr.read3(L.direction); L.type = RT_LIGHT_DIRECTIONAL; }
                else if (c == "LIGHT_ATTENUATION") pr.read3(L.attenuation);
                else { ended = false; break; }
            }
            hs->lights.push_back(L);
            if (!ended) redispatch = true;
            else { have = false; break; }
        }
        else if (cmd == "NEW_PRIMITIVE") {
            rt_primitive p;
            memset(&p, 0, sizeof p);
            p.rotation[3] = 1;          // Quaternion() / Figure::rotation default
            p.kind = RT_MAT_DIFFUSE;
            p.ior = 1.0f;               // the reference leaves Figure::ior uninitialised without IOR; defined here as 1
            p.type = -1;
            if (!shape_in_block) {
                // loadPrimitive: the NEXT line must name the shape (hw3/src/sceneio.cpp:11-32)
                std::string shape_line;
                if (!get_line(in, shape_line)) { shape_line.clear(); }
                LineReader sr(shape_line);
                std::string shape = sr.word();
                if (!read_shape(sr, shape, p)) {
                    fprintf(stderr, "UNKNWOWN FIGURE: %s@%s\n", shape.c_str(), shape_line.c_str());
                    // hw3 pushes a default figure (type uninitialised), hw2 a null pointer, hw1 nothing; all re-dispatch the line.
                    line = shape_line;
                    continue;
                }
            }
            bool ended = true;
            while (get_line(in, line)) {
                LineReader pr(line);
                std::string c = pr.word();
                if (shape_in_block && read_shape(pr, c, p)) continue; // hw5/src/sceneio.cpp:17-29
                if (c == "COLOR") pr.read3(p.color);
                else if (c == "POSITION") pr.read3(p.position);
                else if (c == "ROTATION") { pr.read(p.rotation[0]); pr.read(p.rotation[1]); pr.read(p.rotation[2]); pr.read(p.rotation[3]); }
                else if (c == "METALLIC" && has_depth) p.kind = RT_MAT_METALLIC;
                else if (c == "DIELECTRIC" && has_depth) p.kind = RT_MAT_DIELECTRIC;
                else if (c == "EMISSION" && has_samples) pr.read3(p.emission);
                else if (c == "IOR" && has_depth) pr.read(p.ior);
                else { ended = false; break; }
            }
            if (p.type < 0) throw std::runtime_error("rt_load_txt: NEW_PRIMITIVE block without a shape line (the reference's Figure::type would be uninitialised)");
            hs->primitives.push_back(p);
            if (!ended) { redispatch = true; }
            else { have = false; break; }
        } else if (!cmd.empty()) fprintf(stderr, "UNKNOWN COMMAND: %s\n", cmd.c_str());
        if (redispatch) continue; // `line` already holds the line that ended the block
        have = get_line(in, line);
    }
    hs->camera.fov_y = 0;
    hs->finalize();
    if (w) *w = width;
    if (h) *h = height;
    if (samples) *samples = spp;
    if (depth) *depth = ray_depth;
    return hs.release();
}

} // namespace rtamd
