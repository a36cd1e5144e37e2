// .txt scene loader of the hw1-hw5 snapshots -> flat host scene (analytic primitives).
//
// Replaces sceneio::loadScene(std::istream&) (hw3/src/sceneio.cpp:8-107, hw1/src/sceneio.cpp:8-97).
// The format is line oriented: the first word of a line is a command, the rest are numbers read with
// `stream >> float`.  Two behaviours of that extraction are part of the format:
//   * a field that does not parse sets failbit: the field becomes 0 and every later field of the line keeps
//     its default (hw3/practice3_5.txt:46 "ROTATION 0 0.3826834, 0 0.9238795" -> (0, 0.3826834, 0, 1));
//   * inside a NEW_PRIMITIVE block the first unrecognised line ends the block and is re-dispatched as a
//     header command.
// hw3+ normalises the plane normal at load (hw3/src/sceneio.cpp:25), hw1 does not (hw1/src/sceneio.cpp:24).
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
    if (flavor != RT_INTEGRATOR_HW1 && flavor != RT_INTEGRATOR_HW3) throw std::runtime_error("rt_load_txt: flavor must be RT_INTEGRATOR_HW1 or RT_INTEGRATOR_HW3");
    std::vector<uint8_t> bytes = read_file(path);
    std::istringstream in(std::string(bytes.begin(), bytes.end()));
    std::unique_ptr<rt_host_scene> hs(new rt_host_scene());
    int32_t width = 0, height = 0, spp = 1, ray_depth = 1;
    std::string line;
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
        else if (cmd == "RAY_DEPTH" && flavor != RT_INTEGRATOR_HW1) lr.read(ray_depth);
        else if (cmd == "SAMPLES" && flavor != RT_INTEGRATOR_HW1) lr.read(spp);
        else if (cmd == "NEW_PRIMITIVE") {
            // loadPrimitive: the NEXT line must name the shape (hw3/src/sceneio.cpp:11-32)
            rt_primitive p;
            memset(&p, 0, sizeof p);
            p.rotation[3] = 1;          // Quaternion() / Figure::rotation default
            p.kind = RT_MAT_DIFFUSE;
            p.ior = 1.0f;               // hw3 leaves Figure::ior uninitialised without IOR; defined here as 1
            std::string shape_line;
            if (!get_line(in, shape_line)) { shape_line.clear(); }
            LineReader sr(shape_line);
            std::string shape = sr.word();
            bool known = true;
            if (shape == "ELLIPSOID") { p.type = RT_PRIM_ELLIPSOID; sr.read3(p.data); }
            else if (shape == "PLANE") {
                p.type = RT_PRIM_PLANE; sr.read3(p.data);
                if (flavor != RT_INTEGRATOR_HW1) { // n.normalize(): 1./len() in double, narrowed (vec3.h:78-80)
                    float len = (float)std::sqrt((double)(p.data[0] * p.data[0] + p.data[1] * p.data[1] + p.data[2] * p.data[2]));
                    float k = (float)(1. / (double)len);
                    for (int i = 0; i < 3; i++) p.data[i] = k * p.data[i];
                }
            }
            else if (shape == "BOX") { p.type = RT_PRIM_BOX; sr.read3(p.data); }
            else {
                known = false;
                fprintf(stderr, "UNKNWOWN FIGURE: %s@%s\n", shape.c_str(), shape_line.c_str());
            }
            if (!known) {
                // hw3 pushes a default figure (type uninitialised) and hw1 pushes nothing; both re-dispatch the line.
                line = shape_line;
                continue;
            }
            bool ended = true;
            while (get_line(in, line)) {
                LineReader pr(line);
                std::string c = pr.word();
                if (c == "COLOR") pr.read3(p.color);
                else if (c == "POSITION") pr.read3(p.position);
                else if (c == "ROTATION") { pr.read(p.rotation[0]); pr.read(p.rotation[1]); pr.read(p.rotation[2]); pr.read(p.rotation[3]); }
                else if (c == "METALLIC" && flavor != RT_INTEGRATOR_HW1) p.kind = RT_MAT_METALLIC;
                else if (c == "DIELECTRIC" && flavor != RT_INTEGRATOR_HW1) p.kind = RT_MAT_DIELECTRIC;
                else if (c == "EMISSION" && flavor != RT_INTEGRATOR_HW1) pr.read3(p.emission);
                else if (c == "IOR" && flavor != RT_INTEGRATOR_HW1) pr.read(p.ior);
                else { ended = false; break; }
            }
            hs->primitives.push_back(p);
            if (!ended) { redispatch = true; }
            else { have = false; break; }
        } else if (!cmd.empty()) fprintf(stderr, "UNKNOWN COMMAND: %s\n", cmd.c_str());
        if (redispatch) continue; // `line` already holds the line that ended the primitive block
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
