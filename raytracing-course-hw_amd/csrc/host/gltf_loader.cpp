// glTF 2.0 subset loader -> flat host scene.
//
// Replaces sceneio::loadScene of the reference (hw8/src/sceneio.cpp:348-372; hw6 flavour
// hw6/src/sceneio.cpp) float-for-float: same key set, same defaults, same float32 arithmetic
// for node matrices (hw8/src/include/transition.h:11-138), the same corner order
// Figure(v1, v3, v2) (hw8/src/sceneio.cpp:289) and the same quirks (index accessor byteOffset
// ignored, :258-269; last camera node wins, :317-325; every node instantiated, `scenes` ignored).
// Unlike the reference it validates what it reads and reports errors instead of crashing.
// Compile with -ffp-contract=off: the matrix arithmetic must round exactly like the x86-64
// -O3 build of the reference (no FMA).
#include "host_scene.h"
#include "json.h"
#include "png.h"
#include <cmath>
#include <cstring>
#include <optional>

namespace rtamd {
namespace {

struct F3 { float x = 0, y = 0, z = 0; };
inline F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 normalized(F3 a) { // hw8/src/include/vec3.h:65-67,78-80
    float len = (float)std::sqrt((double)(a.x * a.x + a.y * a.y + a.z * a.z));
    float k = (float)(1. / (double)len);
    return {k * a.x, k * a.y, k * a.z};
}

// 4x4 affine in row-major [row][col] float32 (transition.h).
struct Mat4 {
    float m[4][4];
    static Mat4 identity() {
        Mat4 r;
        memset(r.m, 0, sizeof r.m);
        r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1;
        return r;
    }
    // transition.h:55-66 — accumulate from +0 in k order.
    Mat4 compose(const Mat4 &o) const {
        Mat4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                float acc = 0;
                for (int k = 0; k < 4; k++) acc += m[i][k] * o.m[k][j];
                r.m[i][j] = acc;
            }
        return r;
    }
    // transition.h:68-74
    F3 apply(F3 p) const {
        float r[3];
        for (int i = 0; i < 3; i++) r[i] = m[i][0] * p.x + m[i][1] * p.y + m[i][2] * p.z + m[i][3];
        return {r[0], r[1], r[2]};
    }
    Mat4 transposed() const {
        Mat4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) r.m[j][i] = m[i][j];
        return r;
    }
    // transition.h:81-128: the classic GLU cofactor inverse.  Each cofactor is a signed sum of six
    // triple products taken left to right; the table lists them in the published order because the
    // float result depends on it.  Entry = {sign, i, j, k} meaning sign * a[i]*a[j]*a[k].
    Mat4 inverted() const {
        const float *a = &m[0][0];
        static const signed char T[16][6][4] = {
            /* 0*/ {{+1, 5, 10, 15}, {-1, 5, 11, 14}, {-1, 9, 6, 15}, {+1, 9, 7, 14}, {+1, 13, 6, 11}, {-1, 13, 7, 10}},
            /* 1*/ {{-1, 1, 10, 15}, {+1, 1, 11, 14}, {+1, 9, 2, 15}, {-1, 9, 3, 14}, {-1, 13, 2, 11}, {+1, 13, 3, 10}},
            /* 2*/ {{+1, 1, 6, 15}, {-1, 1, 7, 14}, {-1, 5, 2, 15}, {+1, 5, 3, 14}, {+1, 13, 2, 7}, {-1, 13, 3, 6}},
            /* 3*/ {{-1, 1, 6, 11}, {+1, 1, 7, 10}, {+1, 5, 2, 11}, {-1, 5, 3, 10}, {-1, 9, 2, 7}, {+1, 9, 3, 6}},
            /* 4*/ {{-1, 4, 10, 15}, {+1, 4, 11, 14}, {+1, 8, 6, 15}, {-1, 8, 7, 14}, {-1, 12, 6, 11}, {+1, 12, 7, 10}},
            /* 5*/ {{+1, 0, 10, 15}, {-1, 0, 11, 14}, {-1, 8, 2, 15}, {+1, 8, 3, 14}, {+1, 12, 2, 11}, {-1, 12, 3, 10}},
            /* 6*/ {{-1, 0, 6, 15}, {+1, 0, 7, 14}, {+1, 4, 2, 15}, {-1, 4, 3, 14}, {-1, 12, 2, 7}, {+1, 12, 3, 6}},
            /* 7*/ {{+1, 0, 6, 11}, {-1, 0, 7, 10}, {-1, 4, 2, 11}, {+1, 4, 3, 10}, {+1, 8, 2, 7}, {-1, 8, 3, 6}},
            /* 8*/ {{+1, 4, 9, 15}, {-1, 4, 11, 13}, {-1, 8, 5, 15}, {+1, 8, 7, 13}, {+1, 12, 5, 11}, {-1, 12, 7, 9}},
            /* 9*/ {{-1, 0, 9, 15}, {+1, 0, 11, 13}, {+1, 8, 1, 15}, {-1, 8, 3, 13}, {-1, 12, 1, 11}, {+1, 12, 3, 9}},
            /*10*/ {{+1, 0, 5, 15}, {-1, 0, 7, 13}, {-1, 4, 1, 15}, {+1, 4, 3, 13}, {+1, 12, 1, 7}, {-1, 12, 3, 5}},
            /*11*/ {{-1, 0, 5, 11}, {+1, 0, 7, 9}, {+1, 4, 1, 11}, {-1, 4, 3, 9}, {-1, 8, 1, 7}, {+1, 8, 3, 5}},
            /*12*/ {{-1, 4, 9, 14}, {+1, 4, 10, 13}, {+1, 8, 5, 14}, {-1, 8, 6, 13}, {-1, 12, 5, 10}, {+1, 12, 6, 9}},
            /*13*/ {{+1, 0, 9, 14}, {-1, 0, 10, 13}, {-1, 8, 1, 14}, {+1, 8, 2, 13}, {+1, 12, 1, 10}, {-1, 12, 2, 9}},
            /*14*/ {{-1, 0, 5, 14}, {+1, 0, 6, 13}, {+1, 4, 1, 14}, {-1, 4, 2, 13}, {-1, 12, 1, 6}, {+1, 12, 2, 5}},
            /*15*/ {{+1, 0, 5, 10}, {-1, 0, 6, 9}, {-1, 4, 1, 10}, {+1, 4, 2, 9}, {+1, 8, 1, 6}, {-1, 8, 2, 5}},
        };
        float inv[16];
        for (int e = 0; e < 16; e++) {
            // first term carries its sign on the first factor ("-m[4]*m[10]*m[15]" = ((-m4)*m10)*m15)
            const signed char *t0 = T[e][0];
            float acc = ((t0[0] < 0 ? -a[t0[1]] : a[t0[1]]) * a[t0[2]]) * a[t0[3]];
            for (int q = 1; q < 6; q++) {
                const signed char *t = T[e][q];
                float prod = (a[t[1]] * a[t[2]]) * a[t[3]];
                acc = t[0] < 0 ? acc - prod : acc + prod;
            }
            inv[e] = acc;
        }
        float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
        det = (float)(1.0 / (double)det);
        Mat4 r;
        for (int i = 0; i < 16; i++) (&r.m[0][0])[i] = inv[i] * det;
        return r;
    }
};

// transition.h:11-53 — T * R * S with the reference's exact element formulas.
Mat4 from_trs(F3 t, const float q[4] /*x,y,z,w*/, F3 s) {
    Mat4 T = Mat4::identity(), R = Mat4::identity(), S = Mat4::identity();
    T.m[0][3] = t.x; T.m[1][3] = t.y; T.m[2][3] = t.z;
    float x = q[0], y = q[1], z = q[2], w = q[3];
    R.m[0][0] = 2 * (w * w + x * x) - 1; R.m[0][1] = 2 * (x * y - w * z);     R.m[0][2] = 2 * (x * z + w * y);
    R.m[1][0] = 2 * (x * y + w * z);     R.m[1][1] = 2 * (w * w + y * y) - 1; R.m[1][2] = 2 * (y * z - w * x);
    R.m[2][0] = 2 * (x * z - w * y);     R.m[2][1] = 2 * (y * z + w * x);     R.m[2][2] = 2 * (w * w + z * z) - 1;
    S.m[0][0] = s.x; S.m[1][1] = s.y; S.m[2][2] = s.z;
    return T.compose(R).compose(S);
}

struct Node {
    std::optional<size_t> mesh, camera, parent;
    float rotation[4] = {0, 0, 0, 1};
    F3 scale{1, 1, 1}, translation{0, 0, 0};
    std::optional<Mat4> local;
    std::vector<size_t> children;
    Mat4 total;
};
struct BufferView { size_t buffer, byteLength, byteOffset, byteStride; };
struct Accessor { size_t bufferView, count, componentType; std::string type; size_t byteOffset; };
struct Prim { size_t position; std::optional<size_t> texcoord, normal, tangent; size_t indices, material; };

std::string dir_of(const std::string &path) {
    size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

struct Loader {
    std::string path;
    int flavor;
    Json doc;
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<BufferView> views;
    std::vector<Node> nodes;
    std::vector<std::vector<Prim>> meshes;
    std::vector<Accessor> accessors;
    rt_host_scene *hs = nullptr;

    const uint8_t *accessor_ptr(size_t index, size_t elem_bytes, size_t &count, bool use_accessor_offset) {
        if (index >= accessors.size()) throw std::runtime_error("glTF: accessor index out of range");
        const Accessor &a = accessors[index];
        if (a.bufferView >= views.size()) throw std::runtime_error("glTF: bufferView index out of range");
        const BufferView &v = views[a.bufferView];
        if (v.buffer >= buffers.size()) throw std::runtime_error("glTF: buffer index out of range");
        size_t off = v.byteOffset + (use_accessor_offset ? a.byteOffset : 0);
        if (off + a.count * elem_bytes > buffers[v.buffer].size()) throw std::runtime_error("glTF: accessor overruns its buffer");
        count = a.count;
        return buffers[v.buffer].data() + off;
    }
    std::vector<float> load_floats(size_t index, int comps) { // loadVec2s/3s/4s, sceneio.cpp:136-191
        // Extension: an interleaved bufferView (byteStride larger than the element) is gathered; the reference reads every
        // accessor as tightly packed (sceneio.cpp:136-191) and would produce garbage vertices for such a file.
        const size_t elem = 4 * (size_t)comps;
        size_t stride = elem;
        if (index < accessors.size() && accessors[index].bufferView < views.size() && views[accessors[index].bufferView].byteStride > elem)
            stride = views[accessors[index].bufferView].byteStride;
        size_t count;
        if (stride == elem) {
            const uint8_t *p = accessor_ptr(index, elem, count, true);
            std::vector<float> out(count * comps);
            memcpy(out.data(), p, out.size() * 4);
            return out;
        }
        const uint8_t *p = accessor_ptr(index, 0, count, true);
        const Accessor &a = accessors[index];
        const BufferView &v = views[a.bufferView];
        size_t off = v.byteOffset + a.byteOffset;
        if (count && off + (count - 1) * stride + elem > buffers[v.buffer].size()) throw std::runtime_error("glTF: strided accessor overruns its buffer");
        std::vector<float> out(count * comps);
        for (size_t i = 0; i < count; i++) memcpy(out.data() + i * comps, p + i * stride, elem);
        return out;
    }

    void load() {
        doc = Json::parse([&] { auto f = read_file(path); return std::string(f.begin(), f.end()); }());
        std::string base = dir_of(path);
        // loadBuffers, sceneio.cpp:14-26
        if (const Json *bs = doc.find("buffers"))
            for (const Json &b : bs->arr) {
                size_t sz = b.at("byteLength").as_uint();
                std::vector<uint8_t> data = read_file(base + b.at("uri").as_string());
                if (data.size() < sz) throw std::runtime_error("glTF: buffer file shorter than byteLength: " + b.at("uri").as_string());
                data.resize(sz);
                buffers.push_back(std::move(data));
            }
        // loadTextureImages, sceneio.cpp:329-336 (hw8 only)
        if (flavor == RT_INTEGRATOR_HW8)
            if (const Json *imgs = doc.find("images"))
                for (const Json &im : imgs->arr) {
                    int w, h;
                    std::vector<uint8_t> rgb;
                    load_image_rgb8(base + im.at("uri").as_string(), w, h, rgb);
                    hs->image_data.push_back(std::move(rgb));
                    hs->images.push_back(rt_image{w, h, nullptr});
                }
        // loadBufferViews, sceneio.cpp:28-37 (byteOffset defaulted to 0 instead of required)
        if (const Json *vs = doc.find("bufferViews"))
            for (const Json &v : vs->arr)
                views.push_back(BufferView{v.at("buffer").as_uint(), v.at("byteLength").as_uint(), v.has("byteOffset") ? v.at("byteOffset").as_uint() : 0u,
                                           v.has("byteStride") ? v.at("byteStride").as_uint() : 0u});
        load_nodes();
        // restoreNodeParents + calculateTransitions, sceneio.cpp:117-134
        for (size_t i = 0; i < nodes.size(); i++)
            for (size_t c : nodes[i].children) {
                if (c >= nodes.size()) throw std::runtime_error("glTF: child node index out of range");
                nodes[c].parent = i;
            }
        for (Node &n : nodes) {
            n.total = *n.local;
            std::optional<size_t> par = n.parent;
            size_t guard = 0;
            while (par.has_value()) {
                n.total = nodes[*par].local->compose(n.total);
                par = nodes[*par].parent;
                if (++guard > nodes.size()) throw std::runtime_error("glTF: node hierarchy has a cycle");
            }
        }
        // loadMeshes, sceneio.cpp:82-98
        if (const Json *ms = doc.find("meshes"))
            for (const Json &m : ms->arr) {
                std::vector<Prim> prims;
                for (const Json &p : m.at("primitives").arr) {
                    const Json &at = p.at("attributes");
                    Prim pr;
                    pr.position = at.at("POSITION").as_uint();
                    if (at.has("TEXCOORD_0")) pr.texcoord = at.at("TEXCOORD_0").as_uint();
                    if (at.has("NORMAL")) pr.normal = at.at("NORMAL").as_uint();
                    if (at.has("TANGENT")) pr.tangent = at.at("TANGENT").as_uint();
                    pr.indices = p.at("indices").as_uint();
                    pr.material = p.at("material").as_uint();
                    prims.push_back(pr);
                }
                meshes.push_back(std::move(prims));
            }
        // loadAccessors, sceneio.cpp:100-115
        if (const Json *as = doc.find("accessors"))
            for (const Json &a : as->arr)
                accessors.push_back(Accessor{a.at("bufferView").as_uint(), a.at("count").as_uint(), a.at("componentType").as_uint(),
                                             a.at("type").as_string(), a.has("byteOffset") ? (size_t)a.at("byteOffset").as_float() : 0});
        load_materials();
        load_figures();
        load_camera();
        // loadTextureDescs, sceneio.cpp:338-346
        if (flavor == RT_INTEGRATOR_HW8)
            if (const Json *ts = doc.find("textures"))
                for (const Json &t : ts->arr) {
                    uint32_t src = t.at("source").as_uint();
                    if (src >= hs->images.size()) throw std::runtime_error("glTF: texture source out of range");
                    hs->texture_source.push_back(src);
                }
        for (const rt_material &m : hs->materials)
            for (int32_t t : {m.base_color_texture, m.emissive_texture, m.metallic_roughness_texture, m.normal_texture})
                if (t >= (int32_t)hs->texture_source.size()) throw std::runtime_error("glTF: material texture index out of range");
    }

    void load_nodes() { // sceneio.cpp:39-80
        const Json *ns = doc.find("nodes");
        if (!ns) return;
        for (const Json &n : ns->arr) {
            Node cur;
            if (n.has("mesh")) cur.mesh = n.at("mesh").as_uint();
            if (n.has("camera")) cur.camera = n.at("camera").as_uint();
            if (n.has("rotation")) for (int i = 0; i < 4; i++) cur.rotation[i] = n.at("rotation").idx(i).as_float();
            if (n.has("translation")) cur.translation = F3{n.at("translation").idx(0).as_float(), n.at("translation").idx(1).as_float(), n.at("translation").idx(2).as_float()};
            if (n.has("scale")) cur.scale = F3{n.at("scale").idx(0).as_float(), n.at("scale").idx(1).as_float(), n.at("scale").idx(2).as_float()};
            if (n.has("children")) for (const Json &c : n.at("children").arr) cur.children.push_back(c.as_uint());
            if (n.has("matrix")) { // column-major in the file -> [row][col]
                Mat4 mm;
                for (size_t i = 0; i < 16; i++) mm.m[i % 4][i / 4] = n.at("matrix").idx(i).as_float();
                cur.local = mm;
            }
            if (!cur.local.has_value()) cur.local = from_trs(cur.translation, cur.rotation, cur.scale);
            nodes.push_back(std::move(cur));
        }
    }

    void load_materials() { // hw8: sceneio.cpp:193-245 ; hw6: hw6/src/sceneio.cpp:149-185
        const Json *ms = doc.find("materials");
        if (!ms) return;
        for (const Json &m : ms->arr) {
            rt_material cur{};
            cur.base_color[0] = cur.base_color[1] = cur.base_color[2] = 1;
            cur.metallic_factor = 1; cur.roughness_factor = 1;
            cur.base_color_texture = cur.emissive_texture = cur.metallic_roughness_texture = cur.normal_texture = -1;
            cur.kind = RT_MAT_DIFFUSE; cur.ior = 1.5f; // GltfMaterial::ior, hw6/src/include/gltf_structs.h:36
            float alpha = 1.0f;
            if (const Json *pbr = m.find("pbrMetallicRoughness")) {
                if (const Json *c = pbr->find("baseColorFactor")) {
                    for (int i = 0; i < 3; i++) cur.base_color[i] = c->idx(i).as_float();
                    if (flavor == RT_INTEGRATOR_HW6) alpha = c->idx(3).as_float();
                }
                if (pbr->has("metallicFactor")) cur.metallic_factor = pbr->at("metallicFactor").as_float();
                if (flavor == RT_INTEGRATOR_HW8) {
                    if (pbr->has("baseColorTexture")) cur.base_color_texture = (int32_t)pbr->at("baseColorTexture").at("index").as_uint();
                    if (pbr->has("roughnessFactor")) cur.roughness_factor = pbr->at("roughnessFactor").as_float();
                    if (pbr->has("metallicRoughnessTexture")) cur.metallic_roughness_texture = (int32_t)pbr->at("metallicRoughnessTexture").at("index").as_uint();
                }
            }
            if (const Json *e = m.find("emissiveFactor")) for (int i = 0; i < 3; i++) cur.emission[i] = e->idx(i).as_float();
            if (flavor == RT_INTEGRATOR_HW8) {
                if (m.has("emissiveTexture")) cur.emissive_texture = (int32_t)m.at("emissiveTexture").at("index").as_uint();
                if (m.has("normalTexture")) cur.normal_texture = (int32_t)m.at("normalTexture").at("index").as_uint();
            }
            if (const Json *ext = m.find("extensions"))
                if (const Json *es = ext->find("KHR_materials_emissive_strength")) {
                    float k = es->at("emissiveStrength").as_float();
                    for (int i = 0; i < 3; i++) cur.emission[i] = k * cur.emission[i];
                }
            if (flavor == RT_INTEGRATOR_HW6) { // hw6/src/sceneio.cpp:178-182
                if (alpha < 1) cur.kind = RT_MAT_DIELECTRIC;
                else if (cur.metallic_factor > 0) cur.kind = RT_MAT_METALLIC;
            }
            hs->materials.push_back(cur);
        }
    }

    void load_figures() { // loadFiguresFromNodes + loadFigures, sceneio.cpp:247-310
        const bool full = flavor == RT_INTEGRATOR_HW8;
        for (const Node &node : nodes) {
            if (!node.mesh.has_value()) continue;
            if (*node.mesh >= meshes.size()) throw std::runtime_error("glTF: mesh index out of range");
            for (const Prim &pr : meshes[*node.mesh]) {
                if (pr.material >= hs->materials.size()) throw std::runtime_error("glTF: material index out of range");
                std::vector<float> pos = load_floats(pr.position, 3), uv, nrm, tan;
                size_t nv = pos.size() / 3;
                if (full) {
                    // The reference requires all four attributes (hw8/src/sceneio.cpp:88-91, it crashes
                    // otherwise).  Extension: a missing TEXCOORD_0 reads as (0,0) and a missing TANGENT as
                    // (1,0,0,1), which keeps the default normal-map sample an identity; NORMAL stays required.
                    if (!pr.normal) throw std::runtime_error("glTF: hw8 scenes need NORMAL on every primitive (hw8/src/sceneio.cpp:90)");
                    nrm = load_floats(*pr.normal, 3);
                    if (pr.texcoord) uv = load_floats(*pr.texcoord, 2); else uv.assign(nv * 2, 0.f);
                    if (pr.tangent) tan = load_floats(*pr.tangent, 4);
                    else { tan.assign(nv * 4, 0.f); for (size_t v = 0; v < nv; v++) { tan[4 * v] = 1.f; tan[4 * v + 3] = 1.f; } }
                }
                size_t n_idx;
                const Accessor &ia = accessors.at(pr.indices);
                // 5123 (u16) and 5125 (u32) as the reference; 5121 (u8) is an extension (the reference would misread it as u32)
                size_t isz = ia.componentType == 5123 ? 2 : (ia.componentType == 5121 ? 1 : 4);
                if (ia.componentType != 5121 && ia.componentType != 5123 && ia.componentType != 5125) throw std::runtime_error("glTF: index componentType must be 5121, 5123 or 5125");
                // Extension: the index accessor's byteOffset is honoured (the reference ignores it, sceneio.cpp:258-269, and would
                // read the wrong triangles from a bufferView shared by several index accessors; its shipped scenes all have 0).
                const uint8_t *ip = accessor_ptr(pr.indices, isz, n_idx, true);
                const Mat4 &M = node.total;
                Mat4 NM = M.inverted().transposed();
                F3 shift = M.apply(F3{0, 0, 0});
                for (size_t i = 0; i + 2 < n_idx; i += 3) {
                    size_t id[3];
                    for (int k = 0; k < 3; k++) {
                        if (isz == 1) id[k] = ip[i + k];
                        else if (isz == 2) { uint16_t v; memcpy(&v, ip + 2 * (i + k), 2); id[k] = v; }
                        else { uint32_t v; memcpy(&v, ip + 4 * (i + k), 4); id[k] = v; }
                        if (id[k] >= nv || (full && (id[k] >= uv.size() / 2 || id[k] >= nrm.size() / 3 || id[k] >= tan.size() / 4)))
                            throw std::runtime_error("glTF: vertex index out of range");
                    }
                    const size_t corner[3] = {id[0], id[2], id[1]}; // Figure(v1, v3, v2)
                    for (int k = 0; k < 3; k++) {
                        size_t v = corner[k];
                        F3 p = M.apply(F3{pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]});
                        hs->positions.insert(hs->positions.end(), {p.x, p.y, p.z});
                        if (full) {
                            hs->texcoords.insert(hs->texcoords.end(), {uv[2 * v], uv[2 * v + 1]});
                            F3 n = normalized(NM.apply(F3{nrm[3 * v], nrm[3 * v + 1], nrm[3 * v + 2]}));
                            hs->normals.insert(hs->normals.end(), {n.x, n.y, n.z});
                            F3 t = normalized(sub(M.apply(F3{tan[4 * v], tan[4 * v + 1], tan[4 * v + 2]}), shift));
                            hs->tangents.insert(hs->tangents.end(), {t.x, t.y, t.z, tan[4 * v + 3]});
                        }
                    }
                    hs->material_index.push_back((uint32_t)pr.material);
                }
            }
        }
    }

    void load_camera() { // sceneio.cpp:312-326
        F3 up{0, 1, 0}, fwd{0, 0, -1}, right{1, 0, 0}, pos{0, 0, 0};
        float fov = 0;
        for (const Node &n : nodes)
            if (n.camera.has_value()) {
                fov = doc.at("cameras").idx(*n.camera).at("perspective").at("yfov").as_float();
                pos = n.total.apply(F3{0, 0, 0});
                up = sub(n.total.apply(up), pos);
                right = sub(n.total.apply(right), pos);
                fwd = sub(n.total.apply(fwd), pos);
            }
        rt_camera &c = hs->camera;
        c.position[0] = pos.x; c.position[1] = pos.y; c.position[2] = pos.z;
        c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
        c.right[0] = right.x; c.right[1] = right.y; c.right[2] = right.z;
        c.forward[0] = fwd.x; c.forward[1] = fwd.y; c.forward[2] = fwd.z;
        c.fov_y = fov; c.fov_x = 0;
    }
};

} // namespace

rt_host_scene *load_gltf(const std::string &path, int flavor) {
    if (flavor != RT_INTEGRATOR_HW6 && flavor != RT_INTEGRATOR_HW8) throw std::runtime_error("rt_load_gltf: flavor must be RT_INTEGRATOR_HW6 or RT_INTEGRATOR_HW8");
    std::unique_ptr<rt_host_scene> hs(new rt_host_scene());
    Loader L;
    L.path = path; L.flavor = flavor; L.hs = hs.get();
    L.load();
    hs->finalize();
    return hs.release();
}

} // namespace rtamd
