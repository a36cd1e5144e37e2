// Host-side scene container: owns the arrays an rt_scene_desc points into.
#pragma once
#include "../../../include/rtamd.h"
#include <cstdint>
#include <string>
#include <vector>

struct rt_host_scene {
    std::vector<float> positions, texcoords, normals, tangents;
    std::vector<uint32_t> material_index;
    std::vector<rt_material> materials;
    std::vector<uint32_t> texture_source;
    std::vector<std::vector<uint8_t>> image_data;
    std::vector<rt_image> images;
    std::vector<uint8_t> env_data;
    rt_image env{0, 0, nullptr};
    bool has_env = false;
    std::vector<rt_primitive> primitives;
    std::vector<rt_light> lights;
    float ambient[3] = {0, 0, 0};
    rt_camera camera{};
    float bg[3] = {0, 0, 0};
    rt_scene_desc desc{};

    // Re-point desc at the vectors (call after any mutation).
    void finalize() {
        desc = rt_scene_desc{};
        desc.struct_size = sizeof(rt_scene_desc);
        desc.n_triangles = (uint32_t)material_index.size();
        desc.positions = positions.data();
        desc.texcoords = texcoords.empty() ? nullptr : texcoords.data();
        desc.normals = normals.empty() ? nullptr : normals.data();
        desc.tangents = tangents.empty() ? nullptr : tangents.data();
        desc.material_index = material_index.data();
        desc.n_materials = (uint32_t)materials.size();
        desc.materials = materials.data();
        desc.n_textures = (uint32_t)texture_source.size();
        desc.texture_source = texture_source.data();
        for (size_t i = 0; i < images.size(); i++) images[i].rgb = image_data[i].data();
        desc.n_images = (uint32_t)images.size();
        desc.images = images.data();
        if (has_env) { env.rgb = env_data.data(); desc.environment_map = &env; }
        desc.n_primitives = (uint32_t)primitives.size();
        desc.primitives = primitives.data();
        desc.camera = camera;
        desc.bg_color[0] = bg[0]; desc.bg_color[1] = bg[1]; desc.bg_color[2] = bg[2];
        desc.n_lights = (uint32_t)lights.size();
        desc.lights = lights.data();
        desc.ambient_light[0] = ambient[0]; desc.ambient_light[1] = ambient[1]; desc.ambient_light[2] = ambient[2];
    }
};

namespace rtamd {
void set_error(const std::string &msg);
rt_host_scene *load_gltf(const std::string &path, int flavor);
rt_host_scene *load_txt(const std::string &path, int flavor, int32_t *w, int32_t *h, int32_t *samples, int32_t *depth);
}
