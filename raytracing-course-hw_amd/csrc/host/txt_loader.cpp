// .txt scene loader (hw1/hw3 grammar) — placeholder until the HW1/HW3 integrators land.
#include "host_scene.h"
#include <stdexcept>
namespace rtamd {
rt_host_scene *load_txt(const std::string &, int, int32_t *, int32_t *, int32_t *, int32_t *) {
    throw std::runtime_error(".txt scenes are not implemented in this build");
}
}
