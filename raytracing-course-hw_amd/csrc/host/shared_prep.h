// One host-side preparation of an hw8 / hw7 scene shared by several rt_scene objects (rt_multi_create: one scene per device).
#pragma once
#include "scene_prep.h"
#include <exception>
#include <mutex>

struct rt_scene;

namespace rtamd {

struct SharedPrep {
    std::once_flag once;          // the first scene_create_shared call prepares, the others wait here
    PreparedScene P;              // read-only once prepared
    std::exception_ptr error;     // what prepare_scene threw, rethrown in every caller
};

// rt_scene_create (include/rtamd.h) with the preparation taken from / left in `shared` (nullable).  hw6 and .txt scenes prepare per call.
int scene_create_shared(const rt_scene_desc *desc, rt_scene **out, SharedPrep *shared);

} // namespace rtamd
