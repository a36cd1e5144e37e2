// HIP error handling shared by the translation units of the library: a failed call throws, the C-ABI entry points map it to RT_ERR_HIP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdexcept>
#include <string>

namespace rtamd {
struct HipError : std::runtime_error {
    explicit HipError(const std::string &m) : std::runtime_error(m) {}
};
} // namespace rtamd
#define HIP_CHECK(expr)                                                                                          \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) throw rtamd::HipError(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)
