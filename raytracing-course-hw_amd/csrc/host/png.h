#pragma once
#include <cstdint>
#include <string>
#include <vector>
namespace rtamd {
std::vector<uint8_t> read_file(const std::string &path);
// Decode a PNG byte stream to tightly packed RGB8 (throws std::runtime_error).
void decode_png(const std::vector<uint8_t> &file, int &width, int &height, std::vector<uint8_t> &rgb);
// Decode a baseline / extended-sequential JPEG byte stream to RGB8 following stb_image's pipeline (jpeg.cpp).
void decode_jpeg(const std::vector<uint8_t> &file, int &width, int &height, std::vector<uint8_t> &rgb);
// stbi_load(path, &w, &h, &ch, 3) equivalent: PNG or JPEG by magic bytes.
void load_image_rgb8(const std::string &path, int &width, int &height, std::vector<uint8_t> &rgb);
}
