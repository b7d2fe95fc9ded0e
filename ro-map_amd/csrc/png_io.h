// png_io.h -- minimal PNG reader / writer (zlib); see png_io.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mon {

struct PngImage { int width = 0, height = 0, channels = 0, bit_depth = 0; std::vector<uint8_t> data; };   // 16-bit samples big-endian, as in the file

bool png_read(const std::string& path, PngImage& img, std::string& err);
bool png_write(const std::string& path, int width, int height, int channels, int bit_depth, const uint8_t* pixels_big_endian, std::string& err);

}  // namespace mon
