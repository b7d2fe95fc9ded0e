// png_io.cpp -- minimal PNG reader / writer on zlib, replacing the OpenCV imread / imwrite calls of the reference's
// dataset code (CORE/src/nerf_data.cu:151-221: 8-bit colour, 16-bit depth, 8-bit instance images) and of its test-image
// output (CORE/src/nerf.cu:335-349).  Non-interlaced, bit depth 8 or 16, colour types 0 (gray), 2 (RGB), 4 (gray+alpha), 6 (RGBA).
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "png_io.h"

namespace mon {

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
static int paeth(int a, int b, int c) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

bool png_read(const std::string& path, PngImage& img, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "Can not read image... path: " + path; return false; }
    std::vector<uint8_t> file; uint8_t buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) { err = "not a PNG file: " + path; return false; }
    size_t pos = 8; std::vector<uint8_t> idat; int color = -1, interlace = 0; img.width = img.height = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]); const char* type = (const char*)&file[pos + 4];
        if (pos + 12 + len > file.size()) break;
        const uint8_t* data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) { img.width = (int)be32(data); img.height = (int)be32(data + 4); img.bit_depth = data[8]; color = data[9]; interlace = data[12]; }
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (img.width <= 0 || img.height <= 0 || img.width > 32768 || img.height > 32768 || (uint64_t)img.width * (uint64_t)img.height > (1ull << 26) || idat.empty() || interlace != 0 || (img.bit_depth != 8 && img.bit_depth != 16) ||
        !(color == 0 || color == 2 || color == 4 || color == 6)) {
        err = "unsupported PNG format: " + path; return false;
    }
    img.channels = color == 0 ? 1 : (color == 2 ? 3 : (color == 4 ? 2 : 4));
    const size_t bpp = (size_t)img.channels * img.bit_depth / 8, stride = bpp * img.width;
    std::vector<uint8_t> raw((stride + 1) * img.height);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) { err = "PNG inflate failed: " + path; return false; }
    img.data.assign(stride * img.height, 0);
    for (int y = 0; y < img.height; ++y) {
        const uint8_t ft = raw[(stride + 1) * y]; const uint8_t* in = &raw[(stride + 1) * y + 1];
        if (ft > 4) { err = "PNG: bad filter type in " + path; return false; }
        uint8_t* cur = &img.data[stride * y]; const uint8_t* up = y ? &img.data[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = in[x];
            switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break; default: break; }
            cur[x] = (uint8_t)v;
        }
    }
    return true;
}

static void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4); out.insert(out.end(), data.begin(), data.end());
    put32(out, (uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start)));
}

bool png_write(const std::string& path, int width, int height, int channels, int bit_depth, const uint8_t* pixels_big_endian, std::string& err) {
    if (!(channels == 1 || channels == 3) || !(bit_depth == 8 || bit_depth == 16)) { err = "png_write: unsupported format"; return false; }
    const size_t stride = (size_t)width * channels * bit_depth / 8;
    std::vector<uint8_t> raw((stride + 1) * height);
    for (int y = 0; y < height; ++y) { raw[(stride + 1) * y] = 0; std::memcpy(&raw[(stride + 1) * y + 1], pixels_big_endian + stride * y, stride); }
    uLongf clen = compressBound((uLong)raw.size()); std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "png_write: deflate failed"; return false; }
    comp.resize(clen);
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a }, ihdr;
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height); ihdr.push_back((uint8_t)bit_depth); ihdr.push_back(channels == 1 ? 0 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", comp); chunk(out, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "png_write: cannot open " + path; return false; }
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    if (!ok) err = "png_write: short write " + path;
    return ok;
}

}  // namespace mon
