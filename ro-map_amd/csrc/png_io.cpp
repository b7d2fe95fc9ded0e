// png_io.cpp -- minimal PNG reader / writer on zlib, replacing the OpenCV imread / imwrite calls of the reference's
// dataset code (CORE/src/nerf_data.cu:151-221: 8-bit colour, 16-bit depth, 8-bit instance images) and of its test-image
// output (CORE/src/nerf.cu:335-349).  Reader: all PNG colour types and bit depths, Adam7 included; writer: 8/16-bit gray or RGB.
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
static int paeth(int a, int b, int c) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

// Un-filters `rows` scanlines of `stride` bytes each (filter byte in front of every line) from `in` into `out`.
static bool unfilter(const uint8_t* in, uint8_t* out, size_t stride, size_t rows, size_t bpp) {
    for (size_t y = 0; y < rows; ++y) {
        const uint8_t ft = in[(stride + 1) * y]; const uint8_t* src = &in[(stride + 1) * y + 1];
        if (ft > 4) return false;
        uint8_t* cur = &out[stride * y]; const uint8_t* up = y ? &out[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = src[x];
            switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break; default: break; }
            cur[x] = (uint8_t)v;
        }
    }
    return true;
}

// Reads what cv::imread(..., IMREAD_UNCHANGED) reads of a PNG (nerf_data.cu:151-221): every colour type (gray, RGB, palette,
// gray+alpha, RGBA), bit depths 1/2/4/8/16, Adam7 interlacing.  Palette and sub-byte images come out expanded to 8 bits per
// sample the way libpng's expand transforms do it (palette -> RGB, or RGBA when a tRNS chunk is present; 1/2/4-bit gray scaled to
// 0..255), so an instance mask stored as a palette PNG yields its palette colours, exactly like the reference's loader.
bool png_read(const std::string& path, PngImage& img, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "Can not read image... path: " + path; return false; }
    std::vector<uint8_t> file; uint8_t buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) { err = "not a PNG file: " + path; return false; }
    size_t pos = 8; std::vector<uint8_t> idat, plte, trns; int color = -1, interlace = 0, depth = 0; bool have_ihdr = false; img.width = img.height = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]); const char* type = (const char*)&file[pos + 4];
        if ((uint64_t)pos + 12 + len > file.size()) break;
        const uint8_t* data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13 || have_ihdr) { err = "PNG: bad IHDR chunk in " + path; return false; }
            have_ihdr = true; img.width = (int)be32(data); img.height = (int)be32(data + 4); depth = data[8]; color = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) { err = "PNG: unknown compression / filter method in " + path; return false; }
        }
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    const bool depth_ok = (color == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16))
            || (color == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((color == 2 || color == 4 || color == 6) && (depth == 8 || depth == 16));
    if (!have_ihdr || img.width <= 0 || img.height <= 0 || img.width > 32768 || img.height > 32768 || (uint64_t)img.width * (uint64_t)img.height > (1ull << 26)
            || idat.empty() || interlace > 1 || !depth_ok ||
        (color == 3 && (plte.empty() || plte.size() % 3 != 0))) {
        err = "unsupported PNG format: " + path; return false;
    }
    const int file_ch = color == 0 ? 1 : (color == 2 ? 3 : (color == 3 ? 1 : (color == 4 ? 2 : 4)));
    const size_t bits_pp = (size_t)file_ch * depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;           // filter distance in bytes
    const auto line_bytes = [&](size_t w) { return (w * bits_pp + 7) / 8; };
    // pass geometry: one pass (non-interlaced) or the seven Adam7 passes
    static const int ax0[7] = { 0, 4, 0, 2, 0, 1, 0 }, ay0[7] = { 0, 0, 4, 0, 2, 0, 1 }, adx[7] = { 8, 8, 4, 4, 2, 2, 1 }, ady[7] = { 8, 8, 8, 4, 4, 2, 2 };
    const int n_pass = interlace ? 7 : 1;
    size_t pw[7], ph[7], total = 0;
    for (int p = 0; p < n_pass; ++p) {
        pw[p] = interlace ? ((size_t)img.width + adx[p] - 1 - ax0[p]) / adx[p] : (size_t)img.width;
        ph[p] = interlace ? ((size_t)img.height + ady[p] - 1 - ay0[p]) / ady[p] : (size_t)img.height;
        if (interlace && (img.width <= ax0[p] || img.height <= ay0[p])) pw[p] = ph[p] = 0;
        if (pw[p] && ph[p]) total += (line_bytes(pw[p]) + 1) * ph[p];
    }
    std::vector<uint8_t> raw(total);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) { err = "PNG inflate failed: " + path;
        return false; }
    // samples of the file, un-interlaced, one sample per output slot: 8-bit (sub-byte depths unpacked, not yet scaled) or 16-bit big-endian
    const size_t sample_bytes = depth == 16 ? 2 : 1, px_bytes = (size_t)file_ch * sample_bytes;
    std::vector<uint8_t> samples((size_t)img.width * img.height * px_bytes);
    size_t off = 0; std::vector<uint8_t> lines;
    for (int p = 0; p < n_pass; ++p) {
        if (!pw[p] || !ph[p]) continue;
        const size_t stride = line_bytes(pw[p]);
        lines.assign(stride * ph[p], 0);
        if (!unfilter(&raw[off], lines.data(), stride, ph[p], bpp)) { err = "PNG: bad filter type in " + path; return false; }
        off += (stride + 1) * ph[p];
        for (size_t y = 0; y < ph[p]; ++y)
            for (size_t x = 0; x < pw[p]; ++x) {
                const size_t ox = interlace ? (size_t)ax0[p] + x * adx[p] : x, oy = interlace ? (size_t)ay0[p] + y * ady[p] : y;
                uint8_t* dst = &samples[(oy * img.width + ox) * px_bytes]; const uint8_t* ln = &lines[stride * y];
                if (depth >= 8) std::memcpy(dst, ln + x * px_bytes, px_bytes);
                else { const size_t bit = x * depth; dst[0] = (uint8_t)((ln[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u)); }
            }
    }
    const size_t px = (size_t)img.width * img.height;
    if (color == 3) {                                   // palette -> RGB (RGBA with tRNS), 8 bits
        const size_t n_pal = plte.size() / 3; const bool alpha = !trns.empty();
        img.channels = alpha ? 4 : 3; img.bit_depth = 8; img.data.assign(px * img.channels, 0);
        for (size_t i = 0; i < px; ++i) {
            const size_t k = samples[i];
            if (k >= n_pal) { err = "PNG: palette index out of range in " + path; return false; }
            uint8_t* o = &img.data[i * img.channels]; o[0] = plte[3 * k]; o[1] = plte[3 * k + 1]; o[2] = plte[3 * k + 2];
            if (alpha) o[3] = k < trns.size() ? trns[k] : 255;
        }
    } else if (depth < 8) {                             // 1/2/4-bit gray -> 8-bit gray, scaled to the full range
        img.channels = 1; img.bit_depth = 8; img.data.resize(px);
        const unsigned mul = 255u / ((1u << depth) - 1u);
        for (size_t i = 0; i < px; ++i) img.data[i] = (uint8_t)(samples[i] * mul);
    } else { img.channels = file_ch; img.bit_depth = depth; img.data.swap(samples); }
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
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height); ihdr.push_back((uint8_t)bit_depth); ihdr.push_back(channels == 1 ? 0 : 2); ihdr.push_back(0);
    ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", comp); chunk(out, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "png_write: cannot open " + path; return false; }
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    if (!ok) err = "png_write: short write " + path;
    return ok;
}

}  // namespace mon
