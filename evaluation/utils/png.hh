// png.hh — a minimal PNG codec on zlib for the evaluation harness (the reference reads and writes its KITTI files through
// OpenCV's cv::imread / cv::imwrite, evaluation/utils/kitti.hh:11-12,70; OpenCV is not part of this engine).
// Reads non-interlaced gray / RGB / RGBA at 8 or 16 bits per sample; writes gray or RGB at 8 or 16 bits.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace minipng {

struct image {
  int width = 0, height = 0, channels = 0, depth = 0;  // depth: bits per sample (8 or 16)
  std::vector<uint8_t> bytes;                          // row-major, channels interleaved; 16-bit samples in HOST byte order
  bool ok() const { return width > 0 && height > 0; }
  // sample (row, col, channel) as an integer whatever the depth
  unsigned at(int r, int c, int k) const {
    const size_t i = (size_t(r) * width + c) * channels + k;
    return depth == 16 ? ((const uint16_t*)bytes.data())[i] : bytes[i];
  }
};

namespace detail {
inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
inline void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x)); }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
inline void chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* data, size_t n) {
  put32(out, uint32_t(n));
  const size_t start = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data, data + n);
  put32(out, uint32_t(crc32(0, out.data() + start, uInt(out.size() - start))));
}
}  // namespace detail

inline image read(const std::string& path) {
  image img;
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return img;
  std::vector<uint8_t> file;
  uint8_t buf[1 << 16];
  for (size_t n; (n = std::fread(buf, 1, sizeof buf, f)) > 0;) file.insert(file.end(), buf, buf + n);
  std::fclose(f);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8)) return img;
  int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> z;
  for (size_t p = 8; p + 12 <= file.size();) {
    const uint32_t n = detail::be32(&file[p]);
    const char* type = (const char*)&file[p + 4];
    if (p + 12 + n > file.size()) return img;
    const uint8_t* d = &file[p + 8];
    if (!std::memcmp(type, "IHDR", 4) && n >= 13) { w = int(detail::be32(d)); h = int(detail::be32(d + 4)); depth = d[8]; ctype = d[9]; interlace = d[12]; }
    else if (!std::memcmp(type, "IDAT", 4)) z.insert(z.end(), d, d + n);
    else if (!std::memcmp(type, "IEND", 4)) break;
    p += 12 + n;
  }
  const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 6 ? 4 : ctype == 4 ? 2 : 0;
  if (w <= 0 || h <= 0 || !ch || (depth != 8 && depth != 16) || interlace) return img;
  const int bpp = ch * depth / 8;
  const size_t stride = size_t(w) * bpp;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf rawlen = uLongf(raw.size());
  if (uncompress(raw.data(), &rawlen, z.data(), uLong(z.size())) != Z_OK || rawlen != raw.size()) return img;
  img.bytes.resize(stride * h);
  for (int r = 0; r < h; r++) {  // undo the per-scanline filters (PNG specification, section 9)
    const uint8_t* in = &raw[(stride + 1) * r];
    uint8_t* cur = &img.bytes[stride * r];
    const uint8_t* up = r ? cur - stride : nullptr;
    const int ft = in[0];
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= size_t(bpp) ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= size_t(bpp)) ? up[i - bpp] : 0;
      int x = in[1 + i];
      switch (ft) { case 1: x += a; break; case 2: x += b; break; case 3: x += (a + b) >> 1; break; case 4: x += detail::paeth(a, b, c); break; default: break; }
      cur[i] = uint8_t(x);
    }
  }
  if (depth == 16) { uint16_t* s = (uint16_t*)img.bytes.data(); for (size_t i = 0; i < img.bytes.size() / 2; i++) { const uint8_t* q = (const uint8_t*)&s[i]; s[i] = uint16_t((q[0] << 8) | q[1]); } }
  img.width = w; img.height = h; img.channels = ch; img.depth = depth;
  return img;
}

// pixels: h rows of w * channels samples (uint8_t, or uint16_t in host byte order when depth == 16); channels 1 or 3
inline bool write(const std::string& path, const void* pixels, int w, int h, int channels, int depth, int compression_level = 1) {
  if (w <= 0 || h <= 0 || (channels != 1 && channels != 3) || (depth != 8 && depth != 16)) return false;
  const size_t stride = size_t(w) * channels * depth / 8;
  std::vector<uint8_t> raw((stride + 1) * h);
  for (int r = 0; r < h; r++) {
    uint8_t* out = &raw[(stride + 1) * r];
    out[0] = 0;  // filter type "none"
    if (depth == 8) std::memcpy(out + 1, (const uint8_t*)pixels + stride * r, stride);
    else { const uint16_t* s = (const uint16_t*)pixels + size_t(w) * channels * r; for (int i = 0; i < w * channels; i++) { out[1 + 2 * i] = uint8_t(s[i] >> 8); out[2 + 2 * i] = uint8_t(s[i]); } }
  }
  uLongf zlen = compressBound(uLong(raw.size()));
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), uLong(raw.size()), compression_level) != Z_OK) return false;
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  detail::put32(ihdr, uint32_t(w)); detail::put32(ihdr, uint32_t(h));
  ihdr.push_back(uint8_t(depth)); ihdr.push_back(channels == 1 ? 0 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  detail::chunk(out, "IHDR", ihdr.data(), ihdr.size());
  detail::chunk(out, "IDAT", z.data(), zlen);
  detail::chunk(out, "IEND", nullptr, 0);
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  std::fclose(f);
  return ok;
}

}  // namespace minipng
