// kitti.hh — KITTI optical-flow I/O and error statistics (reference: evaluation/utils/kitti.hh:9-135), against the drop-in
// <vpp/...> headers.  Files go through the minimal PNG codec in png.hh instead of OpenCV.  Flow vectors are (row, col) like
// every coordinate in vpp; a KITTI flow PNG stores R = column flow, G = row flow, B = valid, as (value - 2^15) / 64.
#pragma once
#include <algorithm>
#include <cmath>
#include <iomanip>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <vpp/vpp.hh>

#include "png.hh"

namespace kitti {
using namespace vpp;

// 8-bit image file -> image2d<vuchar3> (a gray file is replicated to three channels, as cv::imread's default flag does: kitti.hh:38-39)
inline image2d<vuchar3> load_image(const std::string& filename) {
  const minipng::image png = minipng::read(filename);
  if (!png.ok() || png.depth != 8) return image2d<vuchar3>();
  image2d<vuchar3> out(png.height, png.width);
  for (int r = 0; r < png.height; r++)
    for (int c = 0; c < png.width; c++) {
      if (png.channels >= 3) out(r, c) = vuchar3((unsigned char)png.at(r, c, 0), (unsigned char)png.at(r, c, 1), (unsigned char)png.at(r, c, 2));
      else { const unsigned char g = (unsigned char)png.at(r, c, 0); out(r, c) = vuchar3(g, g, g); }
    }
  return out;
}

// kitti.hh:9-23: (row flow, col flow, valid)
inline image2d<vfloat3> load_flow(const std::string& filename) {
  const minipng::image png = minipng::read(filename);
  if (!png.ok() || png.depth != 16 || png.channels < 3) return image2d<vfloat3>();
  image2d<vfloat3> out(png.height, png.width);
  for (int r = 0; r < png.height; r++)
    for (int c = 0; c < png.width; c++)
      out(r, c) = vfloat3((float(png.at(r, c, 1)) - (1 << 15)) / 64.f, (float(png.at(r, c, 0)) - (1 << 15)) / 64.f, float(png.at(r, c, 2)));
  return out;
}

// kitti.hh:25-52: training/image_0/NNNNNN_10.png, _11.png and training/flow_noc/NNNNNN_10.png for N pairs
template <class F> void foreach_training_pair(const std::string& kitti_root, int N, F&& fun) {
  const std::string images_root = kitti_root + "/training/image_0/", ref_root = kitti_root + "/training/flow_noc/";
  for (int i = 0; i < N; i++) {
    std::stringstream ss;
    ss << std::setfill('0') << std::setw(6) << i;
    const std::string i1_filename = images_root + ss.str() + "_10.png", i2_filename = images_root + ss.str() + "_11.png";
    const std::string ref_filename = ref_root + ss.str() + "_10.png";
    image2d<vuchar3> i1 = load_image(i1_filename), i2 = load_image(i2_filename);
    if (!i1.has_data()) throw std::runtime_error(std::string("Cannot read image ") + i1_filename);
    if (!i2.has_data()) throw std::runtime_error(std::string("Cannot read image ") + i2_filename);
    image2d<vfloat3> ref_flow = load_flow(ref_filename);
    fun(i1, i2, ref_flow);
  }
}

// kitti.hh:54-73 (a component is stored as 64 * v + 2^15, clamped to the 16-bit range)
inline uint16_t encode(float v) { const float q = v * 64.0f + 32768.0f; return (uint16_t)(q < 0.0f ? 0.0f : (q > 65535.0f ? 65535.0f : q)); }
inline void write_flow(const std::string& filename, const image2d<vfloat2>& flow, const image2d<char>& has_flow) {
  const int nr = flow.nrows(), nc = flow.ncols();
  std::vector<uint16_t> out(size_t(nr) * nc * 3, 0);
  for (int r = 0; r < nr; r++)
    for (int c = 0; c < nc; c++)
      if (has_flow(r, c)) {
        const vfloat2 f = flow(r, c);
        uint16_t* px = &out[(size_t(r) * nc + c) * 3];
        px[0] = encode(f[1]);  // red: column flow
        px[1] = encode(f[0]);  // green: row flow
        px[2] = 1;             // blue: valid
      }
  if (!minipng::write(filename, out.data(), nc, nr, 3, 16, 0)) throw std::runtime_error(std::string("Cannot write ") + filename);
}

struct flow_error_result {
  float n1, n3, n5, n10;          // % of evaluated vectors whose end-point error exceeds 1 / 3 / 5 / 10 px
  float avg;                      // mean end-point error
  std::vector<float> errors;      // sorted
  image2d<unsigned char> errors_map;
  float density;                  // % of pixels carrying a flow vector
};

// kitti.hh:75-135, same raster order of accumulation
inline flow_error_result flow_error_stats(const image2d<vfloat3>& flow, const image2d<vfloat3>& ref) {
  float error_sum = 0.f;
  std::vector<float> errors;
  image2d<unsigned char> errors_map(flow.domain());
  fill(errors_map, (unsigned char)0);
  int cpt = 0;
  for (int r = 0; r < flow.nrows(); r++)
    for (int c = 0; c < flow.ncols(); c++) {
      const vfloat3 f = flow(r, c), g = ref(r, c);
      if (f[2] > 0.f) cpt++;
      if (f[2] > 0.f && g[2] > 0.f) {
        const float err = (f.segment<2>(0) - g.segment<2>(0)).norm();
        error_sum += err;
        errors.push_back(err);
        errors_map(r, c) = (unsigned char)std::min(err * 20.f, 255.f);
      }
    }
  std::sort(errors.begin(), errors.end());
  const float n = float(errors.size());
  // share of the evaluated vectors whose error is strictly above t (the list is sorted: everything past upper_bound(t))
  auto percent_above = [&](float t) { const auto k = errors.end() - std::upper_bound(errors.begin(), errors.end(), t); return k ? 100 * float(k) / n : 0.f; };
  return flow_error_result{percent_above(1.f), percent_above(3.f), percent_above(5.f), percent_above(10.f),
                           error_sum / (errors.empty() ? 1.f : n), errors, errors_map, 100.f * float(cpt) / (flow.nrows() * flow.ncols())};
}
}  // namespace kitti
