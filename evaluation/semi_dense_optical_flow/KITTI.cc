// KITTI.cc — accuracy / runtime of semi_dense_optical_flow on KITTI flow training pairs (reference:
// evaluation/semi_dense_optical_flow/KITTI.cc:103-194), written against the drop-in <vpp/...> headers: every algorithm call
// below (rgb_to_graylevel, fast9, semi_dense_optical_flow) runs on the MI355X engine.
//   usage: KITTI kitti_root n_images config_file result_file
// config_file: one "name value" (or "name: value", "name = value") per line, '#' starts a comment; names and defaults are the
// reference's (KITTI.cc:112-119): nscales 1, winsize 9, propagation 2, min_scale 0, patchsize 5, detector_th 10, block_size 10.
// result_file: "runtime: <mean microseconds per pair>", "errors: <mean % of vectors off by more than 3 px>",
// "nkeypoints: <mean keypoints per pair>" (the three values the reference hands to gpof::write_results, KITTI.cc:190-193),
// followed by the parameters used.  The reference's parameter files go through the gpof library (not in its tree).
#include <cfloat>
#include <chrono>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include <vpp/vpp.hh>
#include <vpp/algorithms/fast_detector/fast.hh>
#include <vpp/algorithms/optical_flow.hh>

#include "../utils/kitti.hh"

using namespace vpp;

struct stats {  // KITTI.cc:21-42
  float min_ = FLT_MAX, max_ = -FLT_MAX, cpt_ = 0.f, sum_ = 0.f;
  void take(float f) { min_ = std::min(min_, f); max_ = std::max(max_, f); cpt_++; sum_ += f; }
  float avg() const { return cpt_ ? sum_ / cpt_ : 0.f; }
};

struct parameters { int nscales = 1, winsize = 9, propagation = 2, min_scale = 0, patchsize = 5, detector_th = 10, block_size = 10; };

static parameters read_parameters(const char* path) {
  parameters p;
  const std::map<std::string, int*> fields = {{"nscales", &p.nscales}, {"winsize", &p.winsize}, {"propagation", &p.propagation}, {"min_scale", &p.min_scale},
                                              {"patchsize", &p.patchsize}, {"detector_th", &p.detector_th}, {"block_size", &p.block_size}};
  std::ifstream f(path);
  if (!f) throw std::runtime_error(std::string("Cannot read parameters ") + path);
  for (std::string line; std::getline(f, line);) {
    line = line.substr(0, line.find('#'));
    for (char& ch : line) if (ch == ':' || ch == '=') ch = ' ';
    std::istringstream ls(line);
    std::string name; double value;
    if (!(ls >> name >> value)) continue;
    auto it = fields.find(name);
    if (it == fields.end()) throw std::runtime_error("Unknown parameter " + name);
    *it->second = int(value);
  }
  return p;
}

int main(int argc, const char* argv[]) {
  if (argc != 5) { std::cerr << "Usage: " << argv[0] << " kitti_root n_images config_file result_file" << std::endl; return 1; }
  try {
    const parameters params = read_parameters(argv[3]);
    const int nframes = std::atoi(argv[2]);
    stats runtime_stats, error_stats, nkeypoints_stats, density_stats, epe_stats;
    kitti::foreach_training_pair(argv[1], nframes, [&](const image2d<vuchar3>& frame1, const image2d<vuchar3>& frame2, const image2d<vfloat3>& ref_flow) {
      image2d<uint8_t> i1_gl = rgb_to_graylevel<uint8_t>(frame1), i2_gl = rgb_to_graylevel<uint8_t>(frame2);
      i1_gl = clone(i1_gl, _border = params.winsize, _aligned = 128);  // KITTI.cc:137-138
      i2_gl = clone(i2_gl, _border = params.winsize, _aligned = 128);
      auto keypoints = fast9(i1_gl, params.detector_th, _blockwise, _block_size = params.block_size);
      image2d<vfloat3> flow(frame1.domain());
      fill(flow, vfloat3(0, 0, 0));
      const auto t0 = std::chrono::steady_clock::now();  // the timed region is the flow call, like the reference's iod::timer (KITTI.cc:148-182)
      semi_dense_optical_flow(keypoints,
                              [&](int i, vint2 pos, int) {
                                const vint2 d = pos - keypoints[i];
                                flow(keypoints[i]) = vfloat3(float(d[0]), float(d[1]), 1.f);
                              },
                              i1_gl, i2_gl, _winsize = params.winsize, _propagation = params.propagation, _nscales = params.nscales, _patchsize = params.patchsize,
                              _min_scale = params.min_scale);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (!ref_flow.has_data()) throw std::runtime_error("Cannot read the reference flow of a pair");
      const auto flow_errors = kitti::flow_error_stats(flow, ref_flow);
      runtime_stats.take(float(us));
      error_stats.take(flow_errors.n3);
      nkeypoints_stats.take(float(keypoints.size()));
      density_stats.take(flow_errors.density);
      epe_stats.take(flow_errors.avg);
    });
    std::ofstream out(argv[4]);
    out << "runtime: " << runtime_stats.avg() << "\nerrors: " << error_stats.avg() << "\nnkeypoints: " << nkeypoints_stats.avg()
        << "\n# extras\nmean_endpoint_error: " << epe_stats.avg() << "\ndensity: " << density_stats.avg() << "\nruntime_min: " << runtime_stats.min_ << "\nruntime_max: " << runtime_stats.max_
        << "\n# parameters\nnscales: " << params.nscales << "\nwinsize: " << params.winsize << "\npropagation: " << params.propagation << "\nmin_scale: " << params.min_scale
        << "\npatchsize: " << params.patchsize << "\ndetector_th: " << params.detector_th << "\nblock_size: " << params.block_size << "\n";
    return out ? 0 : 2;
  } catch (const std::exception& e) { std::cerr << "KITTI: " << e.what() << std::endl; return 3; }
}
