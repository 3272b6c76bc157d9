// kitti_io_test.cc — host-only check of the evaluation harness' file I/O and statistics (evaluation/utils/png.hh, kitti.hh).
// usage: kitti_io_test dir      dir holds PNGs written by tests/test_kitti_harness.py; prints one line per check for it to verify.
#include <cstdio>
#include <string>

#include "../../evaluation/utils/kitti.hh"

using namespace vpp;

int main(int argc, char** argv) {
  if (argc != 2) return 2;
  const std::string dir = argv[1];
  for (const char* name : {"gray8", "rgb8", "rgb16", "gray16", "rgba8"}) {
    const minipng::image im = minipng::read(dir + "/" + name + ".png");
    if (!im.ok()) { std::printf("%s unreadable\n", name); continue; }
    unsigned long crc = 0;
    for (int r = 0; r < im.height; r++) for (int c = 0; c < im.width; c++) for (int k = 0; k < im.channels; k++) {
      const unsigned v = im.at(r, c, k); const unsigned char le[2] = {(unsigned char)(v & 255), (unsigned char)(v >> 8)};
      crc = crc32(crc, le, 2);
    }
    std::printf("%s %d %d %d %d %lu\n", name, im.width, im.height, im.channels, im.depth, crc);
  }
  std::printf("missing %d\n", minipng::read(dir + "/does_not_exist.png").ok() ? 1 : 0);
  const image2d<vuchar3> g = kitti::load_image(dir + "/gray8.png");
  std::printf("gray_as_rgb %d\n", g.has_data() && g(1, 2)[0] == g(1, 2)[1] && g(1, 2)[1] == g(1, 2)[2] ? 1 : 0);
  // flow round trip: decode, re-encode
  const image2d<vfloat3> fin = kitti::load_flow(dir + "/flow_in.png");
  image2d<vfloat2> f2(fin.domain()); image2d<char> has(fin.domain());
  for (auto p : fin.domain()) { f2(p) = fin(p).segment<2>(0); has(p) = fin(p)[2] > 0.f; }
  kitti::write_flow(dir + "/flow_out.png", f2, has);
  std::printf("flow_sample %.6f %.6f %.1f\n", fin(3, 5)[0], fin(3, 5)[1], fin(3, 5)[2]);
  // statistics
  const auto st = kitti::flow_error_stats(kitti::load_flow(dir + "/est.png"), kitti::load_flow(dir + "/ref.png"));
  std::printf("stats %.6f %.6f %.6f %.6f %.6f %.6f %zu %d\n", st.n1, st.n3, st.n5, st.n10, st.avg, st.density, st.errors.size(), int(st.errors_map(2, 2)));
  return 0;
}
