// lambda_call_bench.cc — the reference's LITERAL call form on 4K frames, as a single-source gfx950 program over the drop-in headers (hipcc -x hip -DVPP_AMD_DEVICE
// -DVPP_AMD_HIPCC): the opaque 5 x 5 mean lambdas of benchmarks/box_5x5_filter2.cc:71-81 (`int`) and examples/box_filter.cc:23-32 (`vuchar3`), unmodified, are the
// body of the generic kernels of vpp/core/pixel_wise_device.hh.  Beside each: the same body under `_nbh_read_only` (register window for 4-byte pixels, LDS tile for
// byte-sized ones) and the tagged functor ops::box_mean<5, 5>() (the hand-written kernel) launched per call (`_immediate`).  One launch per call over NS rotating
// frame sets (nothing survives in the 256 MiB Infinity Cache between two uses); at most two calls are queued (vpp/core/device.hh).  Prints one JSON line; bench.py
// reports it as roofline.lambda_call.  Parity of these kernels against the oracle is tests/cpp/device_lambda_test.cc's business; here every form is only compared with
// the tagged functor's result.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include <vpp/vpp.hh>

using namespace vpp;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)
static std::mt19937 rng(11);
static double seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- benchmarks/box_5x5_filter2.cc:71-81, verbatim ----
void vpp_pixel_wise(image2d<int> B, image2d<int> A)
{
  vpp::pixel_wise(B, relative_access(A)) | [&] (int& b, auto a)
  {
    int sum = 0;
    for (int i = -2; i <= 2; i++)
    for (int j = -2; j <= 2; j++)
      sum += a(i, j);
    b = sum / 25;
  };
}

template <class V> static bool same_pixels(const image2d<V>& a, const image2d<V>& b) {
  for (int r = 0; r < a.nrows(); r++)
    if (std::memcmp(&a(r, 0), &b(r, 0), size_t(a.ncols()) * sizeof(V))) return false;
  return true;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? std::atoi(argv[1]) : 200, NS = 10, NR = 2160, NC = 3840;
  double t0;
  double r_int[3], r_u8[3];
  {
    std::vector<image2d<int>> S, D;
    for (int q = 0; q < NS; q++) { S.emplace_back(NR, NC, _border = 2); D.emplace_back(S[q].domain()); }
    image2d<int> T(S[0].domain());
    for (auto p : S[0].domain_with_border()) S[0](p) = int(rng() % 1000);
    for (int q = 1; q < NS; q++) for (auto p : S[0].domain_with_border()) S[q](p) = S[0](p);
    auto body = [] (int& b, auto a) {
      int sum = 0;
      for (int i = -2; i <= 2; i++)
      for (int j = -2; j <= 2; j++)
        sum += a(i, j);
      b = sum / 25;
    };
    for (int q = 0; q < NS; q++) vpp_pixel_wise(D[q], S[q]);
    pixel_wise(T, relative_access(S[0])) | ops::box_mean<5, 5>();
    CHECK(same_pixels(D[0], T) && same_pixels(D[NS - 1], T));
    vpp::device::sync();
    t0 = seconds();
    for (int k = 0; k < K; k++) vpp_pixel_wise(D[k % NS], S[k % NS]);
    vpp::device::sync();
    r_int[0] = (seconds() - t0) / K;
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS]))(_nbh_read_only) | body;
    vpp::device::sync();
    r_int[1] = (seconds() - t0) / K;
    CHECK(same_pixels(D[1], T));
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS]))(_immediate) | ops::box_mean<5, 5>();
    vpp::device::sync();
    r_int[2] = (seconds() - t0) / K;
  }
  {
    std::vector<image2d<vuchar3>> S, D;
    for (int q = 0; q < NS; q++) { S.emplace_back(NR, NC, _border = 2); D.emplace_back(S[q].domain()); }
    image2d<vuchar3> T(S[0].domain());
    for (auto p : S[0].domain_with_border()) S[0](p) = vuchar3(rng() & 255, rng() & 255, rng() & 255);
    for (int q = 1; q < NS; q++) for (auto p : S[0].domain_with_border()) S[q](p) = S[0](p);
    auto k3 = [] (vuchar3& out, auto nbh) {   // examples/box_filter.cc:23-32
      vint3 sum = vint3::Zero();
      for (int i = -2; i <= 2; i++) for (int j = -2; j <= 2; j++) sum += nbh(i, j).template cast<int>();
      out = (sum / 25).template cast<unsigned char>();
    };
    for (int q = 0; q < NS; q++) pixel_wise(D[q], relative_access(S[q])) | k3;
    pixel_wise(T, relative_access(S[0])) | ops::box_mean<5, 5>();
    CHECK(same_pixels(D[0], T) && same_pixels(D[NS - 1], T));
    vpp::device::sync();
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS])) | k3;
    vpp::device::sync();
    r_u8[0] = (seconds() - t0) / K;
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS]))(_nbh_read_only) | k3;
    vpp::device::sync();
    r_u8[1] = (seconds() - t0) / K;
    CHECK(same_pixels(D[1], T));
    t0 = seconds();
    for (int k = 0; k < K; k++) pixel_wise(D[k % NS], relative_access(S[k % NS]))(_immediate) | ops::box_mean<5, 5>();
    vpp::device::sync();
    r_u8[2] = (seconds() - t0) / K;
  }
  const double px = double(NR) * NC, peak = 8000e9;
  auto frac = [&](double bytes_per_px, double s) { return bytes_per_px * px / s / peak; };
  std::printf("{\"int_5x5\": {\"literal_us\": %.2f, \"literal_frac\": %.3f, \"nbh_read_only_us\": %.2f, \"nbh_read_only_frac\": %.3f, \"ops_box_mean_us\": %.2f, \"ops_box_mean_frac\": %.3f}, "
              "\"vuchar3_5x5\": {\"literal_us\": %.2f, \"literal_frac\": %.3f, \"nbh_read_only_us\": %.2f, \"nbh_read_only_frac\": %.3f, \"ops_box_mean_us\": %.2f, \"ops_box_mean_frac\": %.3f}, "
              "\"calls\": %d, \"frame_sets\": %d, \"form\": \"the reference's opaque 5x5 mean lambdas (benchmarks/box_5x5_filter2.cc:71-81 on int, examples/box_filter.cc:23-32 on vuchar3) compiled "
              "single-source, one launch per 4K frame over rotating frame sets, host cost included; frac = 2 sizeof(V) B/px / time / 8 TB/s\"}\n",
              r_int[0] * 1e6, frac(8, r_int[0]), r_int[1] * 1e6, frac(8, r_int[1]), r_int[2] * 1e6, frac(8, r_int[2]),
              r_u8[0] * 1e6, frac(6, r_u8[0]), r_u8[1] * 1e6, frac(6, r_u8[1]), r_u8[2] * 1e6, frac(6, r_u8[2]), K, NS);
  return 0;
}
