// Just enough of OpenCV's names for vpp/utils/opencv_bridge.hh to parse (it is dragged in by vpp/algorithms/optical_flow.hh
// through dense_optical_flow.hpp:2).  Nothing here is ever executed by oracle/ref.
#pragma once
#include <cstdlib>
namespace cv {
struct UMatData { int refcount; };
struct Mat {
  unsigned char* data = nullptr; int rows = 0, cols = 0; size_t step = 0; UMatData* u = nullptr; int* refcount = nullptr;
  Mat() {}
  Mat(int r, int c, int, void* d = nullptr, size_t s = 0) : data((unsigned char*)d), rows(r), cols(c), step(s) {}
  void addref() {}
};
inline void fastFree(void* p) { std::free(p); }
}  // namespace cv
#define CV_8UC1 0
#define CV_8UC2 8
#define CV_8UC3 16
#define CV_8UC4 24
#define CV_8SC1 1
#define CV_8SC2 9
#define CV_8SC3 17
#define CV_8SC4 25
#define CV_16UC1 2
#define CV_16UC2 10
#define CV_16UC3 18
#define CV_16UC4 26
#define CV_16SC1 3
#define CV_16SC2 11
#define CV_16SC3 19
#define CV_16SC4 27
#define CV_32SC1 4
#define CV_32SC2 12
#define CV_32SC3 20
#define CV_32SC4 28
#define CV_32FC1 5
#define CV_32FC2 13
#define CV_32FC3 21
#define CV_32FC4 29
