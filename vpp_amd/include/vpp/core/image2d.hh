#pragma once
#include <vpp/core/imageNd.hh>
namespace vpp { template <class V> using image2d = imageNd<V, 2>; }
