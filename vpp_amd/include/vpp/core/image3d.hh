#pragma once
#include <vpp/core/imageNd.hh>
namespace vpp { template <class V> using image3d = imageNd<V, 3>; }
