// vpp.hh — umbrella header of the vpp-shaped C++ surface (reference: vpp/vpp.hh).
#pragma once
#include <iostream>   // the reference's umbrella header brings it in (its tests print without including it)
#include <unistd.h>   // likewise (through Eigen / OpenMP there): benchmarks/image_iterations.cc:90 calls ::getpid() with no include of its own
#include <vpp/core/vector.hh>
#include <vpp/core/boxNd.hh>
#include <vpp/core/imageNd.hh>
#include <vpp/core/image2d.hh>
#include <vpp/core/image3d.hh>
#include <vpp/core/window.hh>
#include <vpp/core/relative_accessor.hh>
#include <vpp/core/tuple_utils.hh>
#include <vpp/core/pixel_wise.hh>
#include <vpp/core/block_wise.hh>
#include <vpp/core/copy.hh>
#include <vpp/core/clone.hh>
#include <vpp/core/fill.hh>
#include <vpp/core/sum.hh>
#include <vpp/core/colorspace_conversions.hh>
#include <vpp/core/keypoint_container.hh>
#include <vpp/core/keypoint_trajectory.hh>
#include <vpp/core/pyramid.hh>
