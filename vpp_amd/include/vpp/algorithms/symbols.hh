// algorithm option names (reference: vpp/algorithms/symbols.hh; fast.hpp:935-943, lucas_kanade.hpp:139-149,
// semi_dense_optical_flow.hpp:56-66, video_extruder.hpp:35-41).
#pragma once
#include <vpp/core/symbols.hh>

VPP_DEFINE_SYMBOL(mask)
VPP_DEFINE_SYMBOL(scores)
VPP_DEFINE_SYMBOL(local_maxima)
VPP_DEFINE_SYMBOL(blockwise)
VPP_DEFINE_SYMBOL(max_points_per_block)
// accepted for source compatibility; the epipolar variants of semi_dense_optical_flow are not implemented
VPP_DEFINE_SYMBOL(fundamental_matrix)
VPP_DEFINE_SYMBOL(epipolar_flow)
VPP_DEFINE_SYMBOL(epipolar_filter)
VPP_DEFINE_SYMBOL(keypoints)
VPP_DEFINE_SYMBOL(niterations)
VPP_DEFINE_SYMBOL(winsize)
VPP_DEFINE_SYMBOL(nscales)
VPP_DEFINE_SYMBOL(min_ev)
VPP_DEFINE_SYMBOL(delta)
VPP_DEFINE_SYMBOL(prediction)
VPP_DEFINE_SYMBOL(flow)
VPP_DEFINE_SYMBOL(min_scale)
VPP_DEFINE_SYMBOL(propagation)
VPP_DEFINE_SYMBOL(patchsize)
VPP_DEFINE_SYMBOL(detector_th)
VPP_DEFINE_SYMBOL(keypoint_spacing)
VPP_DEFINE_SYMBOL(detector_period)
VPP_DEFINE_SYMBOL(max_trajectory_length)
VPP_DEFINE_SYMBOL(fast9_reference_ring)   // extension: sample the ring as fast_detector9_simd does (default) ...
VPP_DEFINE_SYMBOL(fast9_corrected_ring)   // ... or the true Bresenham ring
