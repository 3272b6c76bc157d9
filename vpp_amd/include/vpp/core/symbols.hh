// symbols.hh — option names of the core (reference: vpp/core/symbols.hh:8-18, vpp/core/pixel_wise.hh:27-33).
#pragma once
#include <vpp/core/options.hh>

VPP_DEFINE_SYMBOL(border)
VPP_DEFINE_SYMBOL(aligned)
VPP_DEFINE_SYMBOL(data)
VPP_DEFINE_SYMBOL(pitch)
VPP_DEFINE_SYMBOL(no_threads)
VPP_DEFINE_SYMBOL(left_to_right)
VPP_DEFINE_SYMBOL(right_to_left)
VPP_DEFINE_SYMBOL(top_to_bottom)
VPP_DEFINE_SYMBOL(bottom_to_top)
VPP_DEFINE_SYMBOL(block_size)
VPP_DEFINE_SYMBOL(mem_forward)
VPP_DEFINE_SYMBOL(mem_backward)
VPP_DEFINE_SYMBOL(tie_arguments)
// not in the reference: where an opaque pixel_wise callable runs in a TU compiled by hipcc (vpp/core/pixel_wise_device.hh)
VPP_DEFINE_SYMBOL(host)
VPP_DEFINE_SYMBOL(device)
// not in the reference: pixel_wise(...)(_nbh_read_only) vouches that the callable only READS through its relative_access / box_nbh2d range and that no tap
// reaches further than 4 pixels from the centre — the device engine may then serve the taps from an LDS tile (pixel_wise_device.hh: pixel_wise_tile_kernel).
// Without it a neighbourhood is a reference into the image, as in the reference (relative_accessor.hh:26-33 returns V&; distance_transforms.hh writes through it).
VPP_DEFINE_SYMBOL(nbh_read_only)
// not in the reference: pixel_wise(...)(_immediate) | ops::add() / ops::box_mean<R, C>() launches this call's kernel at once (the reference's "the call has
// happened when operator| returns", as far as a stream can say it) instead of holding the frame back for a batched launch (vpp/core/device.hh)
VPP_DEFINE_SYMBOL(immediate)

namespace vpp { using namespace s; }
