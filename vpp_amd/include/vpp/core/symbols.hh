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

namespace vpp { using namespace s; }
