// zero.hh — include-path compatibility (reference: vpp/core/zero.hh): zero<V> is defined with the vector types.
#pragma once
#include <vpp/core/vector.hh>
