// cast_to_float.hh — include-path compatibility (reference: vpp/core/cast_to_float.hh): cast_to_float<V> is defined with the vector types.
#pragma once
#include <vpp/core/vector.hh>
