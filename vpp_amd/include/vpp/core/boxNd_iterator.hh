// boxNd_iterator.hh — include-path compatibility (reference: vpp/core/boxNd_iterator.hh): the iterator is defined with boxNd.
#pragma once
#include <vpp/core/boxNd.hh>
