// imageNd_iterator.hh — include-path compatibility (reference: vpp/core/imageNd_iterator.hh): image iteration is part of imageNd.hh here.
#pragma once
#include <vpp/core/imageNd.hh>
