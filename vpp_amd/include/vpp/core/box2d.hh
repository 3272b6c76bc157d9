// box2d.hh — include-path compatibility (reference: vpp/core/box2d.hh): box2d and make_box2d live in boxNd.hh here.
#pragma once
#include <vpp/core/boxNd.hh>
