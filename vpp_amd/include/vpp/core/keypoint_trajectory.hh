#pragma once
#include <vpp/core/keypoint_container.hh>
