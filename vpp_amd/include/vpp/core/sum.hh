#pragma once
#include <vpp/core/copy.hh>
