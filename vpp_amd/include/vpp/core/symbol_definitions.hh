// symbol_definitions.hh — include-path compatibility (reference: vpp/core/symbol_definitions.hh): the option symbols are in symbols.hh.
#pragma once
#include <vpp/core/symbols.hh>
