// HSS/HSSOptions.hpp: the include path the reference's callers use (`#include "HSS/HSSOptions.hpp"`, /root/reference/src/HSS/HSSOptions.hpp);
// the declarations live with the host engine.  Compile with -I<repo>/include.
#pragma once
#include "../../strumpack_amd/csrc/host/HSSOptions.hpp"
