// HSS/HSSMatrix.hpp: the include path the reference's callers use (`#include "HSS/HSSMatrix.hpp"`, /root/reference/src/HSS/HSSMatrix.hpp);
// the declarations live with the host engine.  Compile with -I<repo>/include.
#pragma once
#include "../../strumpack_amd/csrc/host/HSSMatrix.hpp"
