// BLR/BLRMatrix.hpp: the include path the reference's callers use (`#include "BLR/BLRMatrix.hpp"`, /root/reference/src/BLR/BLRMatrix.hpp);
// the declarations live with the host engine.  Compile with -I<repo>/include.
#pragma once
#include "../../strumpack_amd/csrc/host/BLRMatrix.hpp"
