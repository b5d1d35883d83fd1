// structured/StructuredMatrix.hpp: the include path the reference's callers use (`#include "structured/StructuredMatrix.hpp"`, /root/reference/src/structured/StructuredMatrix.hpp);
// the declarations live with the host engine.  Compile with -I<repo>/include.
#pragma once
#include "../../strumpack_amd/csrc/host/StructuredMatrix.hpp"
