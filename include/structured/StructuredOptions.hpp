// structured/StructuredOptions.hpp: the include path the reference's callers use (`#include "structured/StructuredOptions.hpp"`, /root/reference/src/structured/StructuredOptions.hpp);
// the declarations live with the host engine.  Compile with -I<repo>/include.
#pragma once
#include "../../strumpack_amd/csrc/host/StructuredOptions.hpp"
