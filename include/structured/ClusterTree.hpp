// structured/ClusterTree.hpp: the include path the reference's callers use (`#include "structured/ClusterTree.hpp"`, /root/reference/src/structured/ClusterTree.hpp);
// the declarations live with the host engine.  Compile with -I<repo>/include.
#pragma once
#include "../../strumpack_amd/csrc/host/ClusterTree.hpp"
