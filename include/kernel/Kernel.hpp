// kernel/Kernel.hpp: the include path the reference's callers use (`#include "kernel/Kernel.hpp"`, /root/reference/src/kernel/Kernel.hpp);
// the declarations live with the host engine.  Compile with -I<repo>/include.
#pragma once
#include "../../strumpack_amd/csrc/host/Kernel.hpp"
