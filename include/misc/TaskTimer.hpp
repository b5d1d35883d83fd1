// misc/TaskTimer.hpp: the include path the reference's callers use (/root/reference/src/misc/TaskTimer.hpp).
#pragma once
#include "../../strumpack_amd/csrc/host/TaskTimer.hpp"
