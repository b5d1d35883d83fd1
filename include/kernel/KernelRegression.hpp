// kernel/KernelRegression.hpp: the include path the reference's callers use (/root/reference/src/kernel/KernelRegression.hpp:
// Kernel::fit_HSS / predict); here both live with the kernel classes.
#pragma once
#include "../../strumpack_amd/csrc/host/Kernel.hpp"
#include "../../strumpack_amd/csrc/host/HSSMatrix.hpp"
