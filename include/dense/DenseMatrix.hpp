// dense/DenseMatrix.hpp: the include path the reference's callers use (`#include "dense/DenseMatrix.hpp"`, /root/reference/src/dense/DenseMatrix.hpp);
// the declarations live with the host engine.  Compile with -I<repo>/include.
#pragma once
#include "../../strumpack_amd/csrc/host/DenseMatrix.hpp"
