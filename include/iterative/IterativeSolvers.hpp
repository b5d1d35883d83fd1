// forwarder: see strumpack_amd/csrc/host/IterativeSolvers.hpp
#pragma once
#include "../../strumpack_amd/csrc/host/IterativeSolvers.hpp"
