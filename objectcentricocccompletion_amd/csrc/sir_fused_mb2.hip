// csrc/sir_fused_impl.hpp for 32-row tiles
#define OCOCC_SIR_MB 2
#include "sir_fused_impl.hpp"
