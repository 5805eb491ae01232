// csrc/sir_fused_impl.hpp for 64-row tiles
#define OCOCC_SIR_MB 4
#include "sir_fused_impl.hpp"
