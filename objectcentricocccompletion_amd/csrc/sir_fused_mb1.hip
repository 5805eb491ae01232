// csrc/sir_fused_impl.hpp for 16-row tiles
#define OCOCC_SIR_MB 1
#include "sir_fused_impl.hpp"
