// Library-level entry points of the C ABI (include/ococc_hip.h).
#include "common.hpp"

thread_local char ococc_err_buf[512] = {0};

extern "C" const char* ococc_last_error(void) { return ococc_err_buf; }
extern "C" int ococc_version(void) { return 100; /* round 1 */ }
extern "C" const char* ococc_arch(void) { return "gfx950"; }
