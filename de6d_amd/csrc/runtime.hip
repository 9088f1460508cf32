// runtime.hip — library identification and error bookkeeping for libdet6d_hip.so.
#include "common.h"
#include <stdio.h>

namespace {
thread_local char g_err[256] = "";
}

void det6d_set_error(const char *what, hipError_t err) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
}

DET6D_API const char *det6d_version(void) { return "det6d-hip gfx950 abi6"; }
DET6D_API const char *det6d_last_error(void) { return g_err; }
