// Library runtime: version string and thread-local error plumbing of the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace hoisdf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return HOISDF_ERR_LAUNCH;
  }
  return HOISDF_OK;
}

}  // namespace hoisdf

extern "C" const char* hoisdf_version(void) { return "hoisdf-hip 0.1 (gfx950)"; }
extern "C" const char* hoisdf_last_error(void) { return hoisdf::g_err; }
