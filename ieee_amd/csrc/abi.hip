// Error state + library-level entry points of the C ABI (include/ieee_amd.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace ieee {
static thread_local char g_err[512] = "";

void set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  int n = snprintf(g_err, sizeof(g_err), "[ieee_amd %d] ", code);
  vsnprintf(g_err + n, sizeof(g_err) - n, fmt, ap);
  va_end(ap);
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(IEEE_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return IEEE_ERR_HIP;
  }
  return IEEE_OK;
}
}  // namespace ieee

extern "C" {

const char* ieee_last_error(void) { return ieee::g_err; }

int ieee_version(void) { return 1; }

int ieee_device_is_gfx950(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

}  // extern "C"
