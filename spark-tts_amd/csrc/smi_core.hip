// smi_core.hip -- version, error reporting and device check of libsparkmi.
#include "smi_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[1024] = "";

void smi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int smi_version(void) { return SMI_ABI_VERSION; }

const char* smi_last_error(void) { return g_err; }

int smi_device_check(char* name, int n) {
  int dev = 0;
  SMI_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  SMI_HIP(hipGetDeviceProperties(&prop, dev));
  if (name && n > 0) {
    strncpy(name, prop.gcnArchName, (size_t)n - 1);
    name[n - 1] = 0;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    smi_set_error("libsparkmi is built for gfx950 only; device %d is %s", dev, prop.gcnArchName);
    return SMI_EINVAL;
  }
  return SMI_OK;
}

}  // extern "C"
