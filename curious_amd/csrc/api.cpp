// Error reporting and device probing for libcurious_hip.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void curious_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* curious_last_error(void) { return g_err; }

extern "C" int curious_abi_version(void) { return CURIOUS_ABI_VERSION; }

extern "C" int curious_device_info(char* name_host, int name_len, int* cu_count_host) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  CURIOUS_CHECK(e == hipSuccess, "curious_device_info: no HIP device: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  CURIOUS_CHECK(e == hipSuccess, "curious_device_info: hipGetDeviceProperties: %s", hipGetErrorString(e));
  CURIOUS_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                "curious_device_info: libcurious_hip is built for gfx950 (MI355X), found %s", prop.gcnArchName);
  if (name_host && name_len > 0) {
    strncpy(name_host, prop.name, name_len - 1);
    name_host[name_len - 1] = 0;
  }
  if (cu_count_host) *cu_count_host = prop.multiProcessorCount;
  return 0;
}
