// Library-level entry points: version, error text, device probe.
#include "sdm_common.h"

static thread_local char g_err[512] = "";

int sdm_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" int sdm_version(void) { return 100; }

extern "C" const char* sdm_last_error(void) { return g_err; }

extern "C" int sdm_device_info(int* out4) {
  SDM_CHECK_ARG(out4 != nullptr, "device_info: null");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "device_info: no HIP device");
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return sdm_set_error(SDM_ERR_LAUNCH, "device_info: props failed");
  out4[0] = p.multiProcessorCount;
  out4[1] = (int)p.maxSharedMemoryPerMultiProcessor;
  out4[2] = p.warpSize;
  out4[3] = strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
  return SDM_OK;
}

// sizes of the ABI structs, so a binding can assert its own layout (0: igemm desc, 1: wgrad desc, 2: rows,
// 3: linear job, 4: adam job)
extern "C" size_t sdm_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(sdm_igemm_desc);
    case 1: return sizeof(sdm_wgrad_desc);
    case 2: return sizeof(sdm_rows);
    case 3: return sizeof(sdm_linear_job);
    case 4: return sizeof(sdm_adam_job);
    default: return 0;
  }
}
