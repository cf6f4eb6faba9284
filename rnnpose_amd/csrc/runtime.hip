// Error plumbing + device info for librnnpose_hip.so.
#include "common.hpp"

#include <cstring>

namespace rp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail_arg(const char* fn, const char* what) {
  set_error("%s: invalid argument: %s", fn, what);
  return 1;
}

int check_launch(const char* fn) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP error %d (%s)", fn, static_cast<int>(e), hipGetErrorString(e));
    return 2;
  }
  return 0;
}

}  // namespace rp

extern "C" {

int rnnpose_abi_version(void) { return RNNPOSE_ABI_VERSION; }

const char* rnnpose_last_error(void) { return rp::g_err; }

int rnnpose_device_info(int dev, char* h_name, int name_len, int* h_cus) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) {
    rp::set_error("rnnpose_device_info: HIP error %d (%s)", static_cast<int>(e), hipGetErrorString(e));
    return 2;
  }
  if (h_name && name_len > 0) {
    strncpy(h_name, p.gcnArchName, static_cast<size_t>(name_len) - 1);
    h_name[name_len - 1] = 0;
  }
  if (h_cus) *h_cus = p.multiProcessorCount;
  return 0;
}

}  // extern "C"
