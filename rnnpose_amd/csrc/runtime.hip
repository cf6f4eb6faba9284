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

// fp16x3 range guard (rnnpose_f16x3_saturation_check): one 8-byte device counter per device, allocated when the check is
// switched on (never inside a launch path: hipMalloc is illegal under stream capture)
static bool g_sat_on = false;
static unsigned long long* g_sat[64] = {};

unsigned long long* sat_counter() {
  if (!g_sat_on) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return g_sat[dev];
}

}  // namespace rp

namespace rp {
int cu_count() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}
}  // namespace rp

extern "C" {

int rnnpose_f16x3_saturation_check(int enable) {
  if (enable) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return rp::fail_arg("rnnpose_f16x3_saturation_check", "no current device");
    if (!rp::g_sat[dev]) {
      if (hipMalloc(reinterpret_cast<void**>(&rp::g_sat[dev]), sizeof(unsigned long long)) != hipSuccess ||
          hipMemset(rp::g_sat[dev], 0, sizeof(unsigned long long)) != hipSuccess) {
        rp::g_sat[dev] = nullptr;
        rp::set_error("rnnpose_f16x3_saturation_check: cannot allocate the device counter");
        return 2;
      }
    }
  }
  rp::g_sat_on = enable != 0;
  return 0;
}

int rnnpose_f16x3_saturation_count(unsigned long long* h_count, int reset, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_f16x3_saturation_count";
  RP_REQUIRE(h_count, fn, "null pointer");
  int dev = 0;
  RP_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, fn, "no current device");
  *h_count = 0;
  if (!rp::g_sat[dev]) return 0;
  hipStream_t st = rp::as_stream(stream);
  hipError_t e = hipMemcpyAsync(h_count, rp::g_sat[dev], sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && reset) e = hipMemsetAsync(rp::g_sat[dev], 0, sizeof(unsigned long long), st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    rp::set_error("%s: HIP error %d (%s)", fn, static_cast<int>(e), hipGetErrorString(e));
    return 2;
  }
  return 0;
}

int rnnpose_f16x3_saturation_peek(unsigned long long* d_count, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_f16x3_saturation_peek";
  RP_REQUIRE(d_count, fn, "null pointer");
  int dev = 0;
  RP_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, fn, "no current device");
  hipStream_t st = rp::as_stream(stream);
  hipError_t e = rp::g_sat[dev] ? hipMemcpyAsync(d_count, rp::g_sat[dev], sizeof(unsigned long long), hipMemcpyDeviceToDevice, st)
                                : hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess) {
    rp::set_error("%s: HIP error %d (%s)", fn, static_cast<int>(e), hipGetErrorString(e));
    return 2;
  }
  return 0;
}

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
