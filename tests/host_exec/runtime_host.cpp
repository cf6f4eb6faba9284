// TEST INFRASTRUCTURE: csrc/runtime.hip (error plumbing, range-guard counter) compiled for the host, over host versions of the few
// HIP runtime calls the library's host code makes -- "device memory" is host memory here.
#include "harness.hpp"

#include <cstdlib>

extern "C" {
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "host execution"; }
hipError_t hipGetDevice(int* dev) { *dev = 0; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }      // (a 4-CU "device": persistent strip launches of 8 / 16 workgroups)
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof *p);
  std::strcpy(p->gcnArchName, "host");
  return hipSuccess;
}
}

#include "runtime.hip"

// HOSTEXEC_BACKTRACE=1: native backtrace on SIGSEGV (on an alternate stack: a fiber that ran out of stack cannot run a handler)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
void segv_handler(int) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  _exit(139);
}
struct InstallHandler {
  InstallHandler() {
    if (!std::getenv("HOSTEXEC_BACKTRACE")) return;
    static char alt[1 << 16];
    stack_t ss = {};
    ss.ss_sp = alt;
    ss.ss_size = sizeof alt;
    sigaltstack(&ss, nullptr);
    struct sigaction sa = {};
    sa.sa_handler = segv_handler;
    sa.sa_flags = SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
  }
} install_handler;
}  // namespace
