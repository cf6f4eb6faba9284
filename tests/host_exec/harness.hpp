// TEST INFRASTRUCTURE (never linked into librnnpose_hip.so): runs the library's HIP kernels ON THE HOST, lane by lane.
//
// A .hip source of rnnpose_amd/csrc is compiled as PLAIN C++ (hip's host_defines.h leaves __device__ / __host__ empty outside a HIP
// compile; this header supplies __global__, __shared__, threadIdx / blockIdx, __syncthreads, the wave-level operations and
// hipLaunchKernelGGL).  A launch runs every workgroup as a set of FIBERS (ucontext), one per GPU thread, scheduled round-robin by one
// OS thread: deterministic, no data races of the emulation's own.  `__shared__` arrays become thread_local statics (one copy per OS
// thread = per workgroup in flight), `__syncthreads()` and the wave collectives are yield points that release when every LIVE thread of
// the workgroup / wave has arrived (a thread that returned no longer takes part, as on the GPU).  Wave collectives: shuffles, ballot,
// and the MFMA instructions the kernels issue -- each lane deposits its operand fragment, the wave synchronises, each lane computes
// the elements of D it owns in the gfx950 register layout (MI355X_MICROARCH.md / cdna_hip_programming.md).
// `__builtin_amdgcn_wave_barrier()` -- a compiler-only fence on the GPU, where a wave's LDS operations execute in order -- is a real
// wave synchronisation here, because lanes of a wave do not run in lock step.
//
// The C-ABI entry points of the source therefore run unchanged -- argument checks, launch geometry, kernel code -- on host pointers,
// and tests/test_kernels_on_host.py compares what they compute with the CPU oracle and the reference-generated golden vectors in the
// `-m "not gpu"` tier.  gfx950 inline assembly (the LDS-DMA requests and counted waits of the strip convolution kernels, two waits in
// lm.hip, one register constraint) is rewritten in a scratch copy by build_host.py, as are the four places where an epilogue relies on
// a wave's LDS operations executing in order (lanes of a wave are independent fibers between collectives).  Host libm replaces the device's expf / tanhf (1-ulp differences); the MFMA's internal summation order is not modelled
// (the products of one instruction are summed exactly and rounded to fp32 once).
#pragma once
#include <hip/hip_runtime.h>
#include <setjmp.h>
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define amdgpu_num_vgpr(...)                 // (inside __attribute__((...)) of a kernel: kernel-only attributes are dropped)
#define amdgpu_waves_per_eu(...)
#define amdgpu_flat_work_group_size(...)
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)

struct host_idx3 {
  unsigned x, y, z;
};
// ONE instance per process (inline variables): inline functions of the sources' shared headers are merged across translation units
inline thread_local host_idx3 threadIdx, blockIdx;
inline host_idx3 blockDim, gridDim;

namespace hostexec {

constexpr int kWaveSize = 64;
constexpr size_t kStack = 256 * 1024;

struct PendingVm {               // one lane's 16 bytes of an LDS-DMA request that has been issued but not waited for (d == nullptr: a
  unsigned char* d;              // register load / other vector-memory request that only takes a place in the in-order queue)
  const unsigned char* s;
};
struct Wave {
  int live = 0, arrived = 0;
  unsigned gen = 0;
  alignas(64) unsigned char buf[kWaveSize][256];       // one operand fragment per lane
  struct VmQueue {                                      // r06: per lane, in issue order (the same length in every lane: requests are wave collectives)
    std::vector<PendingVm> v;
    size_t head = 0;
    size_t size() const { return v.size() - head; }
    void push_back(const PendingVm& p) { v.push_back(p); }
    const PendingVm& front() const { return v[head]; }
    void pop_front() { if (++head == v.size()) { v.clear(); head = 0; } }
  } vmq[kWaveSize];
};

struct Fiber {
  ucontext_t ctx;               // first entry only (makecontext); every later switch is _setjmp / _longjmp: no signal-mask system calls
  jmp_buf env;
  bool started = false;
  std::unique_ptr<char[]> stack;
  host_idx3 tid;
  int lane = 0, wave = 0;
  bool done = false;
};

struct Block {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  jmp_buf sched_env;
  Fiber* cur = nullptr;
  int live = 0, arrived = 0;
  unsigned gen = 0;
  unsigned long long progress = 0;
  const std::function<void()>* body = nullptr;
};

inline thread_local Block* g_blk = nullptr;

inline void yield() {
  Block* b = g_blk;
  Fiber* me = b->cur;
  if (_setjmp(me->env) == 0) _longjmp(b->sched_env, 1);
}

inline void block_sync() {
  Block* b = g_blk;
  const unsigned g = b->gen;
  ++b->progress;
  if (++b->arrived == b->live) {
    b->arrived = 0;
    ++b->gen;
  } else {
    while (b->gen == g) yield();
  }
}

inline Wave& my_wave() { return g_blk->waves[g_blk->cur->wave]; }
inline int my_lane() { return g_blk->cur->lane; }

inline void wave_sync() {
  Block* b = g_blk;
  Wave& w = my_wave();
  const unsigned g = w.gen;
  ++b->progress;
  if (++w.arrived == w.live) {
    w.arrived = 0;
    ++w.gen;
  } else {
    while (w.gen == g) yield();
  }
}

inline void fiber_main() {
  Block* b = g_blk;
  (*b->body)();
  // this GPU thread has returned: it no longer takes part in barriers / collectives
  Fiber* me = b->cur;
  me->done = true;
  ++b->progress;
  Wave& w = b->waves[me->wave];
  --w.live;
  if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; ++w.gen; }
  --b->live;
  if (b->live > 0 && b->arrived == b->live) { b->arrived = 0; ++b->gen; }
  _longjmp(b->sched_env, 1);      // back to the scheduler for good (this stack is never resumed)
}

inline void run_block(Block& blk, unsigned nt, dim3 block, const std::function<void()>& body) {
  g_blk = &blk;
  blk.body = &body;
  blk.live = static_cast<int>(nt);
  blk.arrived = 0;
  const int nw = static_cast<int>((nt + kWaveSize - 1) / kWaveSize);
  blk.waves.assign(0, Wave());
  blk.waves.resize(nw);
  if (blk.fibers.size() < nt) {
    const size_t have = blk.fibers.size();
    blk.fibers.resize(nt);
    for (size_t t = have; t < nt; ++t) blk.fibers[t].stack.reset(new char[kStack]);
  }
  for (unsigned t = 0; t < nt; ++t) {
    Fiber& f = blk.fibers[t];
    f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
    f.lane = static_cast<int>(t % kWaveSize);
    f.wave = static_cast<int>(t / kWaveSize);
    f.done = false;
    f.started = false;
    ++blk.waves[f.wave].live;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.get();
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, fiber_main, 0);
  }
  // Wave by wave: a wave keeps the processor while any of its lanes makes progress (its collectives release among its own 64
  // fibers: cycling all 256 fibers of the workgroup through every wave-level step was most of the switching); lanes that wait at a
  // workgroup barrier are polled once per pass over the waves.
  auto resume = [&](Fiber& f) {
    blk.cur = &f;
    threadIdx = f.tid;
    if (_setjmp(blk.sched_env) == 0) {
      if (!f.started) {
        f.started = true;
        setcontext(&f.ctx);
      } else {
        _longjmp(f.env, 1);
      }
    }
    // (here again when the fiber yielded or finished)
  };
  while (blk.live > 0) {
    const unsigned long long before = blk.progress;
    for (int w = 0; w < nw; ++w) {
      const unsigned t0 = static_cast<unsigned>(w) * kWaveSize, t1 = std::min(nt, t0 + kWaveSize);
      for (;;) {
        const unsigned long long wbefore = blk.progress;
        for (unsigned t = t0; t < t1; ++t)
          if (!blk.fibers[t].done) resume(blk.fibers[t]);
        if (blk.progress == wbefore) break;          // every live lane of this wave waits for another wave (or the wave is done)
      }
    }
    if (blk.progress == before && blk.live > 0) {
      std::fprintf(stderr, "hostexec: deadlock (a barrier or wave collective that not every live thread reaches)\n");
      std::abort();
    }
  }
}

inline thread_local std::vector<unsigned char> g_dyn_lds;       // `extern __shared__` of the workgroup in flight (see the test's source patch)
inline void* dyn_lds() { return g_dyn_lds.data(); }

// workgroup contexts (256 fiber stacks each) are kept across launches: allocating them per launch was most of the system time
inline std::mutex g_pool_mu;
inline std::vector<std::unique_ptr<Block>> g_pool;
inline std::unique_ptr<Block> pool_get() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_pool.empty()) return std::unique_ptr<Block>(new Block());
  std::unique_ptr<Block> b = std::move(g_pool.back());
  g_pool.pop_back();
  return b;
}
inline void pool_put(std::unique_ptr<Block> b) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pool.push_back(std::move(b));
}

template <class Body>
inline void launch(dim3 grid, dim3 block, size_t shmem, Body body) {
  blockDim = {block.x, block.y, block.z};
  gridDim = {grid.x, grid.y, grid.z};
  const unsigned nt = block.x * block.y * block.z;
  const unsigned long long nb = static_cast<unsigned long long>(grid.x) * grid.y * grid.z;
  const std::function<void()> fn = body;
  std::atomic<unsigned long long> next{0};
  const unsigned nthr = static_cast<unsigned>(std::min<unsigned long long>(nb, std::max(1u, std::thread::hardware_concurrency())));
  std::vector<std::thread> pool;
  for (unsigned i = 0; i < nthr; ++i)
    pool.emplace_back([&] {
      std::unique_ptr<Block> blk = pool_get();
      g_dyn_lds.assign(shmem + 64, 0);
      for (;;) {
        const unsigned long long k = next.fetch_add(1);
        if (k >= nb) break;
        blockIdx = {static_cast<unsigned>(k % grid.x), static_cast<unsigned>((k / grid.x) % grid.y),
                    static_cast<unsigned>(k / (static_cast<unsigned long long>(grid.x) * grid.y))};
        run_block(*blk, nt, block, fn);
      }
      pool_put(std::move(blk));
    });
  for (auto& th : pool) th.join();
}

// ---- wave collectives ---------------------------------------------------------------------------------------------------
template <class T>
inline T exchange(T mine, int src_lane) {
  static_assert(sizeof(T) <= 256, "fragment too large");
  Wave& w = my_wave();
  std::memcpy(w.buf[my_lane()], &mine, sizeof(T));
  wave_sync();
  T r;
  std::memcpy(&r, w.buf[src_lane & (kWaveSize - 1)], sizeof(T));
  wave_sync();
  return r;
}

// LDS-DMA of the strip kernels (global_load_lds_dwordx4): `pieces` requests of one wave, request k moves 16 bytes per lane from
// g + 1024 k (g is per lane) to LDS bytes [dst + 1024 k + 16 lane, + 16) -- dst is the wave-uniform M0 base, a numeric LDS address.
// The kernel's LDS array is registered once per thread (lds_register) under a fixed fake base, which maps such addresses back.
constexpr unsigned kLdsFakeBase = 0x10000u;
inline thread_local unsigned char* g_lds_host = nullptr;
inline unsigned lds_register(unsigned char* lds) { g_lds_host = lds; return kLdsFakeBase; }
// r06: COMPLETION IS DEFERRED.  A request only joins the wave's in-order queue of outstanding vector-memory
// requests; its bytes reach LDS when a counted wait retires it -- vm_wait(N) = s_waitcnt vmcnt(N): everything but the N newest requests
// of the wave completes, for ALL lanes (whichever lane gets there first does it: the hardware completes a request for the whole wave
// before the wave passes the wait).  A wait that is counted one too high therefore leaves a request in flight and the kernel reads
// what the LDS slot held BEFORE -- the stale read a mis-counted s_waitcnt produces on the GPU (until r05 the host completed every
// request at once and could not see that class of bug).  HOSTEXEC_DMA_EAGER=1 restores completion at issue (the other legal timing).
// vm_note(k): k other vector-memory requests of the wave (the register loads of the fp32-source forms) take their places in the queue.
inline bool dma_eager() { static const bool e = std::getenv("HOSTEXEC_DMA_EAGER") != nullptr; return e; }
inline void lds_dma(const void* g, unsigned dst, int pieces) {
  wave_sync();                         // every lane of the wave has passed its reads of what the slot held (the hardware issues the request
  unsigned char* d = g_lds_host + (dst - kLdsFakeBase) + 16 * my_lane();      //  for all lanes at one program point)
  const unsigned char* s = static_cast<const unsigned char*>(g);
  if (dma_eager()) {
    for (int k = 0; k < pieces; ++k) std::memcpy(d + 1024 * k, s + 1024 * k, 16);
  } else {
    auto& q = my_wave().vmq[my_lane()];
    for (int k = 0; k < pieces; ++k) q.push_back(PendingVm{d + 1024 * k, s + 1024 * k});
  }
  wave_sync();
}
inline void vm_note(int k) {
  wave_sync();
  if (!dma_eager()) {
    auto& q = my_wave().vmq[my_lane()];
    for (int i = 0; i < k; ++i) q.push_back(PendingVm{nullptr, nullptr});
  }
  wave_sync();
}
inline void vm_wait(int n) {
  Wave& w = my_wave();
  for (int l = 0; l < kWaveSize; ++l) {
    auto& q = w.vmq[l];
    while (static_cast<int>(q.size()) > n) {
      if (q.front().d) std::memcpy(q.front().d, q.front().s, 16);
      q.pop_front();
    }
  }
}

// ballot: LANE-LOCAL here ("does any lane of my wave ..." answered as "do I").  A true wave ballot would be a collective, and the
// sources call it inside divergent control flow (lanes that left a loop do not take part; the hardware masks them, fibers cannot
// know who will arrive).  Both uses in csrc (pointwise.hip corr_weight, lm.hip normal equations) are early exits that skip work
// whose contribution is exactly zero, so the per-lane answer changes no result.
inline unsigned long long ballot(bool pred) { return pred ? 1ull << my_lane() : 0ull; }

// D = A * B + C of one wave.  Register layouts (gfx950):
//   32x32xK : A row i = lane % 32, K slice kpl * (lane / 32) .. + kpl - 1;  B column j = lane % 32, same K slice;
//             D register v of lane l: row 8 * (v / 4) + 4 * (l / 32) + v % 4, column l % 32
//   16x16xK : A row i = lane % 16, K slice kpl * (lane / 16) ..;  B column j = lane % 16;  D register v: row 4 * (l / 16) + v, column l % 16
template <int MN, int KPL, class AV, class CV>
inline CV mfma(AV a, AV b, CV c) {
  Wave& w = my_wave();
  const int lane = my_lane();
  double af[KPL], bf[KPL];
  for (int k = 0; k < KPL; ++k) { af[k] = static_cast<double>(a[k]); bf[k] = static_cast<double>(b[k]); }
  std::memcpy(w.buf[lane], af, sizeof af);
  std::memcpy(w.buf[lane] + 128, bf, sizeof bf);
  wave_sync();
  constexpr int G = kWaveSize / MN;           // lane groups along K
  constexpr int NV = MN * MN / kWaveSize;     // D registers per lane
  const int j = lane % MN;
  for (int v = 0; v < NV; ++v) {
    const int i = MN == 32 ? 8 * (v / 4) + 4 * (lane / 32) + v % 4 : 4 * (lane / 16) + v;
    double s = c[v];                          // products of the instruction summed exactly, ONE rounding to fp32 per MFMA
    for (int g = 0; g < G; ++g) {
      const double* ar = reinterpret_cast<const double*>(w.buf[g * MN + i]);
      const double* bc = reinterpret_cast<const double*>(w.buf[g * MN + j] + 128);
      for (int k = 0; k < KPL; ++k) s += ar[k] * bc[k];
    }
    c[v] = static_cast<float>(s);
  }
  wave_sync();
  return c;
}

template <class CV>
inline CV mfma_32x32x2f32(float a, float b, CV c) {
  struct S { float v[1]; float operator[](int) const { return v[0]; } };
  return mfma<32, 1>(S{{a}}, S{{b}}, c);
}

}  // namespace hostexec

inline void __syncthreads() { hostexec::block_sync(); }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hostexec::launch((grid), (block), (shmem), [=] { kernel(__VA_ARGS__); })

#define __builtin_amdgcn_ballot_w64(pred) hostexec::ballot(pred)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_wave_barrier() hostexec::wave_sync()
#define __builtin_amdgcn_fence(...) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_s_barrier() hostexec::block_sync()
#define __builtin_amdgcn_readfirstlane(v) (v)          // (used on wave-uniform values only: it makes them scalar registers)
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_s_getreg(r) 0u
#define __builtin_amdgcn_exp2f(v) exp2f(v)
#define __builtin_amdgcn_rcpf(v) (1.0f / (v))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hostexec::mfma<32, 8>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) hostexec::mfma<16, 8>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, x, y, z) hostexec::mfma<16, 4>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hostexec::mfma_32x32x2f32((a), (b), (c))

template <class T> inline T __shfl_xor(T v, int m, int = 64) { return hostexec::exchange(v, hostexec::my_lane() ^ m); }
template <class T> inline T __shfl_down(T v, unsigned d, int = 64) {
  const int s = hostexec::my_lane() + static_cast<int>(d);
  return hostexec::exchange(v, s < hostexec::kWaveSize ? s : hostexec::my_lane());
}
template <class T> inline T __shfl_up(T v, unsigned d, int = 64) {
  const int s = hostexec::my_lane() - static_cast<int>(d);
  return hostexec::exchange(v, s >= 0 ? s : hostexec::my_lane());
}
template <class T> inline T __shfl(T v, int src, int = 64) { return hostexec::exchange(v, src); }
inline int __builtin_amdgcn_readlane(int v, int src) { return hostexec::exchange(v, src); }      // v_readlane_b32: lane `src` (wave-uniform) of v, to every lane

inline int __double2loint(double v) { long long b; std::memcpy(&b, &v, 8); return static_cast<int>(b); }
inline int __double2hiint(double v) { long long b; std::memcpy(&b, &v, 8); return static_cast<int>(b >> 32); }
inline double __hiloint2double(int hi, int lo) {
  const long long b = (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo);
  double v; std::memcpy(&v, &b, 8); return v;
}

// global-memory atomics (workgroups run on several OS threads)
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) {
  float old = *p, want;
  do { want = old + v; } while (!__atomic_compare_exchange(p, &old, &want, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}
inline double atomicAdd(double* p, double v) {
  double old = *p, want;
  do { want = old + v; } while (!__atomic_compare_exchange(p, &old, &want, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
  unsigned long long old = *p;
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
  unsigned long long old = *p;
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline int atomicMin(int* p, int v) {
  int old = *p;
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline int atomicMax(int* p, int v) {
  int old = *p;
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline unsigned atomicMin(unsigned* p, unsigned v) {
  unsigned old = *p;
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = *p;
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline double __longlong_as_double(long long b) { double v; std::memcpy(&v, &b, 8); return v; }
inline long long __double_as_longlong(double v) { long long b; std::memcpy(&b, &v, 8); return b; }
inline float rsqrtf(float v) { return 1.0f / sqrtf(v); }
inline unsigned __float_as_uint(float v) { unsigned u; std::memcpy(&u, &v, 4); return u; }
inline float __uint_as_float(unsigned u) { float v; std::memcpy(&v, &u, 4); return v; }
inline int __float_as_int(float v) { int u; std::memcpy(&u, &v, 4); return u; }
inline float __int_as_float(int u) { float v; std::memcpy(&v, &u, 4); return v; }
using std::max;
using std::min;

#include "common.hpp"      // (rp::set_error / fail_arg / check_launch / sat_counter: runtime_host.cpp compiles csrc/runtime.hip itself)
