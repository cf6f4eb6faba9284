// Stand-alone probe (plain HIP, no library code): is the output of kernel P always visible to the NEXT kernel C of the SAME
// stream when a second stream keeps the chip busy?  (profiles/r04_determinism.txt: the library's
// corr_weight kernel read 64-byte sectors of the flow map as they were before mask_upsample -- its in-stream predecessor -- wrote
// them, but only while a second queue was active.)
//
//   stream A, per iteration it:   P(it): 4-byte stores in mask_upsample's pattern (a 128-byte line of an up-sampled row is
//                                        completed by four store instructions of one wave, 32 bytes each; workgroups end at
//                                        staggered times) -- value = tag(it, index)
//                                 C(it): XCD-contiguous block order like corr_weight, plain loads of both planes, every word
//                                        compared with tag(it, index); a mismatch is classified: the value of iteration
//                                        it - NBUF (what the buffer held before P(it) ran) or something else
//   stream B (optional):          a kernel that keeps the matrix pipes or the memory system busy, launched back to back
//
// Device-side clocks (s_memrealtime, 100 MHz, chip-wide): P records the latest end of any of its workgroups, C the earliest start;
// "overlap" counts iterations in which a workgroup of C started BEFORE the last workgroup of P had finished its stores
// (in-stream ordering itself broken) -- as opposed to a cache-visibility effect, where C starts after P and still reads old data.
//
// Variants (bit flags of argv "fix"): 1 consumer starts with an agent-scope acquire (buffer_inv sc1), 2 consumer loads sc1
// (L1 bypass), 4 every producer workgroup ends with an agent-scope release (buffer_wbl2 sc1 + wait), 8 producer stores sc0 sc1
// (write-through), 16 producer writes whole 128-byte lines with 16-byte stores through an LDS transposition, 32 every producer
// wave waits for its own stores (s_waitcnt vmcnt(0)) before it ends, 64 (with 16) the whole-line stores are write-through
// (global_store_dwordx4 sc0 sc1) and every wave waits for them.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/kernel_visibility.hip -o tools/probes/kernel_visibility
//   tools/probes/kernel_visibility [iterations=2000]          (prints one line per configuration)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int Bn = 4, hL = 60, wL = 80, NL = hL * wL, Hf = 8 * hL, Wf = 8 * wL;      // half batch of the headline shape
constexpr long long Pf = (long long)Hf * Wf;
constexpr int MQ = 32;                                                                  // low-res pixels per producer workgroup
constexpr int NBUF = 3;

struct Stats {
  unsigned long long stale_old, stale_other;      // mismatching words: value of iteration it - NBUF / anything else
  unsigned long long p_end_max, c_start_min;      // per iteration (reset by the host through a tiny kernel)
  unsigned long long overlap_iters, stale_iters;  // iterations with C starting before P's last store / with >= 1 stale word
  unsigned int first[8];                          // first mismatch: it, index, got, expected
  unsigned int iter_flag;
};

__device__ __forceinline__ unsigned tag(unsigned it, unsigned idx) { return it * 0x9E3779B1u + idx * 0x85EBCA6Bu + 0x1234567u; }

__global__ void fill_kernel(unsigned* x, long long n, unsigned it) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] = tag(it, (unsigned)i);
}

// ---- P: the store pattern of csrc/mask_upsample.hip's epilogue.  Workgroup = 32 consecutive low-res pixels, 4 waves; wave ct owns
//      sub-pixel columns [16 ct, 16 ct + 16) (two sub-rows x 8 sub-columns) of every pixel; store instruction r writes pixel
//      16 (r >> 2) + 4 (lane >> 4) + (r & 3): 8 pieces of 32 bytes per instruction and plane.
template <int FIX>
__global__ __launch_bounds__(256) void producer(unsigned* __restrict__ up, unsigned it, unsigned spin, unsigned stagger, Stats* st) {
  __shared__ unsigned tile[2][8][MQ * 8 + 4];       // FIX & 16: [plane][sub-row si][pixel * 8 + sj]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4, ct = wave & 3;
  const long long m0 = (long long)blockIdx.x * MQ;
  // the real kernel computes for ~30 us; workgroups finish at different times
  {
    float a = (float)tid;
    const unsigned n = spin + ((blockIdx.x * 2654435761u) >> 24) * stagger;      // stagger 0: every workgroup ends at the same time, as the real kernel's do
    for (unsigned i = 0; i < n; ++i) a = __builtin_fmaf(a, 1.0000001f, 0.5f);
    if (a == 123.456f) up[0] = 0;                   // (keeps the loop)
  }
  const int sub = 16 * ct + l15, si = sub >> 3, sj = sub & 7;
  if (FIX & 16) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = 16 * (r >> 2) + 4 * lq + (r & 3);
      tile[0][si][row * 8 + sj] = 0; tile[1][si][row * 8 + sj] = 1;     // (planes: the tag is formed at store time)
    }
    __syncthreads();
    // 32 pixels x 8 sub-columns = 256 words = 1 KB per (plane, sub-row): 16 runs; thread -> one 16-byte piece of 4 of them
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int run = (tid >> 6) + 4 * q, pl = run >> 3, sr = run & 7, piece = tid & 63;      // 64 pieces of 16 bytes
      const long long m = m0 + piece / 2;                                                     // low-res pixel of the piece
      if (m < (long long)Bn * NL) {
        const unsigned mu = (unsigned)m, b = mu / NL, pix = mu - b * NL, Y = pix / wL, X = pix - Y * wL;
        const long long o = ((long long)b * 2 + pl) * Pf + (8LL * Y + sr) * Wf + 8 * X + 4 * (piece & 1);
        uint4 v;
        v.x = tag(it, (unsigned)o); v.y = tag(it, (unsigned)o + 1); v.z = tag(it, (unsigned)o + 2); v.w = tag(it, (unsigned)o + 3);
        if (FIX & 64) {
          typedef unsigned u4v __attribute__((ext_vector_type(4)));
          const u4v vv = {v.x, v.y, v.z, v.w};
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(up + o), "v"(vv) : "memory");
        } else {
          *reinterpret_cast<uint4*>(up + o) = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = 16 * (r >> 2) + 4 * lq + (r & 3);
      const long long m = m0 + row;
      if (m < (long long)Bn * NL) {
        const unsigned mu = (unsigned)m, b = mu / NL, pix = mu - b * NL, Y = pix / wL, X = pix - Y * wL;
        const long long o = (8LL * Y + si) * Wf + 8 * X + sj;
        const long long o0 = ((long long)b * 2 + 0) * Pf + o, o1 = ((long long)b * 2 + 1) * Pf + o;
        if (FIX & 8) {
          __hip_atomic_store(up + o0, tag(it, (unsigned)o0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(up + o1, tag(it, (unsigned)o1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
          up[o0] = tag(it, (unsigned)o0);
          up[o1] = tag(it, (unsigned)o1);
        }
      }
    }
  }
  if (FIX & (32 | 64)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (FIX & 4) {
    __syncthreads();
    if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  if (lane == 0) atomicMax(&st->p_end_max, (unsigned long long)__builtin_amdgcn_s_memrealtime());      // (every wave: after its last store was issued)
}

// ---- C: one thread per full-res pixel, XCD-contiguous block order (csrc/pointwise.hip corr_weight_kernel), both planes read.
template <int FIX>
__global__ __launch_bounds__(256) void consumer(const unsigned* up, unsigned it, unsigned it_old, Stats* st, unsigned* sink) {
  if (threadIdx.x == 0) atomicMin(&st->c_start_min, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  if (FIX & 1) {
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
  }
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int per = nblk >> 3, rem = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int bpi = (int)((Pf + 255) / 256);
  const int b = bid / bpi;
  const long long t = (long long)(bid - b * bpi) * 256 + threadIdx.x;
  if (t >= Pf) return;
  unsigned acc = 0;
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    const long long o = ((long long)b * 2 + pl) * Pf + t;
    unsigned v;
    if (FIX & 2) v = __hip_atomic_load(up + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else v = up[o];
    const unsigned want = tag(it, (unsigned)o);
    if (v != want) {
      const bool old = v == tag(it_old, (unsigned)o);
      atomicAdd(old ? &st->stale_old : &st->stale_other, 1ull);
      st->iter_flag = 1u;
      if (atomicCAS(&st->first[0], 0xffffffffu, it) == 0xffffffffu) { st->first[1] = (unsigned)o; st->first[2] = v; st->first[3] = want; st->first[4] = (unsigned)b; st->first[5] = (unsigned)(t / Wf); st->first[6] = (unsigned)(t % Wf); }
    }
    acc += v;
  }
  if (acc == 0x13572468u) sink[0] = acc;
}

__global__ void iter_begin(Stats* st) { st->p_end_max = 0; st->c_start_min = ~0ull; st->iter_flag = 0; }
__global__ void iter_end(Stats* st) {
  if (st->c_start_min < st->p_end_max) st->overlap_iters += 1;
  if (st->iter_flag) st->stale_iters += 1;
}

// ---- stream B load generators
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void busy_mfma(float* out, int n) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 1.2345f) out[0] = s;
}
__global__ __launch_bounds__(256) void busy_mem(const float4* __restrict__ src, float4* __restrict__ dst, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float4 v = src[i];
    v.x += 1.f;
    dst[i] = v;
  }
}

__global__ void tiny_kernel(unsigned* p, unsigned v) { if (threadIdx.x == 0 && v == 0xdeadbeefu) p[0] = v; }
__global__ void small_write_kernel(unsigned* p, unsigned v) { p[blockIdx.x * 256 + threadIdx.x] = v; }

template <int FIX>
static void launch_pc(unsigned* x, unsigned it, unsigned it_old, unsigned spin, unsigned stagger, Stats* st, unsigned* sink, hipStream_t s) {
  hipLaunchKernelGGL(iter_begin, dim3(1), dim3(1), 0, s, st);
  hipLaunchKernelGGL(producer<FIX>, dim3((Bn * NL + MQ - 1) / MQ), dim3(256), 0, s, x, it, spin, stagger, st);
  hipLaunchKernelGGL(consumer<FIX>, dim3((unsigned)(Bn * ((Pf + 255) / 256))), dim3(256), 0, s, x, it, it_old, st, sink);
  hipLaunchKernelGGL(iter_end, dim3(1), dim3(1), 0, s, st);
}
typedef void (*launch_fn)(unsigned*, unsigned, unsigned, unsigned, unsigned, Stats*, unsigned*, hipStream_t);
static launch_fn pick(int fix) {
  switch (fix) {
    case 0: return launch_pc<0>; case 1: return launch_pc<1>; case 2: return launch_pc<2>; case 3: return launch_pc<3>;
    case 4: return launch_pc<4>; case 5: return launch_pc<5>; case 8: return launch_pc<8>; case 10: return launch_pc<10>;
    case 16: return launch_pc<16>; case 17: return launch_pc<17>; case 32: return launch_pc<32>; case 36: return launch_pc<36>;
    case 33: return launch_pc<33>; case 37: return launch_pc<37>; case 40: return launch_pc<40>; case 80: return launch_pc<80>; case 81: return launch_pc<81>;
  }
  return nullptr;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const char* only = argc > 2 ? argv[2] : "";          // e.g. "mfma" : only configurations whose name contains it
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s (%s), %d CUs, clock %d kHz; %d iterations per configuration, %d rotating buffers of %.1f MB\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate, iters, NBUF, Bn * 2 * Pf * 4 / 1e6);
  const long long nx = (long long)Bn * 2 * Pf;
  unsigned* x[NBUF];
  for (int i = 0; i < NBUF; ++i) CK(hipMalloc(&x[i], nx * 4));
  unsigned* xb[NBUF];
  for (int i = 0; i < NBUF; ++i) CK(hipMalloc(&xb[i], nx * 4));
  unsigned* scratch; CK(hipMalloc(&scratch, 64 * 256 * 4));
  Stats* st; CK(hipMalloc(&st, sizeof(Stats)));
  Stats* stb; CK(hipMalloc(&stb, sizeof(Stats)));
  unsigned* sink; CK(hipMalloc(&sink, 64));
  float* bo; CK(hipMalloc(&bo, 64));
  const long long nb4 = 48LL << 20;                     // 768 MB moved per busy_mem launch
  float4 *bs, *bd; CK(hipMalloc(&bs, nb4 * 16)); CK(hipMalloc(&bd, nb4 * 16));
  CK(hipMemset(bs, 0, nb4 * 16));
  hipStream_t sa, sb, sc;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  struct Cfg { const char* name; int load; int fix; int graph; int nullstream; unsigned stagger; };
  const Cfg cfgs[] = {
      {"one stream (control)", 0, 0, 0, 0, 31},
      {"stream B: mfma", 1, 0, 0, 0, 31},
      {"stream B: memory copy", 2, 0, 0, 0, 31},
      {"stream B: mfma, stream C: memory", 3, 0, 0, 0, 31},
      {"stream B: mfma, A = null stream", 1, 0, 0, 1, 31},
      {"stream B: mfma, A as hipGraph (16 iterations per graph)", 1, 0, 1, 0, 31},
      {"uniform end; one stream (control)", 0, 0, 0, 0, 0},
      {"uniform end; stream B: mfma", 1, 0, 0, 0, 0},
      {"uniform end; stream B: memory copy", 2, 0, 0, 0, 0},
      {"uniform end; B mfma + C memory", 3, 0, 0, 0, 0},
      {"uniform end; B mfma + C memory; A = null stream", 3, 0, 0, 1, 0},
      {"uniform end; B mfma + C memory; fix 16 (whole-line stores)", 3, 16, 0, 0, 0},
      {"uniform end; B mfma + C memory; fix 1 (consumer acquire)", 3, 1, 0, 0, 0},
      {"uniform end; B mfma + C memory; fix 4 (producer release)", 3, 4, 0, 0, 0},
      {"uniform end; stream B: 8 tiny kernels per iteration", 4, 0, 0, 0, 0},
      {"uniform end; stream B: 8 small writing kernels per iteration", 8, 0, 0, 0, 0},
      {"uniform end; stream B: the same producer/consumer pair", 16, 0, 0, 0, 0},
      {"uniform end; stream B: tiny + the same pair", 20, 0, 0, 0, 0},
      {"staggered end; stream B: tiny + the same pair", 20, 0, 0, 0, 31},
      {"uniform end; B tiny + pair; A as hipGraph", 20, 0, 1, 0, 0},
      {"uniform end; B tiny + pair; fix 1 (consumer acquire)", 20, 1, 0, 0, 0},
      {"uniform end; B tiny + pair; fix 2 (consumer sc1 loads)", 20, 2, 0, 0, 0},
      {"uniform end; B tiny + pair; fix 32 (producer waves wait for stores)", 20, 32, 0, 0, 0},
      {"uniform end; B tiny + pair; fix 4 (producer agent release per WG)", 20, 4, 0, 0, 0},
      {"uniform end; B tiny + pair; fix 5 (release + acquire)", 20, 5, 0, 0, 0},
      {"uniform end; B tiny + pair; fix 40 (4-byte write-through stores + wait)", 20, 40, 0, 0, 0},
      {"uniform end; B tiny + pair; fix 16 (whole-line stores)", 20, 16, 0, 0, 0},
      {"uniform end; B tiny + pair; fix 80 (whole-line write-through + wait)", 20, 80, 0, 0, 0},
      {"uniform end; B tiny + pair; fix 81 (80 + consumer acquire)", 20, 81, 0, 0, 0},
      {"B mfma + C memory; fix 1 (consumer acquire)", 3, 1, 0, 0, 31},
      {"B mfma + C memory; fix 2 (consumer sc1 loads)", 3, 2, 0, 0, 31},
      {"B mfma + C memory; fix 32 (producer waves wait for their stores)", 3, 32, 0, 0, 31},
      {"B mfma + C memory; fix 4 (producer agent release per workgroup)", 3, 4, 0, 0, 31},
      {"B mfma + C memory; fix 36 (wait + release)", 3, 36, 0, 0, 31},
      {"B mfma + C memory; fix 5 (release + acquire)", 3, 5, 0, 0, 31},
      {"B mfma + C memory; fix 8 (producer write-through stores)", 3, 8, 0, 0, 31},
      {"B mfma + C memory; fix 10 (write-through stores + sc1 loads)", 3, 10, 0, 0, 31},
      {"B mfma + C memory; fix 16 (whole-line 16-byte stores)", 3, 16, 0, 0, 31},
      {"B mfma + C memory; fix 17 (whole lines + consumer acquire)", 3, 17, 0, 0, 31},
  };
  printf("%-72s %10s %10s %10s %10s %8s  first mismatch\n", "configuration", "stale_it", "overlap_it", "old_words", "other_wrds", "ms/iter");
  for (const Cfg& c : cfgs) {
    if (only[0] && !strstr(c.name, only)) continue;
    launch_fn fn = pick(c.fix);
    if (!fn) continue;
    Stats h;
    memset(&h, 0, sizeof(h));
    h.first[0] = 0xffffffffu;
    CK(hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice));
    CK(hipMemcpy(stb, &h, sizeof(h), hipMemcpyHostToDevice));
    for (int i = 0; i < NBUF; ++i) hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, x[i], nx, (unsigned)i);
    for (int i = 0; i < NBUF; ++i) hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, xb[i], nx, (unsigned)i);
    CK(hipDeviceSynchronize());
    hipStream_t A = c.nullstream ? (hipStream_t)0 : sa;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<hipGraphExec_t> execs;
    const int per_graph = 16;
    if (c.graph) {
      // NBUF * per_graph iterations cover every (buffer, iteration) combination only for fixed `it`; capture a graph per
      // block of iterations instead (instantiated up front, launched in order)
      for (int g0 = 0; g0 < iters; g0 += per_graph) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < per_graph && g0 + k < iters; ++k) {
          const unsigned it = NBUF + g0 + k;
          fn(x[it % NBUF], it, it - NBUF, 8000, c.stagger, st, sink, sa);
        }
        CK(hipStreamEndCapture(sa, &g));
        hipGraphExec_t ge;
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        execs.push_back(ge);
      }
    }
    CK(hipEventRecord(e0, A));
    for (int k = 0; k < iters; ++k) {
      const unsigned it = NBUF + k;
      if (c.graph) {
        if (k % per_graph == 0) CK(hipGraphLaunch(execs[k / per_graph], sa));
      } else {
        fn(x[it % NBUF], it, it - NBUF, 8000, c.stagger, st, sink, A);
      }
      if (c.load & 4) for (int q = 0; q < 8; ++q) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, sb, sink, (unsigned)q);
      if (c.load & 8) for (int q = 0; q < 8; ++q) hipLaunchKernelGGL(small_write_kernel, dim3(64), dim3(256), 0, sb, scratch, (unsigned)(k * 8 + q));
      if (c.load & 16) fn(xb[it % NBUF], it, it - NBUF, 8000, c.stagger, stb, sink, sb);
      if (c.load & 1) hipLaunchKernelGGL(busy_mfma, dim3(480), dim3(256), 0, sb, bo, 600);
      if (c.load & 2) hipLaunchKernelGGL(busy_mem, dim3(2048), dim3(256), 0, (c.load & 1) ? sc : sb, bs, bd, nb4 / 8);
    }
    CK(hipEventRecord(e1, A));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost));
    if (c.load & 16) {                      // stream B's pair checks itself too: add its counts
      Stats hb;
      CK(hipMemcpy(&hb, stb, sizeof(hb), hipMemcpyDeviceToHost));
      h.stale_iters += hb.stale_iters; h.overlap_iters += hb.overlap_iters; h.stale_old += hb.stale_old; h.stale_other += hb.stale_other;
      if (h.first[0] == 0xffffffffu) memcpy(h.first, hb.first, sizeof(h.first));
    }
    printf("%-72s %10llu %10llu %10llu %10llu %8.3f ", c.name, h.stale_iters, h.overlap_iters, h.stale_old, h.stale_other, ms / iters);
    if (h.first[0] != 0xffffffffu)
      printf(" it %u word %u (image %u row %u col %u): got %08x want %08x", h.first[0] - NBUF, h.first[1], h.first[4], h.first[5], h.first[6], h.first[2], h.first[3]);
    printf("\n");
    fflush(stdout);
    for (auto ge : execs) CK(hipGraphExecDestroy(ge));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
  return 0;
}
