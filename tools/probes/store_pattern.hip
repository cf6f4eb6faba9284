// Probe (not part of the library): how fast does MI355X absorb the level-0 store pattern of the correlation-volume kernel?
// One workgroup = 128 i rows x one 8 x 16 patch of the j image (as csrc/corr_pyramid.hip), every output written exactly
// once, no loads, no arithmetic.  Patterns differ only in WHICH bytes one wave instruction covers and in the order.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_pattern.hip -o tools/probes/store_pattern && tools/probes/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int B = 8, H = 60, W = 80, N = H * W, BM = 128, PY = 8, PX = 16, ST = 8;

__device__ __forceinline__ bool tile_of(int& b, int& i0, int& y0, int& x0) {
  const int n_it = (N + BM - 1) / BM, n_py = (H + PY - 1) / PY, n_px = (W + PX - 1) / PX;
  const int ntiles = gridDim.x;
  int bid = blockIdx.x;
  const int per = ntiles >> 3, rem = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
  bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  const int n_patch = n_py * n_px, n_ps = (n_patch + ST - 1) / ST, n_is = (n_it + ST - 1) / ST, per_img = n_ps * n_is * ST * ST;
  b = bid / per_img;
  const int tloc = bid - b * per_img, sidx = tloc / (ST * ST), within = tloc - sidx * (ST * ST);
  const int it = (sidx / n_ps) * ST + within / ST, patch = (sidx % n_ps) * ST + within % ST;
  if (it >= n_it || patch >= n_patch) return false;
  i0 = it * BM; y0 = (patch / n_px) * PY; x0 = (patch % n_px) * PX;
  return true;
}

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void pat(float* __restrict__ p) {
  int b, i0, y0, x0;
  if (!tile_of(b, i0, y0, x0)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* pb = p + (long long)b * N * N;
  const f4v v = {1.f, 2.f, 3.f, (float)lane};
  if (MODE == 0) {            // as the kernel: per sub-tile s (2 patch rows), 4 instructions of 8 i rows x 2 y x 64 B
    for (int s = 0; s < 4; ++s)
      for (int q = 0; q < 4; ++q) {
        const int f = lane + 64 * q, row = f >> 3, c = (f & 7) << 2, yy = c >> 4, xx = c & 15;
        const int i = i0 + wave * 32 + row, y = y0 + 2 * s + yy, x = x0 + xx;
        if (i < N && y < H) {
          float* d = pb + (long long)i * N + y * W + x;
          if (NT) __builtin_nontemporal_store(v, (f4v*)d); else *(f4v*)d = v;
        }
      }
  } else if (MODE == 1) {     // 2 i rows x 8 y x 64 B per instruction: the 8 pieces of a row's patch leave together
    for (int q = 0; q < 16; ++q) {
      const int row = 2 * q + (lane >> 5), yy = (lane >> 2) & 7, xx = (lane & 3) << 2;
      const int i = i0 + wave * 32 + row, y = y0 + yy, x = x0 + xx;
      if (i < N && y < H) {
        float* d = pb + (long long)i * N + y * W + x;
        if (NT) __builtin_nontemporal_store(v, (f4v*)d); else *(f4v*)d = v;
      }
    }
  } else if (MODE == 2) {     // straight from the MFMA C layout: dword stores, 2 i rows x 2 y x 64 B per instruction, r-major
    const int kh = lane >> 5, l31 = lane & 31, yy = l31 >> 4, xx = l31 & 15;
    for (int r = 0; r < 16; ++r)
      for (int s = 0; s < 4; ++s) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int i = i0 + wave * 32 + row, y = y0 + 2 * s + yy, x = x0 + xx;
        if (i < N && y < H) {
          float* d = pb + (long long)i * N + y * W + x;
          if (NT) __builtin_nontemporal_store(v.x, d); else *d = v.x;
        }
      }
  } else if (MODE == 3) {     // same bytes per workgroup, but as full image rows: 128 i rows x 1 KB contiguous pieces
    // (what a tile owning whole rows of the j image would write): workgroup t writes bytes [t * 64 KB, +64 KB) of image b
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    for (int q = 0; q < 16; ++q) {
      const long long o = (long long)blockIdx.x * 16384 + q * 1024 + threadIdx.x * 4;
      if (o + 3 < (long long)B * N * N) {
        if (NT) __builtin_nontemporal_store(v, (f4v*)(p + o)); else *(f4v*)(p + o) = v;
      }
    }
    (void)base;
  } else if (MODE >= 5) {     // NP = 2, 4 or 5 x-adjacent patches per workgroup (one per wave group), 128 / NP i rows each:
    // the workgroup writes 128 / 256 / 320 B contiguous pieces within one epilogue.  Tiles are re-indexed so that the
    // same bytes are covered: tile (it, patch) -> i rows [it*128 + (px % NP') * ..] -- emulated by splitting the wave's work
    constexpr int NP = MODE == 5 ? 2 : (MODE == 6 ? 4 : 5);
    // each wave: for its 32 i rows of the tile, loop the NP patches of its x group but only the rows (32 / NP)-th share
    const int n_px = (W + PX - 1) / PX;
    const int px = x0 / PX, g0 = (px / NP) * NP;             // first patch of this tile's x group
    const int members = (g0 + NP <= n_px ? NP : n_px - g0);
    const int me = px - g0;                                    // which i share this tile owns inside the group
    // tile `me` of the group writes i rows [me * 128 / members, (me + 1) * 128 / members) of ALL the group's patches
    const int r_lo = me * 128 / members, r_hi = (me + 1) * 128 / members;
    for (int s = 0; s < 4; ++s)
      for (int idx = threadIdx.x >> 3; idx < (r_hi - r_lo) * members; idx += 32) {
        // 8 lanes x float4 = one patch row pair; consecutive 8-lane groups take the x-adjacent patches of one i row
        const int rr = r_lo + idx / members, sub = idx % members, c = (threadIdx.x & 7) << 2, yy = c >> 4, xx = c & 15;
        const int i = i0 + rr, y = y0 + 2 * s + yy, x = (g0 + sub) * PX + xx;
        if (i < N && y < H) {
          float* d = pb + (long long)i * N + y * W + x;
          if (NT) __builtin_nontemporal_store(v, (f4v*)d); else *(f4v*)d = v;
        }
      }
  } else if (MODE == 8 || MODE == 9) {   // one workgroup walks the 5 patches of its stripe one after the other (as a kernel
    // looping its K loop per patch would), MODE 9 with ~8 us between them: do the 64 B pieces still merge downstream?
    if (x0 != 0) return;
    for (int sub = 0; sub < (W + PX - 1) / PX; ++sub) {
      for (int s = 0; s < 4; ++s)
        for (int q = 0; q < 4; ++q) {
          const int f = lane + 64 * q, row = f >> 3, c = (f & 7) << 2, yy = c >> 4, xx = c & 15;
          const int i = i0 + wave * 32 + row, y = y0 + 2 * s + yy, x = sub * PX + xx;
          if (i < N && y < H) {
            float* d = pb + (long long)i * N + y * W + x;
            if (NT) __builtin_nontemporal_store(v, (f4v*)d); else *(f4v*)d = v;
          }
        }
      if (MODE == 9) for (int k = 0; k < 2; ++k) __builtin_amdgcn_s_sleep(127);     // 2 x 127 x 64 clk ~ 7 us
    }
  } else if (MODE == 4) {     // stripe tile: 32 i rows x (8 image rows x 80 = 2560 B contiguous) per workgroup pass, 4 passes
    // workgroup = (i tile of 128, stripe y0 of 8 rows); same number of bytes per workgroup x 5
    for (int q = 0; q < 80; ++q) {            // 128 rows x 640 floats = 81920 floats / 256 threads / 4
      const int f = q * 256 + threadIdx.x, row = f / 160, c = (f % 160) * 4;
      const int i = i0 + row, y = y0 + c / W, x = c % W;
      if (i < N && y < H && x0 == 0) {
        float* d = pb + (long long)i * N + y * W + x;
        if (NT) __builtin_nontemporal_store(v, (f4v*)d); else *(f4v*)d = v;
      }
    }
  }
}

template <int MODE, bool NT>
float run(float* p, int grid, const char* name) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((pat<MODE, NT>), dim3(grid), dim3(256), 0, 0, p);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms / 10 < best) best = ms / 10;
  }
  const double bytes = 4.0 * B * N * N;
  printf("%-58s %s %8.1f us  %7.1f GB/s\n", name, NT ? "nt " : "   ", best * 1e3, bytes / best / 1e6);
  return best;
}

int main() {
  float* p;
  const size_t bytes = (size_t)B * N * N * 4;
  if (hipMalloc(&p, bytes + (1 << 20)) != hipSuccess) return 1;
  hipMemset(p, 0, bytes);
  const int n_it = (N + BM - 1) / BM, n_patch = ((H + PY - 1) / PY) * ((W + PX - 1) / PX);
  const int grid = B * ((n_it + ST - 1) / ST) * ((n_patch + ST - 1) / ST) * ST * ST;
  const int grid_lin = (int)((bytes / 4 + 16383) / 16384);
  run<0, false>(p, grid, "kernel's pattern (8 i rows x 2 y x 64 B per instruction)");
  run<0, true>(p, grid, "kernel's pattern (8 i rows x 2 y x 64 B per instruction)");
  run<1, false>(p, grid, "2 i rows x 8 y x 64 B per instruction");
  run<1, true>(p, grid, "2 i rows x 8 y x 64 B per instruction");
  run<2, false>(p, grid, "dword stores from the C layout, r-major");
  run<2, true>(p, grid, "dword stores from the C layout, r-major");
  run<4, false>(p, grid, "stripe tiles: 2560 B contiguous per i row (x0 == 0 tiles)");
  run<4, true>(p, grid, "stripe tiles: 2560 B contiguous per i row (x0 == 0 tiles)");
  run<5, false>(p, grid, "2 adjacent patches per workgroup: 128 B pieces");
  run<6, false>(p, grid, "4 adjacent patches per workgroup: 256 B pieces (+ 64 B tail group)");
  run<7, false>(p, grid, "5 adjacent patches per workgroup: 320 B pieces (one image row)");
  run<8, false>(p, grid, "stripe walked patch by patch by one workgroup, back to back");
  run<9, false>(p, grid, "stripe walked patch by patch, ~7 us between patches");
  run<9, true>(p, grid, "stripe walked patch by patch, ~7 us between patches");
  run<3, false>(p, grid_lin, "linear 64 KB per workgroup");
  run<3, true>(p, grid_lin, "linear 64 KB per workgroup");
  hipFree(p);
  return 0;
}
