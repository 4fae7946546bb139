// k_ubench.hip -- the ceiling the root histogram launch is priced against, measured in the
// caller's own process (qr_prof_lds_atomic; bench.py's `roofline.lds_atomic_bound`).
// A MEASUREMENT AID, built as a library of its own (lib/libqr_ubench.so: quickrank_amd/build.py):
// libqr_hip.so holds no microbenchmark and loads this one only when qr_prof_lds_atomic is called.
//
// k_hist_root issues ONE ds_add_u64 per (document, accumulated column) and nothing it does can
// go faster than the CU's LDS pipeline retires them.  This kernel issues the
// same instruction in the same access pattern -- [bin][64] cells of 8 bytes, random bins, the
// sixteen lanes of an LDS lane group on sixteen distinct columns mod 16, sixteen waves per CU,
// one workgroup per CU -- with nothing else in the loop, and reports shader cycles per wave
// instruction (s_memtime) and the shader clock it ran at (cycles / s_memrealtime at 100 MHz).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <vector>

typedef unsigned long long u64;

__global__ __launch_bounds__(1024) void k_ubench_lds_atomic(u64 *out, const int iters) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t i = threadIdx.x; i < 16384u; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  uint32_t r = 12345u * (threadIdx.x + 1) + 777u + blockIdx.x;
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
      r = r * 1664525u + 1013904223u;
      const uint32_t bin = r >> 24;
      const uint32_t col = (lane & 48u) | ((k + lane) & 15u);
      atomicAdd(&lds[bin * 64u + col], (u64)r);
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = (u64)(t1 - t0);
    out[2 * blockIdx.x + 1] = (u64)(w1 - w0);
  }
  if (lds[threadIdx.x] == 0x1234567ull) out[0] = 1;  // (keeps the atomics alive)
}

// 0 on success, a hipError_t otherwise
extern "C" int qr_ubench_lds_atomic(int device, int ncu, void *stream, double *cycles_per_instr, double *shader_ghz) {
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return (int)e;
  const int G = ncu, waves = 16, iters = 600;
  const size_t lds = 16384 * 8;
  u64 *d_out = nullptr;
  if ((e = hipMalloc((void **)&d_out, (size_t)G * 16)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void *)k_ubench_lds_atomic, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds)) != hipSuccess) {
    (void)hipFree(d_out);
    return (int)e;
  }
  std::vector<u64> h((size_t)G * 2);
  double best_cyc = 0.0, best_ghz = 0.0;
  for (int rep = 0; rep < 3; ++rep) {  // (the first launch also warms the clocks up: the best of three)
    hipLaunchKernelGGL(k_ubench_lds_atomic, dim3(G), dim3(waves * 64), lds, st, d_out, iters);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess ||
        (e = hipMemcpy(h.data(), d_out, (size_t)G * 16, hipMemcpyDeviceToHost)) != hipSuccess) {
      (void)hipFree(d_out);
      return (int)e;
    }
    // the median workgroup's cycles over the wave instructions ONE CU retired
    std::vector<u64> cyc((size_t)G), wall((size_t)G);
    for (int i = 0; i < G; ++i) {
      cyc[(size_t)i] = h[2 * (size_t)i];
      wall[(size_t)i] = h[2 * (size_t)i + 1];
    }
    std::nth_element(cyc.begin(), cyc.begin() + G / 2, cyc.end());
    std::nth_element(wall.begin(), wall.begin() + G / 2, wall.end());
    const double ninstr = (double)iters * 16 * waves;
    const double cpi = (double)cyc[(size_t)G / 2] / ninstr;
    const double ghz = wall[(size_t)G / 2] ? (double)cyc[(size_t)G / 2] / ((double)wall[(size_t)G / 2] * 10.0) : 0.0;
    if (rep == 0 || cpi < best_cyc) {
      best_cyc = cpi;
      best_ghz = ghz;
    }
  }
  (void)hipFree(d_out);
  if (cycles_per_instr) *cycles_per_instr = best_cyc;
  if (shader_ghz) *shader_ghz = best_ghz;
  return 0;
}
