// Does a ds_add_u64 with fewer active lanes retire faster?  (the padding columns of
// the third feature block are 10 of a wave's 63 busy lanes at half of the sixteen steps.)
// Same loop as k_ubench_lds_atomic (csrc/k_ubench.hip); `mask` says which lanes issue the atomic.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_lanes scripts/ubench/lds_atomic_lanes.hip && /tmp/lds_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;

__global__ __launch_bounds__(1024) void k(u64 *out, const int iters, const u64 mask, const int pad_mode) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t i = threadIdx.x; i < 16384u; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  uint32_t r = 12345u * (threadIdx.x + 1) + 777u + blockIdx.x;
  const bool on = (mask >> lane) & 1ull;
  // pad_mode: lanes of "chunk 2" (lane % 3 == 2) skip the steps whose rotated column is >= 8
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (uint32_t kk = 0; kk < 16; ++kk) {
      r = r * 1664525u + 1013904223u;
      const uint32_t bin = r >> 24;
      const uint32_t col = (lane & 48u) | ((kk + lane) & 15u);
      bool go = on;
      if (pad_mode) go = !((lane % 3u) == 2u && ((kk + lane) & 15u) >= 8u) && lane < 63;
      if (go) atomicAdd(&lds[bin * 64u + col], (u64)r);
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (u64)(t1 - t0);
  if (lds[threadIdx.x] == 0x1234567ull) out[0] = 1;
}

int main() {
  int G = 256;
  const int iters = 600, waves = 16;
  u64 *d;
  hipMalloc(&d, G * 8);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
  struct { const char *name; u64 mask; int pad; } cases[] = {
      {"64 lanes", ~0ull, 0}, {"63 lanes (lane 63 idle, as fw = 48)", ~0ull >> 1, 0},
      {"63 lanes, chunk-2 lanes skip their 8 padding steps", ~0ull, 1},
      {"48 lanes", 0x0000FFFFFFFFFFFFull, 0}, {"32 lanes (low half)", 0xFFFFFFFFull, 0},
      {"32 lanes (even lanes)", 0x5555555555555555ull, 0}, {"16 lanes", 0xFFFFull, 0}, {"1 lane", 1ull, 0}};
  for (auto &c : cases) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k, dim3(G), dim3(waves * 64), 16384 * 8, 0, d, iters, c.mask, c.pad);
      hipDeviceSynchronize();
      std::vector<u64> h(G);
      hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
      std::nth_element(h.begin(), h.begin() + G / 2, h.end());
      best = std::min(best, (double)h[G / 2] / ((double)iters * 16 * waves));
    }
    printf("%-55s %.2f cycles per wave instruction\n", c.name, best);
  }
  return 0;
}
