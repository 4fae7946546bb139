// Microbenchmark: LDS atomic / store throughput per CU on gfx950 for the access
// patterns the histogram kernel can use.  Reports cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;

template <int MODE>
__global__ __launch_bounds__(1024) void bench(u64 *out, int iters, const uint32_t *rnd) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  uint32_t *lds32 = reinterpret_cast<uint32_t *>(lds);
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  uint32_t r = rnd[threadIdx.x];
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      r = r * 1664525u + 1013904223u;
      const uint32_t bin = (r >> 24);          // random bin 0..255
      if (MODE == 0) {                          // u64 atomic, [bin][64] cells, col distinct mod 16 in 16-lane groups
        const uint32_t col = (lane & 48) | ((k + lane) & 15);
        atomicAdd(&lds[bin * 64 + col], (u64)r);
      } else if (MODE == 1) {                   // u64 atomic, col distinct over all 64 lanes (col = lane)
        atomicAdd(&lds[bin * 64 + lane], (u64)r);
      } else if (MODE == 2) {                   // u32 atomic, [bin][64] cells u32, col = lane
        atomicAdd(&lds32[bin * 64 + lane], r);
      } else if (MODE == 3) {                   // u64 plain store same pattern as MODE 1
        lds[bin * 64 + lane] = r;
      } else if (MODE == 4) {                   // u64 atomic random column too (conflicting)
        atomicAdd(&lds[bin * 64 + ((r >> 8) & 63)], (u64)r);
      } else if (MODE == 5) {                   // two u32 atomics (sum lo + count) col = lane
        atomicAdd(&lds32[bin * 64 + lane], r);
        atomicAdd(&lds32[16384 + bin * 64 + lane], 1u);
      } else if (MODE == 6) {                   // u32 atomic, cols distinct mod 32 in 32-lane halves
        const uint32_t col = (lane & 32) | ((k + lane) & 31);
        atomicAdd(&lds32[bin * 64 + col], r);
      } else if (MODE == 7) {                   // f32 atomic add col = lane
        atomicAdd(reinterpret_cast<float *>(&lds32[bin * 64 + lane]), 1.0f);
      }
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (u64)(t1 - t0);
  if (lds[threadIdx.x] == 0x1234567) out[0] = 1;
}

template <int MODE>
void run(const char *name, int waves) {
  u64 *d_out; uint32_t *d_rnd;
  hipMalloc(&d_out, 256 * 8);
  hipMalloc(&d_rnd, 1024 * 4);
  std::vector<uint32_t> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 12345u * (i + 1) + 777u;
  hipMemcpy(d_rnd, h.data(), 4096, hipMemcpyHostToDevice);
  const int iters = 2000;
  hipFuncSetAttribute((const void *)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(waves * 64), 160 * 1024 - 64, 0, d_out, iters, d_rnd);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(waves * 64), 160 * 1024 - 64, 0, d_out, iters, d_rnd);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  u64 hc[256]; hipMemcpy(hc, d_out, 256 * 8, hipMemcpyDeviceToHost);
  const double ninstr = (double)iters * 16 * waves * (MODE == 5 ? 2 : 1);
  // wall-based: cycles at ~2.4 GHz per wave-instruction per CU
  printf("%-44s waves/CU %2d: %.2f us  -> %.2f ns per wave-instr per CU (clock64 %.1f ticks/instr)\n", name, waves,
         ms * 1e3, ms * 1e6 / ninstr, (double)hc[1] / ninstr);
  hipFree(d_out); hipFree(d_rnd);
}

int main() {
  for (int w : {4, 16}) {
    run<0>("u64 atomic, 16-lane-group conflict-free", w);
    run<1>("u64 atomic, col = lane", w);
    run<2>("u32 atomic, col = lane", w);
    run<3>("u64 store, col = lane", w);
    run<4>("u64 atomic, random col (conflicts)", w);
    run<5>("2 x u32 atomic (sum + count), col = lane", w);
    run<6>("u32 atomic, 32-lane-half conflict-free", w);
    run<7>("f32 atomic, col = lane", w);
  }
  return 0;
}
