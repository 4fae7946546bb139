// Microbenchmark: what COLD straight-line code costs on the MI355X.  The control step of a
// growth step (k_decide_part) is ~6 KB of single-lane code executed once per launch, and every
// launch starts with cold instruction caches: is it bound by instruction fetch?
// One wave runs N dependent integer multiply-adds, fully unrolled (N x ~12 B of code, every
// instruction fetched once) or as a rolled loop (the same work out of a few cache lines).
//   hipcc --offload-arch=gfx950 -O3 -w -o scripts/ubench/icache scripts/ubench/icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int N, bool UNROLL>
__global__ void k_code(uint32_t *out, uint32_t a, uint32_t b) {
  uint32_t x = threadIdx.x + blockIdx.x;
  const long long t0 = clock64();
  if (UNROLL) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      x = x * a + b;
      asm volatile("" : "+v"(x));   // (keeps every step a separate instruction sequence)
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
      x = x * a + b;
      asm volatile("" : "+v"(x));
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = x;
    out[2 * blockIdx.x + 1] = (uint32_t)(t1 - t0);
  }
}

// another kernel's worth of code in between (evicts, like the other kernels of a step)
template <int N>
__global__ void k_other(uint32_t *out, uint32_t a, uint32_t b) {
  uint32_t x = threadIdx.x;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    x = (x ^ a) + b;
    asm volatile("" : "+v"(x));
  }
  if (x == 0x12345) out[0] = x;
}

static uint32_t *d_out;

template <int N, bool UNROLL>
static void run(int grid, bool evict) {
  uint32_t h[2 * 512];
  double tot = 0, cyc0 = 0, cycmax = 0;
  const int reps = 50;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < reps + 5; ++r) {
    if (evict) hipLaunchKernelGGL(k_other<6000>, dim3(256), dim3(256), 0, 0, d_out + 2048, 3u, 5u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_code<N, UNROLL>), dim3(grid), dim3(64), 0, 0, d_out, 1664525u, 1013904223u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d_out, sizeof(uint32_t) * 2 * grid, hipMemcpyDeviceToHost);
    if (r >= 5) {
      tot += ms;
      cyc0 += h[1];
      uint32_t mx = 0;
      for (int g = 0; g < grid; ++g) mx = h[2 * g + 1] > mx ? h[2 * g + 1] : mx;
      cycmax += mx;
    }
  }
  printf("N %5d %-8s grid %3d evict %d: %.2f us per launch, wg0 %.0f cycles (%.1f per step), slowest wg %.0f\n", N,
         UNROLL ? "unrolled" : "rolled", grid, (int)evict, tot * 1e3 / reps, cyc0 / reps, cyc0 / reps / N, cycmax / reps);
}

int main() {
  hipMalloc(&d_out, 1 << 16);
  for (int grid : {1, 256, 512}) {
    for (int ev = 0; ev < 2; ++ev) {
      run<512, false>(grid, ev);
      run<512, true>(grid, ev);
      run<2048, false>(grid, ev);
      run<2048, true>(grid, ev);
    }
  }
  return 0;
}
