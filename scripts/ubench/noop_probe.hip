// Microbenchmark: why does a growth-step launch that finds nothing to do take ~4.5 us when a
// trivial kernel takes 1.55 us in the same chain (launch_chain.hip)?  Candidates: the size of
// the kernel's code (instruction fetch of a cold kernel), the kernel-argument block, the
// 98 KB of dynamic LDS, the descriptor read (32 B per workgroup, fresh or not).
//   hipcc --offload-arch=gfx950 -O3 -w -o scripts/ubench/noop_probe scripts/ubench/noop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct Desc { uint32_t begin, count, slot, a, b, c, d, e; };

// BODY = dependent multiply-adds behind the exit test (never executed): code size only
template <int BODY, int VARIANT>
__global__ __launch_bounds__(1024) void k_noop(const Desc *__restrict__ wgs, const uint32_t *__restrict__ p1,
                                               const uint32_t *__restrict__ p2, const uint32_t *__restrict__ p3,
                                               const double *__restrict__ p4, const double *__restrict__ p5,
                                               uint32_t *__restrict__ out, int x1, int x2) {
  extern __shared__ char lds[];
  const Desc d = wgs[blockIdx.x];
  if (d.count == 0) return;
  uint32_t x = d.begin + threadIdx.x + (uint32_t)VARIANT;
#pragma unroll
  for (int i = 0; i < BODY; ++i) {
    x = x * (uint32_t)x1 + p1[(x >> 20) & 15] + (uint32_t)x2;
    asm volatile("" : "+v"(x));
  }
  out[threadIdx.x] = x + (uint32_t)lds[threadIdx.x] + p2[0] + p3[0] + (uint32_t)p4[0] + (uint32_t)p5[0];
}

__global__ void k_fill(Desc *wgs, int n) {   // rewrites the descriptors (same zeros): fresh data
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { Desc z = {0, 0, 0, 0, 0, 0, 0, 0}; wgs[i] = z; }
}

__global__ void k_block(uint32_t *out, long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (ticks < 0) out[0] = 1;
}

static Desc *d_wgs;
static uint32_t *d_u;
static double *d_d;

template <typename F>
static float chain(int reps, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int w = 0; w < 2; ++w) {
    hipLaunchKernelGGL(k_block, dim3(1), dim3(64), 0, 0, d_u, (long long)(reps * 12 * 100));
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch(i);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms * 1e3f / reps;
}

#define NOOP(BODY, VAR, grid, block, lds) \
  hipLaunchKernelGGL((k_noop<BODY, VAR>), dim3(grid), dim3(block), lds, 0, d_wgs, d_u, d_u, d_u, d_d, d_d, d_u, 3, 5)

int main() {
  hipMalloc(&d_wgs, 4096 * sizeof(Desc));
  hipMemset(d_wgs, 0, 4096 * sizeof(Desc));
  hipMalloc(&d_u, 1 << 16);
  hipMalloc(&d_d, 1 << 16);
  hipFuncSetAttribute((const void *)k_noop<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  hipFuncSetAttribute((const void *)k_noop<1500, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  hipFuncSetAttribute((const void *)k_noop<1500, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  hipFuncSetAttribute((const void *)k_noop<1500, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  const int R = 300;
  printf("tiny body, 256 x 1024, no LDS            : %.2f us\n", chain(R, [&](int) { NOOP(0, 0, 256, 1024, 0); }));
  printf("tiny body, 256 x 1024, 98 KB LDS         : %.2f us\n", chain(R, [&](int) { NOOP(0, 0, 256, 1024, 98304); }));
  printf("~24 KB body, 256 x 1024, 98 KB LDS       : %.2f us\n", chain(R, [&](int) { NOOP(1500, 0, 256, 1024, 98304); }));
  printf("three ~24 KB kernels in rotation         : %.2f us\n", chain(R, [&](int i) {
    if (i % 3 == 0) NOOP(1500, 0, 256, 1024, 98304);
    else if (i % 3 == 1) NOOP(1500, 1, 272, 1024, 0);
    else NOOP(1500, 2, 491, 256, 0);
  }));
  printf("fill + tiny body (fresh descriptors)     : %.2f us per pair\n", chain(R, [&](int) {
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, 0, d_wgs, 4096);
    NOOP(0, 0, 256, 1024, 98304);
  }));
  printf("fill + ~24 KB body (fresh descriptors)   : %.2f us per pair\n", chain(R, [&](int) {
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, 0, d_wgs, 4096);
    NOOP(1500, 0, 256, 1024, 98304);
  }));
  printf("fill alone                               : %.2f us\n", chain(R, [&](int) {
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, 0, d_wgs, 4096);
  }));
  return 0;
}
