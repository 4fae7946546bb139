// Microbenchmark: what one link of a chain of dependent launches costs on the MI355X, by
// grid size, workgroup size, dynamic LDS and the number of dependent global reads a
// workgroup makes before it can leave (the "nothing to do" launches of a growth step, and
// the floor under every kernel of the per-step chain).  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -w -o scripts/ubench/launch_chain scripts/ubench/launch_chain.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int LINKS>
__global__ void k_chain(const uint32_t *__restrict__ a, uint32_t *__restrict__ out) {
  extern __shared__ char lds[];
  uint32_t x = blockIdx.x * 64u;          // one 256-byte line per workgroup
#pragma unroll
  for (int i = 0; i < LINKS; ++i) x = a[x];
  if (x == 0xdeadbeefu) out[threadIdx.x] = x + (uint32_t)lds[threadIdx.x];
}

// the same chain over FRESH data: every launch rewrites (with the same values) the words the
// next launch's chain will read -- workgroup b those of workgroup b + 1, i.e. of another XCD
template <int LINKS>
__global__ void k_fresh(uint32_t *a, uint32_t *__restrict__ out) {
  uint32_t x = blockIdx.x * 64u;
#pragma unroll
  for (int i = 0; i < LINKS; ++i) x = a[x];
  if (x == 0xdeadbeefu) out[threadIdx.x] = x;
  if (threadIdx.x == 0) {
    uint32_t y = ((blockIdx.x + 1) % gridDim.x) * 64u;
#pragma unroll
    for (int i = 0; i < LINKS; ++i) {
      const uint32_t nx = (uint32_t)(((size_t)y * 64u + 4096u * 17u) % (1u << 20));
      a[y] = nx;
      y = nx;
    }
  }
}

// keeps the GPU busy while the host enqueues the whole chain (the host needs ~3.5 us per
// launch: without this the short links would measure the host)
__global__ void k_block(uint32_t *out, long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (ticks < 0) out[0] = 1;
}

static uint32_t *d_a, *d_out;

template <int LINKS>
static void run(int grid, int block, size_t lds, int ev) {
  hipFuncSetAttribute((const void *)k_chain<LINKS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  const int reps = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<hipEvent_t> evs;
  for (int w = 0; w < 2; ++w) {
    hipLaunchKernelGGL(k_block, dim3(1), dim3(64), 0, 0, d_out, (long long)(reps * 6 * 100));   // reps x 6 us at 100 MHz
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) {
      if (ev && (i % ev) == 0) {   // events on the launch itself, as bench.py's roofline legs do
        hipEvent_t a0, a1;
        hipEventCreate(&a0);
        hipEventCreate(&a1);
        hipExtLaunchKernelGGL(k_chain<LINKS>, dim3(grid), dim3(block), lds, 0, a0, a1, 0, d_a, d_out);
        evs.push_back(a0);
        evs.push_back(a1);
      } else
        hipLaunchKernelGGL(k_chain<LINKS>, dim3(grid), dim3(block), lds, 0, d_a, d_out);
    }
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("links %d grid %4d block %4d lds %6zu events 1/%d : %.2f us per launch\n", LINKS, grid, block, lds, ev,
         ms * 1e3 / reps);
  for (auto e : evs) hipEventDestroy(e);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

template <int LINKS>
static void run_fresh(int grid, int block) {
  const int reps = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) {
    hipLaunchKernelGGL(k_block, dim3(1), dim3(64), 0, 0, d_out, (long long)(reps * 12 * 100));
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_fresh<LINKS>, dim3(grid), dim3(block), 0, 0, d_a, d_out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("FRESH links %2d grid %4d block %4d : %.2f us per launch\n", LINKS, grid, block, ms * 1e3 / reps);
}

int main() {
  const size_t n = 1u << 20;
  std::vector<uint32_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)((i * 64u + 4096u * 17u) % n);   // another line, another page
  hipMalloc(&d_a, n * 4);
  hipMalloc(&d_out, 4096);
  hipMemcpy(d_a, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int grid : {1, 64, 256, 512, 2048})
    for (int block : {64, 256, 1024}) {
      run<0>(grid, block, 0, 0);
      run<1>(grid, block, 0, 0);
    }
  run<0>(256, 1024, 98304, 0);
  run<1>(256, 1024, 98304, 0);
  run<2>(256, 1024, 98304, 0);
  run<3>(256, 1024, 98304, 0);
  run<2>(512, 256, 0, 0);
  run<3>(512, 256, 0, 0);
  run<8>(256, 256, 0, 0);
  run<16>(256, 256, 0, 0);
  run<32>(256, 256, 0, 0);
  run_fresh<1>(256, 256);
  run_fresh<2>(256, 256);
  run_fresh<4>(256, 256);
  run_fresh<8>(256, 256);
  run_fresh<16>(256, 256);
  run<1>(256, 1024, 98304, 1);   // every launch carries a start and a stop event
  run<1>(256, 1024, 98304, 4);
  return 0;
}
