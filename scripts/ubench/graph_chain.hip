// Microbenchmark: a chain of dependent launches enqueued on a stream against the same chain
// captured into a hipGraph and launched as one (does a graph shorten a link of the per-step
// chain on the MI355X?).  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -w -o scripts/ubench/graph_chain scripts/ubench/graph_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int LINKS>
__global__ void k_chain(const uint32_t *__restrict__ a, uint32_t *__restrict__ out) {
  uint32_t x = blockIdx.x * 64u;
#pragma unroll
  for (int i = 0; i < LINKS; ++i) x = a[x];
  if (x == 0xdeadbeefu) out[threadIdx.x] = x;
}

__global__ void k_block(uint32_t *out, long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (ticks < 0) out[0] = 1;
}

static uint32_t *d_a, *d_out;

template <int LINKS>
static void run(int grid, int block, int nodes) {
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 20;   // graph launches / stream repetitions of the same chain
  float ms_stream = 0, ms_graph = 0;
  for (int w = 0; w < 2; ++w) {
    hipLaunchKernelGGL(k_block, dim3(1), dim3(64), 0, st, d_out, (long long)(reps * nodes * 6 * 100));
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r)
      for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_chain<LINKS>, dim3(grid), dim3(block), 0, st, d_a, d_out);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    hipEventElapsedTime(&ms_stream, e0, e1);
  }
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_chain<LINKS>, dim3(grid), dim3(block), 0, st, d_a, d_out);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int w = 0; w < 2; ++w) {
    hipLaunchKernelGGL(k_block, dim3(1), dim3(64), 0, st, d_out, (long long)(reps * nodes * 6 * 100));
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    hipEventElapsedTime(&ms_graph, e0, e1);
  }
  printf("links %d grid %4d block %4d, %3d launches per chain: stream %.2f us per launch, graph %.2f us per launch\n",
         LINKS, grid, block, nodes, ms_stream * 1e3 / (reps * nodes), ms_graph * 1e3 / (reps * nodes));
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  hipStreamDestroy(st);
}

int main() {
  const size_t n = 1u << 20;
  std::vector<uint32_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)((i * 64u + 4096u * 17u) % n);
  hipMalloc(&d_a, n * 4);
  hipMalloc(&d_out, 4096);
  hipMemcpy(d_a, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int nodes : {26, 100})
    for (int grid : {1, 256, 2048}) {
      run<0>(grid, 256, nodes);
      run<1>(grid, 256, nodes);
      run<3>(grid, 1024, nodes);
    }
  return 0;
}
