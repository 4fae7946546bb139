// lost_writes.hip -- NO product code: does this machine lose the stores of whole workgroups when
// several processes share the GPU and the host is oversubscribed?  (Round 6)
//
// The r06 hunt (profiles/r06_hunt.md) caught the intermittent mismatch of rounds 4-6 with the device's
// state read back at the first differing tree: the BIN MAP on the device was wrong in runs of eight
// documents -- exactly the 256 bytes one workgroup of k_binning writes (an element-wise kernel: one
// thread, one byte, no LDS, no inter-workgroup protocol) -- zeros where ~11 % of its workgroups should
// have stored, in a process that shared the GPU with seven others on a host running 128 threads on 16
// cores; the same hunt saw six `Memory access fault by GPU node` aborts.  Four processes with four
// threads each: 144,000 configurations clean.  This program repeats the shape of that launch chain
// with nothing of the product in it:
//   host rows (pageable) -> hipMemcpy -> k_map (a byte per (row, column): the first slot whose
//   threshold is not below the value; 256 threads, 256 bytes per workgroup) -> k_tile (64-row tiles
//   through LDS into a column-major copy) -> hipMemcpy back, both compared with the host's own result,
// on a stream of its own (hipStreamCreate: blocking, like the product's), buffers allocated and freed
// every iteration.  Run P copies side by side with busy host threads around them:
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/lost_writes scripts/ubench/lost_writes.hip
//   scripts/ubench/lost_writes ITERATIONS [SEED]
// Prints one line per damaged iteration (cells wrong, the runs they form) and a summary; exit 1 if any.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <string>
#include <random>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                                 \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

__global__ __launch_bounds__(256) void k_map(const float *__restrict__ raw, const uint32_t N, const uint32_t F,
                                             const float *__restrict__ thr, const uint32_t T, const uint32_t fw,
                                             uint8_t *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)N * fw) return;
  const uint32_t d = (uint32_t)(i / fw), c = (uint32_t)(i % fw);
  uint8_t o = 0;
  if (c < F) {
    const float x = raw[(size_t)d * F + c];
    const float *t = thr + (size_t)c * T;
    uint32_t lo = 0, hi = T;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (x <= t[mid])
        hi = mid;
      else
        lo = mid + 1;
    }
    o = (uint8_t)(1 + (lo >= T ? T - 1 : lo));   // (never 0: a byte nobody stored stays recognisable)
  }
  out[(size_t)d * fw + c] = o;
}

__global__ __launch_bounds__(256) void k_tile(const uint8_t *__restrict__ in, const uint32_t N, const uint32_t F,
                                              const uint32_t fw, uint8_t *__restrict__ cm) {
  __shared__ uint8_t tile[64][65];
  const uint32_t d0 = blockIdx.x * 64;
  for (uint32_t i = threadIdx.x; i < 64 * fw; i += 256) {
    const uint32_t dd = i / fw, cc = i % fw;
    if (cc < 64) tile[dd][cc] = d0 + dd < N ? in[(size_t)(d0 + dd) * fw + cc] : 0;
  }
  __syncthreads();
  const uint32_t fc = F < 64 ? F : 64;
  for (uint32_t i = threadIdx.x; i < 64 * fc; i += 256) {
    const uint32_t cc = i / 64, dd = i % 64;
    if (d0 + dd < N) cm[(size_t)cc * N + d0 + dd] = tile[dd][cc];
  }
}

// the same comparison on the device, in a launch of its own behind the two (a store that never reached
// memory is as missing to this kernel as to the copy back): lets an iteration cost milliseconds
__global__ __launch_bounds__(256) void k_check(const float *__restrict__ raw, const uint32_t N, const uint32_t F,
                                               const float *__restrict__ thr, const uint32_t T, const uint32_t fw,
                                               const uint8_t *__restrict__ out, const uint8_t *__restrict__ cm,
                                               unsigned long long *__restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)N * fw) return;
  const uint32_t d = (uint32_t)(i / fw), c = (uint32_t)(i % fw);
  if (c >= F) return;
  const float x = raw[(size_t)d * F + c];
  const float *t = thr + (size_t)c * T;
  uint32_t lo = 0, hi = T;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (x <= t[mid])
      hi = mid;
    else
      lo = mid + 1;
  }
  const uint8_t want = (uint8_t)(1 + (lo >= T ? T - 1 : lo));
  if (out[(size_t)d * fw + c] != want) {
    atomicAdd(&bad[0], 1ull);
    atomicAdd(&bad[2 + (i / 256) % 8], 1ull);
  }
  if (c < 64 && cm[(size_t)c * N + d] != want) {
    atomicAdd(&bad[1], 1ull);
    atomicAdd(&bad[10 + (d / 64) % 8], 1ull);
  }
}

// "fast" mode: scripts/ubench/lost_writes fast SECONDS SEED QUEUES -- the same chain checked by k_check on the
// device, rows taken from one host pool, QUEUES extra streams of different priorities kept alive and
// poked every iteration (the product's contexts hold up to five streams each: eight processes
// oversubscribe the hardware queues, four do not), one more stream created and destroyed per iteration.
static int run_fast(const double seconds, const unsigned seed, const int nq) {
  std::mt19937 rng(seed);
  const size_t POOL = (size_t)8 << 20;
  std::vector<float> pool(POOL + 100000 * 64);
  for (auto &v : pool) v = (float)(rng() % 100000) * 1e-5f;
  std::vector<hipStream_t> qs((size_t)nq);
  for (int i = 0; i < nq; ++i) CK(hipStreamCreateWithPriority(&qs[(size_t)i], hipStreamNonBlocking, -(i % 3)));
  unsigned long long *d_bad;
  float *d_poke;
  CK(hipMalloc(&d_bad, 18 * 8));
  CK(hipMalloc(&d_poke, 4096));
  long its = 0, bad_iters = 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const uint32_t N = 1000 + rng() % 100000, F = 5 + rng() % 60, T = 8 + rng() % 56;
    const uint32_t fw = (F + 15) / 16 * 16;
    const float *x = pool.data() + rng() % POOL;
    std::vector<float> thr((size_t)F * T);
    for (uint32_t f = 0; f < F; ++f)
      for (uint32_t t = 0; t < T; ++t) thr[(size_t)f * T + t] = (float)(t + 1) / (float)(T + 1);
    float *d_raw, *d_thr;
    uint8_t *d_out, *d_cm;
    CK(hipMalloc(&d_raw, (size_t)N * F * 4));
    CK(hipMalloc(&d_thr, thr.size() * 4));
    CK(hipMalloc(&d_out, (size_t)N * fw));
    CK(hipMalloc(&d_cm, (size_t)F * N));
    CK(hipMemcpy(d_raw, x, (size_t)N * F * 4, hipMemcpyDefault));
    CK(hipMemcpy(d_thr, thr.data(), thr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_bad, 0, 18 * 8));
    const unsigned g = (unsigned)(((size_t)N * fw + 255) / 256);
    hipLaunchKernelGGL(k_map, dim3(g), dim3(256), 0, st, d_raw, N, F, d_thr, T, fw, d_out);
    hipLaunchKernelGGL(k_tile, dim3((N + 63) / 64), dim3(256), 0, st, d_out, N, F, fw, d_cm);
    for (int i = 0; i < nq; ++i) CK(hipMemsetAsync(d_poke, 0, 64, qs[(size_t)i]));
    CK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(k_check, dim3(g), dim3(256), 0, st, d_raw, N, F, d_thr, T, fw, d_out, d_cm, d_bad);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    unsigned long long b[18];
    CK(hipMemcpy(b, d_bad, sizeof(b), hipMemcpyDeviceToHost));
    if (b[0] || b[1]) {
      ++bad_iters;
      printf("iteration %ld (N %u F %u fw %u T %u): row-major copy wrong in %llu cells, column-major copy in %llu; by (workgroup index %% 8): "
             "k_map [%llu %llu %llu %llu %llu %llu %llu %llu] k_tile [%llu %llu %llu %llu %llu %llu %llu %llu]\n", its, N, F, fw, T, b[0], b[1],
             b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11], b[12], b[13], b[14], b[15], b[16], b[17]);
      fflush(stdout);
    }
    for (int i = 0; i < nq; ++i) CK(hipStreamSynchronize(qs[(size_t)i]));
    CK(hipFree(d_raw));
    CK(hipFree(d_thr));
    CK(hipFree(d_out));
    CK(hipFree(d_cm));
    CK(hipStreamDestroy(st));
    ++its;
  }
  printf("fast mode: %ld iterations in %.0f s with %d extra queues, %ld damaged\n", its, seconds, nq, bad_iters);
  return bad_iters ? 1 : 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "fast")
    return run_fast(argc > 2 ? atof(argv[2]) : 60.0, argc > 3 ? (unsigned)atoi(argv[3]) : 1u, argc > 4 ? atoi(argv[4]) : 5);
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
  std::mt19937 rng(seed);
  long bad_iters = 0, bad_cells = 0;
  for (int it = 0; it < iters; ++it) {
    // a stream of its own per iteration, and an auxiliary one with a priority, created and destroyed
    // like a context of the product: every creation / destruction changes the set of hardware queues
    // the scheduler maps, for every process on the device
    hipStream_t st, aux;
    CK(hipStreamCreate(&st));
    CK(hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, 0));
    const uint32_t N = 1000 + rng() % 100000, F = 5 + rng() % 60, T = 8 + rng() % 56;
    const uint32_t fw = (F + 15) / 16 * 16;
    std::vector<float> x((size_t)N * F), thr((size_t)F * T);
    for (auto &v : x) v = (float)(rng() % 100000) * 1e-5f;
    for (uint32_t f = 0; f < F; ++f)
      for (uint32_t t = 0; t < T; ++t) thr[(size_t)f * T + t] = (float)(t + 1) / (float)(T + 1);
    float *d_raw, *d_thr;
    uint8_t *d_out, *d_cm;
    CK(hipMalloc(&d_raw, x.size() * 4));
    CK(hipMalloc(&d_thr, thr.size() * 4));
    CK(hipMalloc(&d_out, (size_t)N * fw));
    CK(hipMalloc(&d_cm, (size_t)F * N));
    CK(hipMemcpy(d_raw, x.data(), x.size() * 4, hipMemcpyDefault));
    CK(hipMemcpy(d_thr, thr.data(), thr.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_map, dim3((unsigned)(((size_t)N * fw + 255) / 256)), dim3(256), 0, st, d_raw, N, F, d_thr, T, fw, d_out);
    hipLaunchKernelGGL(k_tile, dim3((N + 63) / 64), dim3(256), 0, st, d_out, N, F, fw, d_cm);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    CK(hipMemsetAsync(d_raw, 0, 16, aux));   // (the auxiliary queue has something to do)
    CK(hipStreamSynchronize(aux));
    std::vector<uint8_t> out((size_t)N * fw), cm((size_t)F * N);
    CK(hipMemcpy(out.data(), d_out, out.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(cm.data(), d_cm, cm.size(), hipMemcpyDeviceToHost));
    long w1 = 0, w2 = 0, zero1 = 0, first = -1, runs = 0, prev = -2;
    long by_xcd1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, by_xcd2[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // wrong cells by (workgroup index % 8)
    for (uint32_t d = 0; d < N; ++d) {
      bool row_bad = false;
      for (uint32_t c = 0; c < F; ++c) {
        const float v = x[(size_t)d * F + c];
        const float *t = &thr[(size_t)c * T];
        const uint32_t lo = (uint32_t)(std::lower_bound(t, t + T, v) - t);
        const uint8_t want = (uint8_t)(1 + (lo >= T ? T - 1 : lo));
        const uint8_t g1 = out[(size_t)d * fw + c];
        if (g1 != want) {
          ++w1;
          zero1 += g1 == 0;
          row_bad = true;
          ++by_xcd1[(((size_t)d * fw + c) / 256) % 8];
        }
        if (c < 64 && cm[(size_t)c * N + d] != want) {
          ++w2;
          ++by_xcd2[(d / 64) % 8];
        }
      }
      if (row_bad) {
        if (first < 0) first = d;
        if ((long)d != prev + 1) ++runs;
        prev = d;
      }
    }
    if (w1 || w2) {
      ++bad_iters;
      bad_cells += w1 + w2;
      printf("iteration %d (N %u F %u fw %u T %u): row-major copy wrong in %ld cells (%ld of them zero = never stored) in %ld runs of rows "
             "starting at row %ld; column-major copy wrong in %ld cells\n", it, N, F, fw, T, w1, zero1, runs, first, w2);
      printf("   wrong cells by (workgroup index %% 8) -- workgroups go round the eight XCDs in turn: k_map [%ld %ld %ld %ld %ld %ld %ld %ld]  "
             "k_tile [%ld %ld %ld %ld %ld %ld %ld %ld]\n", by_xcd1[0], by_xcd1[1], by_xcd1[2], by_xcd1[3], by_xcd1[4], by_xcd1[5], by_xcd1[6],
             by_xcd1[7], by_xcd2[0], by_xcd2[1], by_xcd2[2], by_xcd2[3], by_xcd2[4], by_xcd2[5], by_xcd2[6], by_xcd2[7]);
      fflush(stdout);
    }
    CK(hipFree(d_raw));
    CK(hipFree(d_thr));
    CK(hipFree(d_out));
    CK(hipFree(d_cm));
    CK(hipStreamDestroy(aux));
    CK(hipStreamDestroy(st));
  }
  printf("%d iterations, %ld damaged (%ld cells)\n", iters, bad_iters, bad_cells);
  return bad_iters ? 1 : 0;
}
