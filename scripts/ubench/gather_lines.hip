// gather_lines.hip -- what a gather of 48-byte rows in ascending document order costs on this chip,
// WITHOUT the histogram: the ceiling of k_hist_batch's memory side.
//
// The child launches of the tree growth read, per document of the node's list: its id (4 B,
// coalesced), one 16-byte piece per lane of its 48-byte row in one feature block (3 lanes per
// document) and its 8-byte pseudo-response.  This kernel issues exactly those requests -- same
// workgroup shape (1024 threads), same lane mapping (lane = (document, 16-byte chunk), 21 documents
// per wave step), NS tiles in flight per lane -- and only XORs what arrives.  Lists are ascending
// random subsets of N documents at density p (the root's smaller child ~0.4-0.5, deeper nodes less).
// Printed per (p, row pitch 48 | 64, with / without the 8-byte gather): us, documents / us, useful
// GB/s and the 64-byte lines the requests touch per second.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/gather_lines scripts/ubench/gather_lines.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NS, int PITCH, bool LAM>
__global__ __launch_bounds__(1024) void k_gather(const uint8_t *__restrict__ rows, const double *__restrict__ lam,
                                                 const uint32_t *__restrict__ ids, const uint32_t n,
                                                 unsigned long long *__restrict__ out) {
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t r0 = blockIdx.x * per, r1 = min(n, r0 + per);
  if (r0 >= r1) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dsub = lane / 3, c = lane - 3 * dsub;
  const bool ok = dsub < 21;
  const uint32_t step = 16 * 21, last = r1 - 1;
  unsigned long long acc = 0;
  uint32_t id[NS];
  uint4 row[NS];
  double lv[NS];
  const uint32_t p0 = r0 + wave * 21 + dsub;
#pragma unroll
  for (int i = 0; i < NS; ++i) id[i] = ids[min(p0 + i * step, last)];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    row[i] = *reinterpret_cast<const uint4 *>(rows + (size_t)id[i] * PITCH + 16 * c);
    if (LAM) lv[i] = lam[id[i]];
    id[i] = ids[min(p0 + (NS + i) * step, last)];
  }
  for (uint32_t pos = p0; pos < r1 + step; pos += NS * step) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (ok && pos + i * step < r1) {
        acc ^= row[i].x ^ ((unsigned long long)row[i].y << 7) ^ row[i].z ^ ((unsigned long long)row[i].w << 9);
        if (LAM) acc += (unsigned long long)__double_as_longlong(lv[i]);
      }
      row[i] = *reinterpret_cast<const uint4 *>(rows + (size_t)id[i] * PITCH + 16 * c);
      if (LAM) lv[i] = lam[id[i]];
      id[i] = ids[min(pos + (2 * NS + i) * step, last)];
    }
  }
  if (acc == 0x1234567ull) out[blockIdx.x] = acc;  // (never: keeps the loads alive)
}

template <int NS, int PITCH, bool LAM>
static float run(const uint8_t *rows, const double *lam, const uint32_t *ids, uint32_t n, unsigned long long *out, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<NS, PITCH, LAM>), dim3(grid), dim3(1024), 0, 0, rows, lam, ids, n, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep) best = std::min(best, ms);
  }
  return best * 1000.f;
}

int main(int argc, char **argv) {
  const uint32_t N = argc > 1 ? (uint32_t)atol(argv[1]) : 8000000u;
  uint8_t *rows; double *lam; uint32_t *ids; unsigned long long *out;
  CK(hipMalloc(&rows, (size_t)N * 64 + 64));
  CK(hipMalloc(&lam, (size_t)N * 8));
  CK(hipMalloc(&ids, (size_t)N * 4));
  CK(hipMalloc(&out, 4096 * 8));
  CK(hipMemset(rows, 1, (size_t)N * 64 + 64));
  CK(hipMemset(lam, 0, (size_t)N * 8));
  printf("N = %u documents; 256 / 768 workgroups of 1024 threads; 64-byte lines counted on the host\n", N);
  printf("| density | docs | pitch | +lambda | NS | grid | us | docs/us | useful GB/s | lines GB/s |\n|---|---|---|---|---|---|---|---|---|---|\n");
  std::mt19937_64 rng(7);
  for (double p : {1.0, 0.5, 0.25, 0.1, 0.03, 0.01}) {
    std::vector<uint32_t> h;
    h.reserve((size_t)(N * p * 1.1) + 16);
    std::bernoulli_distribution bd(p);
    for (uint32_t i = 0; i < N; ++i)
      if (p >= 1.0 || bd(rng)) h.push_back(i);
    const uint32_t n = (uint32_t)h.size();
    CK(hipMemcpy(ids, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    // lines touched by the row pieces (pitch 48 / 64) and by the 8-byte gather
    auto lines_rows = [&](int pitch) {
      size_t cnt = 0; long long prev = -1;
      for (uint32_t i : h) {
        const long long a = (long long)i * pitch / 64, b = ((long long)i * pitch + 47) / 64;
        for (long long l = std::max(a, prev + 1); l <= b; ++l) ++cnt;
        prev = b;
      }
      return cnt;
    };
    size_t ll = 0; { long long prev = -1; for (uint32_t i : h) { const long long l = i / 8; if (l != prev) ++ll; prev = l; } }
    const size_t l48 = lines_rows(48), l64 = lines_rows(64);
    for (int grid : {256, 768}) {
      struct { int pitch; bool lam; int ns; float us; } v[] = {
          {48, false, 3, run<3, 48, false>(rows, lam, ids, n, out, grid)}, {48, true, 3, run<3, 48, true>(rows, lam, ids, n, out, grid)},
          {48, true, 6, run<6, 48, true>(rows, lam, ids, n, out, grid)},   {64, true, 3, run<3, 64, true>(rows, lam, ids, n, out, grid)},
          {64, true, 6, run<6, 64, true>(rows, lam, ids, n, out, grid)}};
      for (auto &x : v) {
        const double useful = (double)n * (48 + 4 + (x.lam ? 8 : 0));
        const double lines = 64.0 * ((x.pitch == 48 ? l48 : l64) + (x.lam ? ll : 0) + (size_t)n / 16);
        printf("| %.2f | %u | %d | %s | %d | %d | %.1f | %.1f | %.0f | %.0f |\n", p, n, x.pitch, x.lam ? "yes" : "no", x.ns, grid, x.us,
               n / x.us, useful / x.us * 1e-3, lines / x.us * 1e-3);
      }
    }
  }
  return 0;
}
