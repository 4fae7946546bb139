// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the histogram
// launches (attribute the child launches' traffic).  Each kernel reads a known
// number of bytes from a 2 GB buffer (far beyond the 256 MB Infinity Cache):
//   stream    16 B per lane, fully coalesced (the root launch's rows)
//   gather64  random 64-byte-aligned 64-byte rows, 4 lanes x 16 B per row
//   gather48  rows of 48 B at a 48-byte stride, ascending ids with a random gap (mean 4 rows):
//             the child launches' bin rows; 3 lanes x 16 B per row
//   gather8   8-byte words at ascending ids with a random gap (mean 4 words): lambda[id]
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o c --output-format csv -- ./fetch_calib
// and compare FETCH_SIZE (KB) of each kernel with the `useful` / `lines64` bytes printed here.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;

__global__ void k_stream(const uint4 *p, size_t n16, u64 *out) {
  u64 acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 0x123456789ull) out[0] = acc;
}
__global__ void k_gather64(const uint4 *p, const uint32_t *ids, size_t nrows, u64 *out) {
  u64 acc = 0;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (size_t r = t / 4; r < nrows; r += (size_t)gridDim.x * blockDim.x / 4) {
    const uint4 v = p[(size_t)ids[r] * 4 + (t & 3)];
    acc += v.x + v.w;
  }
  if (acc == 0x123456789ull) out[0] = acc;
}
__global__ void k_gather48(const uint8_t *p, const uint32_t *ids, size_t nrows, u64 *out) {
  u64 acc = 0;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t lanes = (size_t)gridDim.x * blockDim.x;
  for (size_t u = t; u < nrows * 3; u += lanes) {
    const size_t r = u / 3, c = u - r * 3;
    const uint4 v = *reinterpret_cast<const uint4 *>(p + (size_t)ids[r] * 48 + c * 16);
    acc += v.x + v.w;
  }
  if (acc == 0x123456789ull) out[0] = acc;
}
__global__ void k_gather8(const u64 *p, const uint32_t *ids, size_t n, u64 *out) {
  u64 acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += p[ids[i]];
  if (acc == 0x123456789ull) out[0] = acc;
}

int main() {
  const size_t BYTES = (size_t)2 << 30;
  uint8_t *buf; u64 *out; uint32_t *ids;
  hipMalloc(&buf, BYTES); hipMalloc(&out, 64); hipMemset(buf, 1, BYTES);
  const size_t NR = 4u << 20;  // rows gathered
  hipMalloc(&ids, NR * 4);
  std::vector<uint32_t> h(NR);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  // (a) random 64-byte rows over the whole buffer
  for (size_t i = 0; i < NR; ++i) h[i] = rnd() % (uint32_t)(BYTES / 64);
  hipMemcpy(ids, h.data(), NR * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const uint4 *)buf, BYTES / 16, out);
  hipLaunchKernelGGL(k_gather64, dim3(2048), dim3(256), 0, 0, (const uint4 *)buf, ids, NR, out);
  hipDeviceSynchronize();
  printf("k_stream   useful %zu bytes\n", BYTES);
  printf("k_gather64 useful %zu bytes (= lines64 %zu)\n", NR * 64, NR * 64);
  // (b) ascending ids, random gap 1..7 (mean 4): rows of 48 B
  {
    uint32_t id = 0;
    size_t lines = 0; long long last = -1;
    for (size_t i = 0; i < NR; ++i) {
      id += 1 + rnd() % 7;
      h[i] = id;
      const long long l0 = (long long)((size_t)id * 48 / 64), l1 = (long long)(((size_t)id * 48 + 47) / 64);
      for (long long l = l0; l <= l1; ++l) if (l > last) { ++lines; last = l; }
    }
    hipMemcpy(ids, h.data(), NR * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_gather48, dim3(2048), dim3(256), 0, 0, buf, ids, NR, out);
    hipDeviceSynchronize();
    printf("k_gather48 useful %zu bytes, distinct 64-B lines %zu = %zu bytes (+ ids %zu)\n", NR * 48, lines, lines * 64, NR * 4);
    size_t l8 = 0; last = -1;
    for (size_t i = 0; i < NR; ++i) { const long long l = (long long)((size_t)h[i] * 8 / 64); if (l > last) { ++l8; last = l; } }
    hipLaunchKernelGGL(k_gather8, dim3(2048), dim3(256), 0, 0, (const u64 *)buf, ids, NR, out);
    hipDeviceSynchronize();
    printf("k_gather8  useful %zu bytes, distinct 64-B lines %zu = %zu bytes (+ ids %zu)\n", NR * 8, l8, l8 * 64, NR * 4);
  }
  return 0;
}
