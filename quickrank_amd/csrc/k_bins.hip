// k_bins.hip -- one-time data preparation kernels (gfx950).
//
// Stands behind Mart::init (mart.cc:117-176) and the RTRootHistogram ctor
// (rtnode_histogram.cc:227-253): threshold candidates per feature and the bin
// map.  The reference argsorts every column (radix.cc:35-73) only to obtain
// (a) the distinct values in ascending order, capped at nthresholds+1, and
// (b) min/max.  Here each column is streamed once by one workgroup that keeps
// an LDS hash set of distinct bit patterns (early-out once it overflows) plus
// min/max of the radix key; the host finishes the (tiny) threshold arithmetic
// with the exact f32 operations of mart.cc:147-169.  The bin map is stored as
// uint8 in 64-feature blocks, row-major inside a block ([block][doc][fw]), the
// layout the histogram kernels stream and gather (DESIGN.md "HBM layout").
#include <cstdio>
#include <cstring>

#include "qr_internal.h"

// ---------------------------------------------------------------------------
// [N][F] row-major -> [F][N] column-major (vertical_dataset.cc:45-51)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose(const float *__restrict__ in,
                                                   float *__restrict__ out,
                                                   uint32_t N, uint32_t F) {
  __shared__ float tile[32][33];
  const uint32_t bx = blockIdx.x * 32;  // feature tile
  const uint32_t by = blockIdx.y * 32;  // doc tile
  const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (uint32_t j = ty; j < 32; j += 8) {
    const uint32_t d = by + j, f = bx + tx;
    if (d < N && f < F) tile[j][tx] = in[(size_t)d * F + f];
  }
  __syncthreads();
  for (uint32_t j = ty; j < 32; j += 8) {
    const uint32_t f = bx + j, d = by + tx;
    if (d < N && f < F) out[(size_t)f * N + d] = tile[tx][j];
  }
}

int qr_k_transpose(qr_ctx *c, const float *raw, float *col, size_t N, size_t F) {
  dim3 grid((unsigned)((F + 31) / 32), (unsigned)((N + 31) / 32));
  hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, c->stream, raw, col,
                     (uint32_t)N, (uint32_t)F);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// ---------------------------------------------------------------------------
// Distinct values (as bit patterns) of one column, capped at `limit`, and the
// min/max of the radix key (radix.cc:28-30: sign-flip so unsigned order ==
// float order with -0.0 < +0.0).
// ---------------------------------------------------------------------------
#define QR_HASH_SLOTS 2048u
#define QR_HASH_EMPTY 0xFFFFFFFFu

__device__ __forceinline__ uint32_t flip_key(uint32_t x) {
  return x ^ ((uint32_t)(-(int32_t)(x >> 31)) | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_colstats(const float *__restrict__ col,
                                                  uint32_t N, uint32_t limit,
                                                  uint32_t *__restrict__ vals,
                                                  uint32_t *__restrict__ cnt,
                                                  uint32_t *__restrict__ minmax) {
  __shared__ uint32_t table[QR_HASH_SLOTS];
  __shared__ uint32_t s_count, s_min, s_max;
  const uint32_t f = blockIdx.x;
  const uint32_t *x = reinterpret_cast<const uint32_t *>(col) + (size_t)f * N;
  for (uint32_t i = threadIdx.x; i < QR_HASH_SLOTS; i += blockDim.x)
    table[i] = QR_HASH_EMPTY;
  if (threadIdx.x == 0) {
    s_count = 0;
    s_min = 0xFFFFFFFFu;
    s_max = 0;
  }
  __syncthreads();
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) {
    const uint32_t v = x[i];
    const uint32_t k = flip_key(v);
    kmin = k < kmin ? k : kmin;
    kmax = k > kmax ? k : kmax;
    // the set saturates at limit+1 entries: beyond that only min/max matter
    if (*(volatile uint32_t *)&s_count <= limit) {
      uint32_t h = (v * 2654435761u) >> 21;  // 11 bits
      for (uint32_t probe = 0; probe < QR_HASH_SLOTS; ++probe) {
        const uint32_t cur = table[h];
        if (cur == v) break;
        if (cur == QR_HASH_EMPTY) {
          const uint32_t old = atomicCAS(&table[h], QR_HASH_EMPTY, v);
          if (old == QR_HASH_EMPTY) {
            atomicAdd(&s_count, 1u);
            break;
          }
          if (old == v) break;
        }
        h = (h + 1) & (QR_HASH_SLOTS - 1);
        if (*(volatile uint32_t *)&s_count > limit) break;
      }
    }
  }
  atomicMin(&s_min, kmin);
  atomicMax(&s_max, kmax);
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt[f] = s_count;
    minmax[2 * f] = s_min;
    minmax[2 * f + 1] = s_max;
  }
  // compact the (<= limit+1 + racing inserts) distinct patterns
  __shared__ uint32_t s_out;
  if (threadIdx.x == 0) s_out = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < QR_HASH_SLOTS; i += blockDim.x) {
    const uint32_t v = table[i];
    if (v != QR_HASH_EMPTY) {
      const uint32_t o = atomicAdd(&s_out, 1u);
      if (o < limit + 1) vals[(size_t)f * (limit + 1) + o] = v;
    }
  }
}

int qr_k_colstats(qr_ctx *c, const float *col, size_t N, size_t F,
                  uint32_t limit, uint32_t *d_vals, uint32_t *d_cnt,
                  uint32_t *d_minmax) {
  hipLaunchKernelGGL(k_colstats, dim3((unsigned)F), dim3(256), 0, c->stream, col,
                     (uint32_t)N, limit, d_vals, d_cnt, d_minmax);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// ---------------------------------------------------------------------------
// Bin map: bin = first slot t with x <= thr[f][t] (rtnode_histogram.cc:241-251)
// = lower_bound over the non-decreasing threshold row.  One thread per
// (doc, local column); columns fastest so both the f32 row read and the u8
// block-row write are coalesced.
// ---------------------------------------------------------------------------
static int qr_k_binning_blocks(qr_ctx *c);

__global__ __launch_bounds__(256) void k_binning(const float *__restrict__ raw,
                                                 uint32_t N, uint32_t F,
                                                 const float *__restrict__ thr,
                                                 const uint32_t *__restrict__ thr_size,
                                                 QrBlock blk,
                                                 uint8_t *__restrict__ bins) {
  const uint32_t fw = (uint32_t)blk.fw;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * fw;
  if (idx >= total) return;
  const uint32_t d = (uint32_t)(idx / fw);
  const uint32_t cidx = (uint32_t)(idx % fw);
  uint8_t out = 0;  // padded columns: constant bin 0 (never a split candidate)
  if (cidx < (uint32_t)blk.nreal) {
    const uint32_t f = blk.f0 + cidx;
    const float x = raw[(size_t)d * F + f];
    const float *t = thr + (size_t)f * QR_MAX_BINS;
    uint32_t lo = 0, hi = thr_size[f];  // first lo with x <= t[lo]
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (x <= t[mid])
        hi = mid;
      else
        lo = mid + 1;
    }
    const uint32_t last = thr_size[f] - 1;
    out = (uint8_t)(lo > last ? last : lo);  // NaN/+inf -> sentinel slot
  }
  bins[blk.off + (size_t)d * fw + cidx] = out;
}

// feature-major copy of one block, [lf][doc]: 64 docs x fw columns per workgroup
// through LDS so both sides are coalesced
__global__ __launch_bounds__(256) void k_bins_fm(const uint8_t *__restrict__ bins, uint32_t N,
                                                 QrBlock blk, uint8_t *__restrict__ fm) {
  __shared__ uint8_t tile[64][65];
  const uint32_t fw = (uint32_t)blk.fw;
  const uint32_t d0 = blockIdx.x * 64;
  for (uint32_t i = threadIdx.x; i < 64 * fw; i += 256) {
    const uint32_t dd = i / fw, cc = i % fw;
    tile[dd][cc] = d0 + dd < N ? bins[blk.off + (size_t)(d0 + dd) * fw + cc] : 0;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 64 * (uint32_t)blk.nreal; i += 256) {
    const uint32_t cc = i / 64, dd = i % 64;
    if (d0 + dd < N) fm[(size_t)(blk.lf0 + cc) * N + d0 + dd] = tile[dd][cc];
  }
}

// Verify-after-write of the resident bin map (round 6).  The bin map and its feature-major copy are
// written ONCE and read by every launch of every tree; the r06 hunt (profiles/r06_hunt.md) saw
// this platform drop the stores of one workgroup in eight of exactly these two launches when eight
// processes shared the GPU -- zeros in 12.5 % of the map, a wrong model from then on, silently.  So
// the map is checked once behind its kernels: every cell recomputed from the raw rows and compared
// with BOTH copies, by a workgroup that is NOT the one that stored it (the index is rotated by three
// workgroups, so another XCD's L2 / TLB reads it back).  One more pass over the raw matrix per data set
// (0.9 ms per million documents x 136 features, next to 18 ms of upload).  out[0] / out[1]: cells of the
// block rows / of the feature-major copy that do not hold what the binning computes.
__global__ __launch_bounds__(256) void k_bins_verify(const float *__restrict__ raw, uint32_t N, uint32_t F,
                                                     const float *__restrict__ thr,
                                                     const uint32_t *__restrict__ thr_size, QrBlock blk,
                                                     const uint8_t *__restrict__ bins,
                                                     const uint8_t *__restrict__ fm,
                                                     unsigned long long *__restrict__ out) {
  const uint32_t fw = (uint32_t)blk.fw;
  const unsigned g = (blockIdx.x + 3u) % gridDim.x;
  const size_t idx = (size_t)g * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * fw) return;
  const uint32_t d = (uint32_t)(idx / fw), cidx = (uint32_t)(idx % fw);
  uint8_t want = 0;
  const bool real = cidx < (uint32_t)blk.nreal;
  if (real) {
    const uint32_t f = blk.f0 + cidx;
    const float x = raw[(size_t)d * F + f];
    const float *t = thr + (size_t)f * QR_MAX_BINS;
    uint32_t lo = 0, hi = thr_size[f];
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (x <= t[mid])
        hi = mid;
      else
        lo = mid + 1;
    }
    const uint32_t last = thr_size[f] - 1;
    want = (uint8_t)(lo > last ? last : lo);
  }
  // out[2..9] / out[10..17]: the same by the index mod 8 of the workgroup that STORED the cell (workgroups of a
  // launch go round the eight XCDs: the hunt's events are one residue each)
  if (bins[blk.off + (size_t)d * fw + cidx] != want) {
    atomicAdd(&out[0], 1ull);
    atomicAdd(&out[2 + g % 8u], 1ull);
  }
  if (real && fm[(size_t)(blk.lf0 + cidx) * N + d] != want) {
    atomicAdd(&out[1], 1ull);
    atomicAdd(&out[10 + (d / 64u) % 8u], 1ull);
  }
}

int qr_k_bins_verify(qr_ctx *c, unsigned long long *bad_rows, unsigned long long *bad_fm, unsigned long long *by_wg) {
  unsigned long long *d_out = nullptr;
  QR_CHECK(c, hipMalloc((void **)&d_out, 18 * 8));
  QR_CHECK(c, hipMemsetAsync(d_out, 0, 18 * 8, c->stream));
  for (int b = 0; b < c->nblocks; ++b) {
    const QrBlock &blk = c->blocks[b];
    const unsigned grid = (unsigned)((c->N * (size_t)blk.fw + 255) / 256);
    if (!grid) continue;
    hipLaunchKernelGGL(k_bins_verify, dim3(grid), dim3(256), 0, c->stream, c->d_raw, (uint32_t)c->N, (uint32_t)c->F,
                       c->d_thr, c->d_thr_size, blk, (const uint8_t *)c->d_bins, (const uint8_t *)c->d_bins_fm, d_out);
  }
  unsigned long long h[18] = {};
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d_out);
  QR_CHECK(c, e);
  *bad_rows = h[0];
  *bad_fm = h[1];
  if (by_wg) memcpy(by_wg, h + 2, 16 * 8);
  return QR_OK;
}

static int binning_launches(qr_ctx *c) {
  const int rc = qr_k_binning_blocks(c);
  if (rc) return rc;
  for (int b = 0; b < c->nblocks; ++b) {
    const QrBlock &blk = c->blocks[b];
    hipLaunchKernelGGL(k_bins_fm, dim3((unsigned)((c->N + 63) / 64)), dim3(256), 0, c->stream,
                       c->d_bins, (uint32_t)c->N, blk, c->d_bins_fm);
    QR_CHECK(c, hipGetLastError());
  }
  return QR_OK;
}

int qr_k_binning(qr_ctx *c) {
  // (the raw rows are on the device when the map is made from them: contexts that were handed a
  // map's thresholds only -- qr_bins_build_with -- hold them too)
  for (int attempt = 0;; ++attempt) {
    int rc = binning_launches(c);
    if (rc) return rc;
    if (c->debug_lose_binning > 0 && c->N >= 16) {  // (test aid: what a workgroup's lost stores look like)
      --c->debug_lose_binning;
      QR_CHECK(c, hipStreamSynchronize(c->stream));
      QR_CHECK(c, hipMemset(c->d_bins + c->blocks[0].off + 8 * (size_t)c->blocks[0].fw, 0xFF, 8 * (size_t)c->blocks[0].fw));
    }
    unsigned long long bad_rows = 0, bad_fm = 0, by_wg[16];
    if ((rc = qr_k_bins_verify(c, &bad_rows, &bad_fm, by_wg))) return rc;
    if (!bad_rows && !bad_fm) {
      c->bins_rebuilt = attempt;
      return QR_OK;
    }
    char where[400];
    int w = 0;
    for (int i = 0; i < 16; ++i)
      w += snprintf(where + w, sizeof(where) - (size_t)w, "%s%llu", i == 8 ? " | " : i ? " " : "", by_wg[i]);
    fprintf(stderr, "qr: the bin map on the device does not hold what the binning kernel stored (%llu cells of the block "
                    "rows, %llu of the feature-major copy, attempt %d; by the storing workgroup's index mod 8: %s): device "
                    "memory lost stores -- %s\n",
            bad_rows, bad_fm, attempt + 1, where, attempt < 2 ? "building it again at another address" : "giving up");
    if (attempt == 2) {
      c->err = "the bin map on the device does not hold what the binning kernel stored, three times in a row: "
               "device memory loses stores (profiles/r06_hunt.md)";
      return QR_ERR_HIP;
    }
    // Call 12 of the r06 hunt: a rebuild IN PLACE lost the same cells three times over (one XCD's view of those
    // addresses stays wrong for the life of the mapping).  So the next attempt gets other addresses: the new
    // buffers are allocated while the old ones still hold theirs, then the old ones go.
    uint8_t *nb = nullptr, *nf = nullptr;
    QR_CHECK(c, hipMalloc((void **)&nb, c->bins_bytes ? c->bins_bytes : 1));
    if (const hipError_t e = hipMalloc((void **)&nf, (size_t)c->flocal * c->N ? (size_t)c->flocal * c->N : 1)) {
      (void)hipFree(nb);
      QR_CHECK(c, e);
    }
    (void)hipFree(c->d_bins);
    (void)hipFree(c->d_bins_fm);
    c->d_bins = nb;
    c->d_bins_fm = nf;
  }
}

static int qr_k_binning_blocks(qr_ctx *c) {
  for (int b = 0; b < c->nblocks; ++b) {
    const QrBlock &blk = c->blocks[b];
    const size_t total = c->N * (size_t)blk.fw;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_binning, dim3(grid), dim3(256), 0, c->stream, c->d_raw,
                       (uint32_t)c->N, (uint32_t)c->F, c->d_thr, c->d_thr_size,
                       blk, c->d_bins);
    QR_CHECK(c, hipGetLastError());
  }
  return QR_OK;
}
