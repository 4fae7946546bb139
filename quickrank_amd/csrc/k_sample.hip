// k_sample.hip -- --subsample (mart.cc:287-329): every boosting iteration fits its
// tree on a fresh uniform sample (without replacement) of k training documents.
//
// The reference shuffles an id array with a clock-seeded engine and keeps its
// first k entries.  Here the sample of iteration t is the k documents with the
// smallest keys hash(seed, t, doc): a device radix sort of (key, doc) pairs picks
// them, a flag array marks them (the lambda kernel "cleans" its queries with it,
// lambdamart.cc:85-102) and a stream compaction writes them in ascending order
// into the document list the root node is built from.  Same distribution,
// reproducible, and no host round trip.
#include <hipcub/hipcub.hpp>

#include "qr_internal.h"

namespace {

// (document-sharded ranks: N = the documents of ALL ranks, `present` covers the rank's own
// Nloc -- the key of a document is a function of its GLOBAL index, so every rank sorts the
// same keys and finds the same sample, of which it marks its own part: no exchange)
__global__ __launch_bounds__(256) void k_sample_keys(uint32_t *__restrict__ keys,
                                                     uint32_t *__restrict__ ids,
                                                     uint8_t *__restrict__ present,
                                                     const uint32_t N, const unsigned long long seed,
                                                     const uint32_t Nloc) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < Nloc) present[i] = 0;
  if (i >= N) return;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  keys[i] = (uint32_t)(z >> 32);
  ids[i] = i;
}

// the radix sort is stable: equal keys stay in ascending document order, so the
// first k pairs are a well-defined set
__global__ __launch_bounds__(256) void k_sample_mark(const uint32_t *__restrict__ sorted_ids,
                                                     uint8_t *__restrict__ present, const uint32_t k,
                                                     const uint32_t first, const uint32_t Nloc) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= k) return;
  const uint32_t d = sorted_ids[j] - first;  // (unsigned: documents before `first` wrap far beyond Nloc)
  if (d < Nloc) present[d] = 1;
}

// per-slice (sum of squares, sum) of the pseudo-responses over the sample: the
// root node's statistics (rtnode_histogram.cc:199-203 runs over the sample ids)
__global__ __launch_bounds__(256) void k_sample_sums(const uint32_t *__restrict__ list,
                                                     const uint32_t k,
                                                     const double *__restrict__ lambda,
                                                     double *__restrict__ ssq) {
  __shared__ double a[256], b[256];
  const uint32_t base = blockIdx.x * QR_SLICE;
  double sq = 0.0, sm = 0.0;
  for (uint32_t j = base + threadIdx.x; j < base + QR_SLICE && j < k; j += 256) {
    const double l = lambda[list[j]];
    sq += l * l;
    sm += l;
  }
  a[threadIdx.x] = sq;
  b[threadIdx.x] = sm;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      a[threadIdx.x] += a[threadIdx.x + off];
      b[threadIdx.x] += b[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ssq[2 * blockIdx.x] = a[0];
    ssq[2 * blockIdx.x + 1] = b[0];
  }
}

}  // namespace

size_t qr_k_sample_temp_bytes(size_t N) {
  size_t a = 0, b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)N);
  (void)hipcub::DeviceSelect::Flagged(nullptr, b, (const uint32_t *)nullptr, (const uint8_t *)nullptr,
                                      (uint32_t *)nullptr, (uint32_t *)nullptr, (int)N);
  return a > b ? a : b;
}

// draws the sample of the next iteration: c->d_present, and the ascending list in
// c->d_order[0][0 .. sub_k)
int qr_k_sample_draw(qr_ctx *c) {
  // N: the documents the sample is drawn from (all ranks' on a document-sharded context,
  // whose own documents are [sub_first, sub_first + Nloc) of them)
  const uint32_t Nloc = (uint32_t)c->N;
  const uint32_t N = c->dmode ? (uint32_t)c->Nglobal : Nloc;
  const uint32_t first = c->dmode ? (uint32_t)c->sub_first : 0u;
  const unsigned grid = (std::max(N, Nloc) + 255) / 256;
  c->sub_iter += 0x9E3779B97F4A7C15ull;
  uint32_t *keys = c->d_sample_keys, *ids = keys + N, *keys2 = ids + N, *ids2 = keys2 + N;
  hipLaunchKernelGGL(k_sample_keys, dim3(grid), dim3(256), 0, c->stream, keys, ids, c->d_present, N,
                     (unsigned long long)(c->sub_seed + c->sub_iter), Nloc);
  QR_CHECK(c, hipGetLastError());
  size_t tb = c->sample_temp_bytes;
  QR_CHECK(c, hipcub::DeviceRadixSort::SortPairs(c->d_sample_temp, tb, keys, keys2, ids, ids2, (int)N, 0,
                                                 32, c->stream));
  hipLaunchKernelGGL(k_sample_mark, dim3((unsigned)((c->sub_k + 255) / 256)), dim3(256), 0, c->stream,
                     ids2, c->d_present, (uint32_t)c->sub_k, first, Nloc);
  QR_CHECK(c, hipGetLastError());
  tb = c->sample_temp_bytes;
  // ids[] still holds 0, 1, ...: compaction by flag = the rank's part of the sample, ascending
  QR_CHECK(c, hipcub::DeviceSelect::Flagged(c->d_sample_temp, tb, ids, c->d_present, c->d_order[0],
                                            c->d_sample_count, (int)Nloc, c->stream));
  c->sub_n = c->sub_k;
  if (c->dmode) {  // how many of the sample's documents are this rank's: the host sizes launches by it
    uint32_t cnt = 0;
    QR_D2H(c, &cnt, c->d_sample_count, 4);
    c->sub_n = cnt;
  }
  return QR_OK;
}

int qr_k_sample_sums(qr_ctx *c) {
  const unsigned grid = (unsigned)std::max<size_t>(1, (c->sub_n + QR_SLICE - 1) / QR_SLICE);
  hipLaunchKernelGGL(k_sample_sums, dim3(grid), dim3(256), 0, c->stream, c->d_order[0],
                     (uint32_t)c->sub_n, c->d_lambda, c->d_ssq);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}
