// k_wide.hip -- more than 255 thresholds per feature (gfx950).
//
// QuickRank's default is `--num-thresholds 0`: every distinct value of a feature is a
// threshold candidate (quicklearn.cc:103, mart.cc:147-158) -- tens of thousands of
// slots on a real-valued column -- and any `--num-thresholds` above 255 gives the
// equal-width branch (mart.cc:159-169) that many.  The u8 bins and 256-slot LDS
// histograms of k_tree.hip cannot hold that.  This file is the same path with no bound
// on the slots of a feature (4 G cells in all):
//
//   thresholds   every column is radix-sorted on the device (the order of radix.cc:28-73);
//                the host runs mart.cc:140-169 over the sorted values with the reference's
//                own f32 operations.  Rows are ragged: feature f owns cells
//                [woff[f], woff[f + 1]) of one flat array.
//   bins         u32, feature-major [F][N]: bin = first slot with x <= threshold
//                (rtnode_histogram.cc:241-251), a binary search per (document, feature)
//   k_whist      node histogram: one workgroup = one feature x a range of the node's
//                documents.  Rows of up to 16384 slots are accumulated in LDS (the packed
//                count + sum cell of k_tree.hip: one LDS atomic per document; the lanes of a
//                wave that share a hot slot are summed in registers first) and flushed with
//                global atomics per touched cell; longer rows go straight to global atomics.
//                Integer sums: any order, same bits (as k_tree.hip).
//   k_wscan      per feature: prefix over the slots in chunks of 1024 with a carry,
//                sibling = parent - child on the cumulative arrays
//                (rtnode_histogram.cc:72-87, 206-217), gain of every slot and the first
//                maximum (rt.cc:257-292) -- the records k_decide merges; rows longer than
//                QR_WCHUNK slots are scanned in chunks (k_wscan_tot / _chunk / _best)
//   k_wobl_*     the same for level-wise (oblivious) growth, ot.cc:32-201
// Growth itself -- k_decide, k_partition, k_finish, the leaf and score kernels -- is
// k_tree.hip's one-split-per-step path, reading u32 bins and ragged thresholds.
// Single GPU.  The u8 path keeps its kernels and its speed; this one is the general
// one: on a 713k-document set 1.9x the u8 path's time per iteration at 1024 slots per
// feature, 28x with every distinct value a threshold (67 M cells per node histogram).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "qr_internal.h"
#include "qr_wave.h"
#include "qr_dev.h"

#define QR_WLDS_SLOTS 16384u  /* rows up to this many slots are accumulated in LDS (128 KB)          */
#define QR_WDOCS 16384u       /* documents per histogram workgroup: an LDS cell is the packed   */
                              /* count * 2^QR_SB + sum of k_tree.hip, good for QR_DPW documents */

// ---------------------------------------------------------------------------
// thresholds
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t w_flip(uint32_t x) {
  return x ^ ((uint32_t)(-(int32_t)(x >> 31)) | 0x80000000u);  // radix.cc:28-30
}
static inline uint32_t h_unflip(uint32_t x) { return x ^ (((x >> 31) - 1) | 0x80000000u); }
static inline float bits2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

__global__ __launch_bounds__(256) void k_wkeys(const float *__restrict__ col, const uint32_t N,
                                               uint32_t *__restrict__ keys) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) keys[i] = w_flip(__float_as_uint(col[i]));
}

// col: device [F][N] column-major.  Fills c->h_wthr / h_woff for the rank's OWN features
// (c->h_lf2gf: every feature on one GPU, the rank's range on a feature-sharded context --
// local index lf, rows concatenated in local order) and c->h_thr_size by GLOBAL feature
// (0 for a feature another rank owns).
int qr_k_wide_thresholds(qr_ctx *c, const float *d_col, size_t nthresholds) {
  const size_t N = c->N, F = c->h_lf2gf.size();
  uint32_t *d_keys = nullptr, *d_sorted = nullptr;
  void *d_temp = nullptr;
  size_t temp_bytes = 0;
  QR_CHECK(c, hipMalloc((void **)&d_keys, N * 4));
  QR_CHECK(c, hipMalloc((void **)&d_sorted, N * 4));
  QR_CHECK(c, hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d_keys, d_sorted, (int)N));
  QR_CHECK(c, hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
  std::vector<uint32_t> h(N);
  c->h_wthr.clear();
  c->h_woff.assign(F + 1, 0);
  c->h_thr_size.assign(c->F, 0);
  std::vector<float> uniqs;
  for (size_t f = 0; f < F; ++f) {
    const size_t gf = (size_t)c->h_lf2gf[f];
    hipLaunchKernelGGL(k_wkeys, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream,
                       d_col + gf * N, (uint32_t)N, d_keys);
    QR_CHECK(c, hipGetLastError());
    size_t tb = temp_bytes;
    QR_CHECK(c, hipcub::DeviceRadixSort::SortKeys(d_temp, tb, d_keys, d_sorted, (int)N, 0, 32, c->stream));
    QR_D2H(c, h.data(), d_sorted, N * 4);
    // mart.cc:140-152: the distinct values in sorted order, early stop once there are
    // nthresholds + 1 of them (a strict `<` decides "distinct": -0.0 and 0.0 are one
    // value, a NaN is never appended behind a number)
    uniqs.clear();
    uniqs.push_back(bits2f(h_unflip(h[0])));
    for (size_t j = 1; j < N && (nthresholds == 0 || uniqs.size() != nthresholds + 1); ++j) {
      const float v = bits2f(h_unflip(h[j]));
      if (uniqs.back() < v) uniqs.push_back(v);
    }
    c->h_woff[f] = (uint32_t)c->h_wthr.size();
    if (uniqs.size() <= nthresholds || nthresholds == 0) {  // mart.cc:155-158
      c->h_wthr.insert(c->h_wthr.end(), uniqs.begin(), uniqs.end());
      c->h_wthr.push_back(FLT_MAX);
      c->h_thr_size[gf] = (uint32_t)uniqs.size() + 1;
    } else {  // mart.cc:159-169: equal width, a running f32 sum
      float t = bits2f(h_unflip(h[0]));
      const float step = (float)fabs(bits2f(h_unflip(h[N - 1])) - t) / nthresholds;
      for (size_t j = 0; j != nthresholds; t += step, ++j) c->h_wthr.push_back(t);
      c->h_wthr.push_back(FLT_MAX);
      c->h_thr_size[gf] = (uint32_t)nthresholds + 1;
    }
    if (c->h_wthr.size() >= 0xFFFFFFF0ull) {
      (void)hipFree(d_keys); (void)hipFree(d_sorted); (void)hipFree(d_temp);
      QR_FAIL(c, QR_ERR_UNSUPPORTED, "more than 2^32 threshold slots in all");
    }
  }
  c->h_woff[F] = (uint32_t)c->h_wthr.size();
  (void)hipFree(d_keys);
  (void)hipFree(d_sorted);
  (void)hipFree(d_temp);
  return QR_OK;
}

// Column statistics of this rank's documents for the thresholds of a DOCUMENT-SHARDED set (the
// wide counterpart of qr_k_colstats): per feature the first `limit` distinct values in sorted order
// (mart.cc:140-152's own loop over the radix-sorted column: strict `<`, so -0.0 and 0.0 are one
// value and a NaN never follows a number), their number (limit + 1 = "more than limit"), and the
// radix keys of the smallest and largest value.  vals: raw f32 bit patterns.
int qr_k_wide_stats(qr_ctx *c, const float *d_col, size_t limit, uint32_t *vals, uint32_t *cnt, uint32_t *mm) {
  const size_t N = c->N, F = c->F;
  if (!N) {
    for (size_t f = 0; f < F; ++f) cnt[f] = 0, mm[2 * f] = 0xFFFFFFFFu, mm[2 * f + 1] = 0;
    return QR_OK;
  }
  uint32_t *d_keys = nullptr, *d_sorted = nullptr;
  void *d_temp = nullptr;
  size_t temp_bytes = 0;
  std::vector<uint32_t> h(N);
  // (one exit: the scratch goes whatever happens)
  auto run = [&]() -> int {
    QR_CHECK(c, hipMalloc((void **)&d_keys, N * 4));
    QR_CHECK(c, hipMalloc((void **)&d_sorted, N * 4));
    QR_CHECK(c, hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d_keys, d_sorted, (int)N));
    QR_CHECK(c, hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
    for (size_t f = 0; f < F; ++f) {
      hipLaunchKernelGGL(k_wkeys, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, d_col + f * N,
                         (uint32_t)N, d_keys);
      QR_CHECK(c, hipGetLastError());
      size_t tb = temp_bytes;
      QR_CHECK(c, hipcub::DeviceRadixSort::SortKeys(d_temp, tb, d_keys, d_sorted, (int)N, 0, 32, c->stream));
      QR_D2H(c, h.data(), d_sorted, N * 4);
      uint32_t *out = vals + f * limit;
      size_t n = 0;
      float last = bits2f(h_unflip(h[0]));
      out[n++] = h_unflip(h[0]);
      for (size_t j = 1; j < N && n <= limit; ++j) {
        const float v = bits2f(h_unflip(h[j]));
        if (last < v) {
          if (n < limit) out[n] = h_unflip(h[j]);
          ++n;
          last = v;
        }
      }
      cnt[f] = (uint32_t)n;  // (limit + 1: there are more)
      mm[2 * f] = h[0];
      mm[2 * f + 1] = h[N - 1];
    }
    return QR_OK;
  };
  const int rc = run();
  if (d_keys) (void)hipFree(d_keys);
  if (d_sorted) (void)hipFree(d_sorted);
  if (d_temp) (void)hipFree(d_temp);
  return rc;
}

// ---------------------------------------------------------------------------
// bins: first slot t with x <= thr[t] (lower bound over the non-decreasing row);
// NaN and anything above the last finite threshold land in the FLT_MAX slot
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wbinning(const float *__restrict__ col, const uint32_t N,
                                                  const float *__restrict__ thr,
                                                  const uint32_t *__restrict__ woff,
                                                  uint32_t *__restrict__ bins,
                                                  const int32_t *__restrict__ lf2gf) {
  const uint32_t f = blockIdx.y;   // local feature
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  if (d >= N) return;
  const float x = col[(size_t)lf2gf[f] * N + d];
  const float *t = thr + woff[f];
  const uint32_t size = woff[f + 1] - woff[f];
  uint32_t lo = 0, hi = size;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (x <= t[mid])
      hi = mid;
    else
      lo = mid + 1;
  }
  bins[(size_t)f * N + d] = lo > size - 1 ? size - 1 : lo;
}

// Rows of up to QR_W16_SLOTS slots (--num-thresholds up to 1151, e.g. the common 1024): a second
// copy of the bins as u16, row-major in groups of 16 features -- [group][doc][16], 32 bytes
// per (document, group): one sector per gathered document -- for k_whist16.  Columns beyond
// F are padded with slot 0 (never flushed).
__global__ __launch_bounds__(256) void k_wbins16(const uint32_t *__restrict__ bins, const uint32_t N,
                                                 const uint32_t F, uint16_t *__restrict__ out) {
  const uint32_t g = blockIdx.y;
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  if (d >= N) return;
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t f0 = g * 16 + 2 * k, f1 = f0 + 1;
    const uint32_t a = f0 < F ? bins[(size_t)f0 * N + d] : 0u;
    const uint32_t b = f1 < F ? bins[(size_t)f1 * N + d] : 0u;
    w[k] = (a & 0xffffu) | (b << 16);
  }
  uint4 *o = reinterpret_cast<uint4 *>(out + ((size_t)g * N + d) * 16);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// verify-after-write of the wide bin map (k_bins.hip: k_bins_verify says why): every slot recomputed
// from the column and compared with the u32 map and, where it exists, with the u16 copy -- by a
// workgroup three places away from the one that stored it.  out[0] / out[1]: cells that differ.
__global__ __launch_bounds__(256) void k_wbins_verify(const float *__restrict__ col, const uint32_t N,
                                                      const float *__restrict__ thr,
                                                      const uint32_t *__restrict__ woff,
                                                      const uint32_t *__restrict__ bins,
                                                      const uint16_t *__restrict__ bins16, const uint32_t F,
                                                      const int32_t *__restrict__ lf2gf,
                                                      unsigned long long *__restrict__ out) {
  const uint32_t f = blockIdx.y;
  const uint32_t d = ((blockIdx.x + 3u) % gridDim.x) * 256 + threadIdx.x;
  if (d >= N) return;
  const float x = col[(size_t)lf2gf[f] * N + d];
  const float *t = thr + woff[f];
  const uint32_t size = woff[f + 1] - woff[f];
  uint32_t lo = 0, hi = size;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (x <= t[mid])
      hi = mid;
    else
      lo = mid + 1;
  }
  const uint32_t want = lo > size - 1 ? size - 1 : lo;
  if (bins[(size_t)f * N + d] != want) atomicAdd(&out[0], 1ull);
  if (bins16 && bins16[((size_t)(f / 16) * N + d) * 16 + (f % 16)] != (uint16_t)want) atomicAdd(&out[1], 1ull);
}

static int wide_binning_launches(qr_ctx *c, const float *d_col) {
  hipLaunchKernelGGL(k_wbinning, dim3((unsigned)((c->N + 255) / 256), (unsigned)c->flocal), dim3(256), 0,
                     c->stream, d_col, (uint32_t)c->N, c->d_wthr, c->d_woff, c->d_wbins, c->d_lf2gf);
  QR_CHECK(c, hipGetLastError());
  if (c->d_wbins16) {
    hipLaunchKernelGGL(k_wbins16, dim3((unsigned)((c->N + 255) / 256), (unsigned)((c->flocal + 15) / 16)), dim3(256), 0,
                       c->stream, c->d_wbins, (uint32_t)c->N, (uint32_t)c->flocal, c->d_wbins16);
    QR_CHECK(c, hipGetLastError());
  }
  return QR_OK;
}

int qr_k_wide_binning(qr_ctx *c, const float *d_col) {
  if (!c->N || !c->flocal) return wide_binning_launches(c, d_col);
  unsigned long long *d_out = nullptr;
  QR_CHECK(c, hipMalloc((void **)&d_out, 16));
  for (int attempt = 0;; ++attempt) {
    int rc = wide_binning_launches(c, d_col);
    if (rc) {
      (void)hipFree(d_out);
      return rc;
    }
    unsigned long long h[2] = {0, 0};
    hipError_t e = hipMemsetAsync(d_out, 0, 16, c->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_wbins_verify, dim3((unsigned)((c->N + 255) / 256), (unsigned)c->flocal), dim3(256), 0, c->stream,
                         d_col, (uint32_t)c->N, c->d_wthr, c->d_woff, (const uint32_t *)c->d_wbins,
                         (const uint16_t *)c->d_wbins16, (uint32_t)c->flocal, c->d_lf2gf, d_out);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      (void)hipFree(d_out);
      QR_CHECK(c, e);
    }
    if (!h[0] && !h[1]) break;
    fprintf(stderr, "qr: the wide bin map on the device does not hold what the binning kernel stored (%llu cells of the u32 "
                    "map, %llu of the u16 copy, attempt %d): device memory lost stores -- %s\n",
            h[0], h[1], attempt + 1, attempt < 2 ? "building it again at another address" : "giving up");
    if (attempt == 2) {
      (void)hipFree(d_out);
      c->err = "the wide bin map on the device does not hold what the binning kernel stored, three times in a row: "
               "device memory loses stores (profiles/r06_hunt.md)";
      return QR_ERR_HIP;
    }
    // (other addresses for the next attempt, as in qr_k_binning: new buffers first, then the old ones go)
    uint32_t *nb = nullptr;
    uint16_t *n16 = nullptr;
    const size_t FL = (size_t)c->flocal;
    e = hipMalloc((void **)&nb, c->N * FL * sizeof(uint32_t));
    if (e == hipSuccess && c->d_wbins16) e = hipMalloc((void **)&n16, c->N * 16 * ((FL + 15) / 16) * sizeof(uint16_t));
    if (e != hipSuccess) {
      (void)hipFree(d_out);
      if (nb) (void)hipFree(nb);
      QR_CHECK(c, e);
    }
    (void)hipFree(c->d_wbins);
    c->d_wbins = nb;
    if (n16) {
      (void)hipFree(c->d_wbins16);
      c->d_wbins16 = n16;
    }
  }
  (void)hipFree(d_out);
  return QR_OK;
}

// ---------------------------------------------------------------------------
// node histograms
// ---------------------------------------------------------------------------
// what a launch works on: the root (every document or the sample's list), the directly
// built child of the current split (leaf-wise growth), or that of node `j` of a level
struct WSeg {
  uint32_t begin, n;
  int buf, slot;
  bool active;
};
__device__ __forceinline__ WSeg w_segment(const QrTreeState *ts, const int mode, const uint32_t rootn,
                                          const int root_buf, const uint32_t j) {
  WSeg s;
  if (mode == 0) {          // root
    s.begin = 0;
    s.n = rootn;
    s.buf = root_buf;
    s.slot = 0;
    s.active = true;
  } else if (mode == 1 || mode == 4) {   // the split being applied (4: on a document-sharded rank --
    s.active = ts->desc.active != 0;     // the segment of its OWN lists, QrLocalSplit)
    s.begin = mode == 4 ? ts->loc.small_begin : ts->desc.small_begin;
    s.n = mode == 4 ? ts->loc.small_n : ts->desc.small_n;
    s.buf = ts->desc.dst_buf;
    s.slot = ts->desc.small_slot;
  } else {                  // node j of the level (2) / job j of a batched growth step (3)
    const QrLevelNode &ln = ts->lnode[j];
    s.active = (mode == 3 || !ts->obl_done) && (int)j < ts->l_nodes && ln.active;
    s.begin = ln.small_begin;
    s.n = ln.small_n;
    s.buf = ln.dst_buf;
    s.slot = ln.small_slot;
  }
  return s;
}

// the raw (not yet cumulative) cells of the segment's slot start at zero
__global__ __launch_bounds__(256) void k_wzero(const QrTreeState *__restrict__ ts, const int mode,
                                               const size_t cells, long long *__restrict__ hsum,
                                               uint32_t *__restrict__ hcnt) {
  const WSeg s = w_segment(ts, mode, 0, 0, blockIdx.y);
  if (!s.active) return;
  const size_t base = (size_t)s.slot * cells;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (size_t)gridDim.x * 256) {
    hsum[base + i] = 0;
    hcnt[base + i] = 0;
  }
}

__global__ __launch_bounds__(1024) void k_whist(
    const QrTreeState *__restrict__ ts, const int mode, const uint32_t rootn, const int root_buf,
    const uint32_t N, const uint32_t *__restrict__ bins, const uint32_t *__restrict__ woff,
    const size_t cells, const uint32_t *__restrict__ order0, const uint32_t *__restrict__ order1,
    const double *__restrict__ lambda, const QrScalars *__restrict__ scal,
    long long *__restrict__ hsum, uint32_t *__restrict__ hcnt) {
  extern __shared__ __attribute__((aligned(16))) char wlds[];
  const WSeg s = w_segment(ts, mode, rootn, root_buf, blockIdx.z);
  if (!s.active) return;
  const uint32_t r0 = blockIdx.x * QR_WDOCS;
  if (r0 >= s.n) return;
  const uint32_t r1 = r0 + QR_WDOCS < s.n ? r0 + QR_WDOCS : s.n;
  const uint32_t lf = blockIdx.y;
  const uint32_t base = woff[lf], size = woff[lf + 1] - base;
  const uint32_t *order = s.buf == 0 ? order0 : order1;
  const uint32_t *row = bins + (size_t)lf * N;
  const double scale = scal->scale;
  long long *gs = hsum + (size_t)s.slot * cells + base;
  uint32_t *gc = hcnt + (size_t)s.slot * cells + base;
  const bool in_lds = size <= QR_WLDS_SLOTS;
  u64 *cell = reinterpret_cast<u64 *>(wlds);  // [size] packed count * 2^QR_SB + sum (qr_internal.h)
  if (in_lds) {
    for (uint32_t i = threadIdx.x; i < size; i += 1024) cell[i] = 0;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  for (uint32_t p0 = r0 + (threadIdx.x & ~63u); p0 < r1; p0 += 1024) {  // wave-uniform trip count
    const uint32_t p = p0 + lane;
    const bool in = p < r1;
    const uint32_t id = in ? (s.buf == 2 ? s.begin + p : order[s.begin + p]) : 0u;
    const uint32_t b = in ? row[id] : 0xFFFFFFFFu;
    long long q = in ? quantize(lambda[id] * scale) : 0;
    uint32_t cnt = in ? 1u : 0u;
    // A hot slot (the zeros of a count feature: most of a wave's documents) would make the
    // lanes' atomics queue on one address: the lanes that share the first active lane's
    // slot add up in registers and their leader issues one atomic for all of them.
    const unsigned long long act = __ballot(in);
    if (act) {
      const int first = __ffsll((long long)act) - 1;
      const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)b, first);
      const unsigned long long same = __ballot(in && b == b0);
      if (__popcll(same) >= 8) {
        const bool mine = in && b == b0;
        const long long tot = readlane_i64(wave_scan_i64(mine ? q : 0), 63);
        if (lane == first) {
          q = tot;
          cnt = (uint32_t)__popcll(same);
        } else if (mine) {
          cnt = 0;  // carried by the leader
        }
      }
    }
    if (cnt) {
      if (in_lds) {
        atomicAdd(&cell[b], ((u64)cnt << QR_SB) + (u64)q);
      } else {
        atomicAdd(reinterpret_cast<u64 *>(&gs[b]), (u64)q);
        atomicAdd(&gc[b], cnt);
      }
    }
  }
  if (!in_lds) return;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < size; i += 1024) {
    const u64 v = cell[i];
    if (v) {
      const u64 cn = (v + (1ull << (QR_SB - 1))) >> QR_SB;
      atomicAdd(reinterpret_cast<u64 *>(&gs[i]), v - (cn << QR_SB));
      atomicAdd(&gc[i], (uint32_t)cn);
    }
  }
}

// The same histogram for rows of up to QR_W16_SLOTS slots, the way k_tree.hip builds the
// 256-slot ones: one workgroup = 16 features x a range of the node's documents, an LDS
// histogram [slot][16] of packed cells, lane = document.  A lane loads the 16 u16 bins of its
// document (32 bytes, one sector) and issues 16 LDS atomics; at step k it updates column
// (k + lane) & 15, so the 16 lanes of an LDS lane group touch 16 distinct bank pairs whatever
// the slots are -- k_whist's one atomic per 12 gathered bytes at random banks was 64 % of an
// iteration with 1024 thresholds (787 of 1230 us, profiles/r03_wide.md).  The halfwords are
// rotated in registers (dword rotation by selects, the odd halfword by v_alignbyte) so that
// step k reads a fixed position.  Flush: global atomics per touched cell into the node's
// ragged arrays, as k_whist does -- same integers, same bits.
#define QR_W16_SLOTS 1152u
#ifndef QR_W16_DOCS
#define QR_W16_DOCS 8192u   /* documents per k_whist16 workgroup (<= QR_WDOCS: the cell's count bits) */
#endif
// first partial slot (in units of one document range) of node j of the launch: the ranges of
// the active nodes before it (level-wise growth; a single node otherwise)
__device__ __forceinline__ uint32_t w16_slot_base(const QrTreeState *ts, const int mode, const uint32_t j) {
  if (mode < 2 || mode == 4) return 0;
  uint32_t b = 0;
  for (uint32_t k = 0; k < j; ++k) {
    const QrLevelNode &ln = ts->lnode[k];
    if (ln.active) b += (ln.small_n + QR_W16_DOCS - 1) / QR_W16_DOCS;
  }
  return b;
}
__global__ __launch_bounds__(1024) void k_whist16(
    const QrTreeState *__restrict__ ts, const int mode, const uint32_t rootn, const int root_buf,
    const uint32_t N, const uint32_t F, const uint16_t *__restrict__ bins16,
    const uint32_t *__restrict__ woff, const size_t cells, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, const double *__restrict__ lambda,
    const QrScalars *__restrict__ scal, u64 *__restrict__ wpart, const uint32_t slots) {
  extern __shared__ __attribute__((aligned(16))) char wlds[];
  const WSeg s = w_segment(ts, mode, rootn, root_buf, blockIdx.z);
  if (!s.active) return;
  const uint32_t r0 = blockIdx.x * QR_W16_DOCS;
  if (r0 >= s.n) return;
  const uint32_t r1 = r0 + QR_W16_DOCS < s.n ? r0 + QR_W16_DOCS : s.n;
  const uint32_t g = blockIdx.y;
  const uint32_t *order = s.buf == 0 ? order0 : order1;
  const bool identity = s.buf == 2;
  const uint16_t *rows = bins16 + (size_t)g * N * 16;
  const double scale = scal->scale;
  u64 *cell = reinterpret_cast<u64 *>(wlds);  // [slots][16]
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t r = lane & 15, dr = r >> 1, hb = 2u * (r & 1u);
  uint32_t colk8[16];  // byte offset of the column a step updates
#pragma unroll
  for (int k = 0; k < 16; ++k) colk8[k] = ((k + r) & 15u) * 8u;
  auto get_id = [&](uint32_t p) -> uint32_t { return identity ? s.begin + p : order[s.begin + p]; };
  const uint32_t plast = r1 - 1;
  auto clampp = [&](uint32_t p) { return p < plast ? p : plast; };
  auto process = [&](const uint4 &lo4, const uint4 &hi4, const double lam) {
    const u64 addend = (1ull << QR_SB) + (u64)quantize(lam * scale);
    const uint32_t w[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    uint32_t t[8], u[8], v[8], R[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (dr & 1) ? w[(i + 1) & 7] : w[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = (dr & 2) ? t[(i + 2) & 7] : t[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (dr & 4) ? u[(i + 4) & 7] : u[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) R[i] = __builtin_amdgcn_alignbyte(v[(i + 1) & 7], v[i], hb);
#pragma unroll
    for (int k = 0; k < 16; ++k) {  // halfword k of R = bin of column (k + r) & 15
      const uint32_t bin = (k & 1) ? (R[k >> 1] >> 16) : (R[k >> 1] & 0xffffu);
      atomicAdd(reinterpret_cast<u64 *>(wlds + bin * 128u + colk8[k]), addend);
    }
  };
  // two register sets: the rows and gradients of the next tile are in flight behind the
  // tile whose atomics are issuing (unconditional, clamped loads, as hist_accumulate)
  const uint32_t step = 1024;
  const uint32_t p0 = r0 + threadIdx.x;
  uint4 a0, a1, b0, b1;
  double la, lb;
  uint32_t ida = get_id(clampp(p0)), idb = get_id(clampp(p0 + step));
  {
    const uint4 *ra = reinterpret_cast<const uint4 *>(rows + (size_t)ida * 16);
    a0 = ra[0];
    a1 = ra[1];
    la = lambda[ida];
    const uint4 *rb = reinterpret_cast<const uint4 *>(rows + (size_t)idb * 16);
    b0 = rb[0];
    b1 = rb[1];
    lb = lambda[idb];
  }
  ida = get_id(clampp(p0 + 2 * step));
  idb = get_id(clampp(p0 + 3 * step));
  for (uint32_t i = threadIdx.x * 2; i < slots * 16; i += 2048) {
    cell[i] = 0;
    cell[i + 1] = 0;
  }
  __syncthreads();
  const uint32_t wbase = r0 + (threadIdx.x & ~63u);
  for (uint32_t wb = wbase; wb < r1; wb += 2 * step) {  // (wave-uniform trip count)
    const uint32_t p = wb + lane;
    if (p < r1) process(a0, a1, la);
    {
      const uint4 *ra = reinterpret_cast<const uint4 *>(rows + (size_t)ida * 16);
      a0 = ra[0];
      a1 = ra[1];
      la = lambda[ida];
      ida = get_id(clampp(p + 4 * step));
    }
    if (p + step < r1) process(b0, b1, lb);
    {
      const uint4 *rb = reinterpret_cast<const uint4 *>(rows + (size_t)idb * 16);
      b0 = rb[0];
      b1 = rb[1];
      lb = lambda[idb];
      idb = get_id(clampp(p + 5 * step));
    }
  }
  __syncthreads();
  // flush: the whole LDS histogram, as it is, into the workgroup's partial slot (plain
  // coalesced stores; k_wreduce16 adds the node's slots up).  Global atomics per touched cell
  // -- two per cell, ~13 M for a root of 713k documents -- were 90 % of this kernel's time.
  u64 *dst = wpart + ((size_t)(w16_slot_base(ts, mode, blockIdx.z) + blockIdx.x) * gridDim.y + g) * ((size_t)slots * 16);
  for (uint32_t i = threadIdx.x * 2; i < slots * 16; i += 2048) {
    ulonglong2 v;
    v.x = cell[i];
    v.y = cell[i + 1];
    *reinterpret_cast<ulonglong2 *>(dst + i) = v;
  }
}

// sum of the partial slots of one node's launch, unpacked into the node's ragged arrays
// (every cell of every row is written: no zeroing pass): grid = (cells / 256, groups, nodes)
__global__ __launch_bounds__(256) void k_wreduce16(
    const QrTreeState *__restrict__ ts, const int mode, const uint32_t rootn, const int root_buf,
    const uint32_t F, const uint32_t *__restrict__ woff, const size_t cells,
    const u64 *__restrict__ wpart, long long *__restrict__ hsum, uint32_t *__restrict__ hcnt,
    const uint32_t slots) {
  const WSeg s = w_segment(ts, mode, rootn, root_buf, blockIdx.z);
  if (!s.active) return;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= slots * 16) return;
  const uint32_t g = blockIdx.y, t = i >> 4, c = i & 15u, f = g * 16 + c;
  if (f >= F) return;
  const uint32_t base = woff[f], size = woff[f + 1] - base;
  if (t >= size) return;
  const uint32_t nch = (s.n + QR_W16_DOCS - 1) / QR_W16_DOCS;
  const u64 *src = wpart + ((size_t)w16_slot_base(ts, mode, blockIdx.z) * gridDim.y + g) * ((size_t)slots * 16) + i;
  const size_t stride = (size_t)gridDim.y * slots * 16;
  long long sum = 0;
  uint32_t cn = 0;
  for (uint32_t k = 0; k < nch; ++k) {
    const u64 v = src[k * stride];
    const u64 cc = (v + (1ull << (QR_SB - 1))) >> QR_SB;
    sum += (long long)(v - (cc << QR_SB));
    cn += (uint32_t)cc;
  }
  hsum[(size_t)s.slot * cells + base + t] = sum;
  hcnt[(size_t)s.slot * cells + base + t] = cn;
}

// ---------------------------------------------------------------------------
// scan: prefix, sibling, gains, first maximum
// ---------------------------------------------------------------------------
// inclusive scan of (s, cn) over the 1024 threads of the workgroup, plus the carry
__device__ __forceinline__ void w_block_scan(long long &s, uint32_t &cn, long long &carry_s,
                                             uint32_t &carry_c, long long *sh_s, uint32_t *sh_c) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s = wave_scan_i64(s);
  cn = wave_scan_u32(cn);
  __syncthreads();  // (the previous round's readers of sh_* are done)
  if (lane == 63) {
    sh_s[wave] = s;
    sh_c[wave] = cn;
  }
  __syncthreads();
  long long ps = carry_s;
  uint32_t pc = carry_c;
  for (int w = 0; w < wave; ++w) {
    ps += sh_s[w];
    pc += sh_c[w];
  }
  s += ps;
  cn += pc;
  long long ts_ = carry_s;
  uint32_t tc_ = carry_c;
  for (int w = 0; w < 16; ++w) {
    ts_ += sh_s[w];
    tc_ += sh_c[w];
  }
  carry_s = ts_;
  carry_c = tc_;
}

// first maximum over the workgroup: highest score, equal scores -> lowest slot
__device__ __forceinline__ Best w_block_best(Best v, Best *sh) {
  const double m = wave_max(v.score);
  const uint32_t t = wave_min_u32(v.score == m && v.t != 0xFFFFFFFFu ? v.t : 0xFFFFFFFFu);
  Best w;
  w.score = t != 0xFFFFFFFFu ? m : -1.0;
  w.t = t;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
  __syncthreads();
  Best r = sh[0];
  for (int i = 1; i < 16; ++i) r = best_pick(r, sh[i]);
  return r;
}

__device__ __forceinline__ void w_record(qr_split_t *o, float *othr, const Best v, const int gf,
                                         const uint32_t *cnt_row, const uint32_t total_c,
                                         const float *thr_row) {
  if (threadIdx.x != 0) return;
  o->score = v.score;
  o->feature = v.t == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)gf;
  o->thr_id = v.t;
  o->lcount = v.t == 0xFFFFFFFFu ? 0 : cnt_row[v.t];
  o->rcount = v.t == 0xFFFFFFFFu ? 0 : total_c - cnt_row[v.t];
  *othr = v.t == 0xFFFFFFFFu ? 0.f : thr_row[v.t];
}

// One feature of the node just accumulated (mode 0: root, 1: the split being applied):
// cumulative arrays of the directly built child in place, the sibling's by subtraction,
// and the per-feature best-split record of each (featrec[which * flocal + lf], which =
// 0 left / root, 1 right), as k_scan leaves them for k_decide.
__global__ __launch_bounds__(1024) void k_wscan(
    const QrTreeState *__restrict__ ts, const int mode, const uint32_t *__restrict__ woff,
    const size_t cells, long long *__restrict__ hsum, uint32_t *__restrict__ hcnt, const int flocal,
    const int32_t *__restrict__ lf2gf, const float *__restrict__ thr,
    const QrScalars *__restrict__ scal, qr_split_t *__restrict__ featrec,
    float *__restrict__ featthr, const double *__restrict__ part_ss = nullptr,
    double *__restrict__ jobsum = nullptr, const u64 minls_root = ~0ull) {
  // mode 3: job blockIdx.y of a batched growth step (k_tree.hip, k_decide_part): the split's
  // descriptor is ts->lnode[job], the records go to featrec[2 job + which], and feature 0's
  // workgroup adds up the partition workgroups' sums of the directly built child (jobsum)
  __shared__ long long sh_s[16];
  __shared__ uint32_t sh_c[16];
  __shared__ Best sh_b[16];
  int small_slot = 0, big_slot = -1, parent_slot = -1, small_is_left = 1;
  const bool child = mode != 0;
  const uint32_t job = mode == 3 ? blockIdx.y : 0u;
  uint32_t part_first = 0, part_nwg = 0;
  if (mode == 1) {
    if (!ts->desc.active) return;
    small_slot = ts->desc.small_slot;
    big_slot = ts->desc.big_slot;
    parent_slot = ts->desc.parent_slot;
    small_is_left = ts->desc.small_is_left;
  } else if (mode == 3) {
    if ((int)job >= ts->l_nodes) return;
    const QrLevelNode &ln = ts->lnode[job];
    if (!ln.active) return;
    small_slot = ln.small_slot;
    big_slot = ln.big_slot;
    parent_slot = ln.parent_slot;
    small_is_left = ln.small_is_left;
    part_first = ln.part_first;
    part_nwg = (ln.end - ln.begin + QR_PART_SLICE - 1) / QR_PART_SLICE;
    featrec += (size_t)2 * job * flocal;
    featthr += (size_t)2 * job * flocal;
  }
  const int lf = blockIdx.x;
  if (mode == 3 && lf == 0 && (threadIdx.x >> 6) == 15 && jobsum) {
    // fixed-order reduction of the partition workgroups' partials (lane-strided, wave_sum)
    double pa = 0.0, pb = 0.0;
    for (uint32_t i = threadIdx.x & 63; i < part_nwg; i += 64) {
      pa += part_ss[2 * (size_t)(part_first + i)];
      pb += part_ss[2 * (size_t)(part_first + i) + 1];
    }
    pa = wave_sum(pa);
    pb = wave_sum(pb);
    if ((threadIdx.x & 63) == 0) {
      jobsum[2 * job] = pa;
      jobsum[2 * job + 1] = pb;
    }
  }
  const uint32_t base = woff[lf], size = woff[lf + 1] - base;
  long long *ss = hsum + (size_t)small_slot * cells + base;
  uint32_t *sc = hcnt + (size_t)small_slot * cells + base;
  long long *bs = child ? hsum + (size_t)big_slot * cells + base : nullptr;
  uint32_t *bc = child ? hcnt + (size_t)big_slot * cells + base : nullptr;
  const long long *ps = child ? hsum + (size_t)parent_slot * cells + base : nullptr;
  const uint32_t *pc = child ? hcnt + (size_t)parent_slot * cells + base : nullptr;
  // pass 1: prefix over the slots (exact integers: any association), sibling
  long long carry_s = 0;
  uint32_t carry_c = 0;
  for (uint32_t t0 = 0; t0 < size; t0 += 1024) {
    const uint32_t t = t0 + threadIdx.x;
    long long s = t < size ? ss[t] : 0;
    uint32_t cn = t < size ? sc[t] : 0u;
    w_block_scan(s, cn, carry_s, carry_c, sh_s, sh_c);
    if (t < size) {
      ss[t] = s;
      sc[t] = cn;
      if (child) {
        bs[t] = ps[t] - s;
        bc[t] = pc[t] - cn;
      }
    }
  }
  __syncthreads();
  // pass 2: gains (rt.cc:268-291) and the first maximum (rt.cc:285)
  // (the root of a batched tree: its state is only initialised by the first control call,
  // behind this launch -- the host passes the tree's minimum leaf support)
  const u64 minls = (mode == 0 && minls_root != ~0ull) ? minls_root : ts->minls;
  const double inv_scale = scal->inv_scale;
  const int gf = lf2gf[lf];
  const long long S0 = carry_s;
  const uint32_t C0 = carry_c;
  const long long S1 = child ? ps[size - 1] - S0 : 0;
  const uint32_t C1 = child ? pc[size - 1] - C0 : 0u;
  Best a, b;
  a.score = b.score = -1.0;
  a.t = b.t = 0xFFFFFFFFu;
  for (uint32_t t = threadIdx.x; t < size; t += 1024) {  // ascending t per thread: strict > keeps the first
    const Best v = slot_gain(ss[t], sc[t], S0, C0, t, size, minls, inv_scale);
    if (v.score > a.score) a = v;
    if (child) {
      const Best w = slot_gain(bs[t], bc[t], S1, C1, t, size, minls, inv_scale);
      if (w.score > b.score) b = w;
    }
  }
  a = w_block_best(a, sh_b);
  const int wa = mode == 0 ? 0 : (small_is_left ? 0 : 1);
  w_record(&featrec[(size_t)wa * flocal + lf], &featthr[(size_t)wa * flocal + lf], a, gf, sc, C0,
           thr + base);
  if (child) {
    b = w_block_best(b, sh_b);
    const int wb = small_is_left ? 1 : 0;
    w_record(&featrec[(size_t)wb * flocal + lf], &featthr[(size_t)wb * flocal + lf], b, gf, bc, C1,
             thr + base);
  }
}

// ---- the same scan for long rows, in chunks --------------------------------------------
// A row of 10^5 - 10^6 slots (--num-thresholds 0 on a real-valued column) scanned by ONE
// workgroup leaves the launch with as many busy CUs as there are long rows.  The chunked
// scan cuts every row into pieces of QR_WCHUNK slots (table built once with the bins):
//   k_wscan_tot    per chunk: total of its raw cells
//   k_wscan_chunk  per chunk: carry-in = the totals of the row's earlier chunks, prefix,
//                  sibling, gains with the row totals (= all of the row's chunk totals),
//                  the chunk's best slot of each child
//   k_wscan_best   per feature: first maximum over the row's chunks -> featrec
// Exact integers and the first maximum in slot order: the same records as k_wscan.
struct WChunk {
  uint32_t lf, t0;
};

__device__ __forceinline__ void w_scan_setup(const QrTreeState *ts, const int mode, int &small_slot,
                                             int &big_slot, int &parent_slot, int &small_is_left,
                                             bool &active) {
  small_slot = 0;
  big_slot = parent_slot = -1;
  small_is_left = 1;
  active = true;
  if (mode == 1) {
    active = ts->desc.active != 0;
    small_slot = ts->desc.small_slot;
    big_slot = ts->desc.big_slot;
    parent_slot = ts->desc.parent_slot;
    small_is_left = ts->desc.small_is_left;
  }
}

__global__ __launch_bounds__(1024) void k_wscan_tot(
    const QrTreeState *__restrict__ ts, const int mode, const WChunk *__restrict__ chunks,
    const uint32_t *__restrict__ woff, const size_t cells, const long long *__restrict__ hsum,
    const uint32_t *__restrict__ hcnt, long long *__restrict__ tot_s, uint32_t *__restrict__ tot_c) {
  __shared__ long long sh_s[16];
  __shared__ uint32_t sh_c[16];
  int small_slot, big_slot, parent_slot, small_is_left;
  bool active;
  w_scan_setup(ts, mode, small_slot, big_slot, parent_slot, small_is_left, active);
  if (!active) return;
  const WChunk ch = chunks[blockIdx.x];
  const uint32_t base = woff[ch.lf], size = woff[ch.lf + 1] - base;
  const uint32_t t1 = ch.t0 + QR_WCHUNK < size ? ch.t0 + QR_WCHUNK : size;
  const long long *ss = hsum + (size_t)small_slot * cells + base;
  const uint32_t *sc = hcnt + (size_t)small_slot * cells + base;
  long long s = 0;
  uint32_t cn = 0;
  for (uint32_t t = ch.t0 + threadIdx.x; t < t1; t += 1024) {
    s += ss[t];
    cn += sc[t];
  }
  s = wave_scan_i64(s);   // (lane 63 holds the wave's sum)
  cn = wave_scan_u32(cn);
  if ((threadIdx.x & 63) == 63) {
    sh_s[threadIdx.x >> 6] = s;
    sh_c[threadIdx.x >> 6] = cn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long a = 0;
    uint32_t b = 0;
    for (int w = 0; w < 16; ++w) {
      a += sh_s[w];
      b += sh_c[w];
    }
    tot_s[blockIdx.x] = a;
    tot_c[blockIdx.x] = b;
  }
}

__global__ __launch_bounds__(1024) void k_wscan_chunk(
    const QrTreeState *__restrict__ ts, const int mode, const WChunk *__restrict__ chunks,
    const uint32_t *__restrict__ chunk0, const uint32_t *__restrict__ woff, const size_t cells,
    long long *__restrict__ hsum, uint32_t *__restrict__ hcnt, const long long *__restrict__ tot_s,
    const uint32_t *__restrict__ tot_c, const QrScalars *__restrict__ scal, Best *__restrict__ cbest) {
  __shared__ long long sh_s[16];
  __shared__ uint32_t sh_c[16];
  __shared__ Best sh_b[16];
  __shared__ long long row_s[2];
  __shared__ uint32_t row_c[2];
  int small_slot, big_slot, parent_slot, small_is_left;
  bool active;
  w_scan_setup(ts, mode, small_slot, big_slot, parent_slot, small_is_left, active);
  if (!active) return;
  const WChunk ch = chunks[blockIdx.x];
  const uint32_t base = woff[ch.lf], size = woff[ch.lf + 1] - base;
  const uint32_t t1 = ch.t0 + QR_WCHUNK < size ? ch.t0 + QR_WCHUNK : size;
  long long *ss = hsum + (size_t)small_slot * cells + base;
  uint32_t *sc = hcnt + (size_t)small_slot * cells + base;
  long long *bs = mode == 1 ? hsum + (size_t)big_slot * cells + base : nullptr;
  uint32_t *bc = mode == 1 ? hcnt + (size_t)big_slot * cells + base : nullptr;
  const long long *ps = mode == 1 ? hsum + (size_t)parent_slot * cells + base : nullptr;
  const uint32_t *pc = mode == 1 ? hcnt + (size_t)parent_slot * cells + base : nullptr;
  // the row's chunk totals: those before this chunk are its carry, all of them the row total
  const uint32_t c0 = chunk0[ch.lf], c1 = chunk0[ch.lf + 1], me = blockIdx.x;
  long long a_s = 0, b_s = 0;
  uint32_t a_c = 0, b_c = 0;
  for (uint32_t k = c0 + threadIdx.x; k < c1; k += 1024) {
    const long long v = tot_s[k];
    const uint32_t w = tot_c[k];
    b_s += v;
    b_c += w;
    if (k < me) {
      a_s += v;
      a_c += w;
    }
  }
  // (rows have at most a few hundred chunks: the first wave's lanes hold everything)
  a_s = wave_scan_i64(a_s);
  a_c = wave_scan_u32(a_c);
  b_s = wave_scan_i64(b_s);
  b_c = wave_scan_u32(b_c);
  if ((threadIdx.x & 63) == 63) {
    sh_s[threadIdx.x >> 6] = a_s;
    sh_c[threadIdx.x >> 6] = a_c;
  }
  __syncthreads();
  long long carry_s = 0;
  uint32_t carry_c = 0;
  for (int w = 0; w < 16; ++w) {
    carry_s += sh_s[w];
    carry_c += sh_c[w];
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 63) {
    sh_s[threadIdx.x >> 6] = b_s;
    sh_c[threadIdx.x >> 6] = b_c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long x = 0;
    uint32_t y = 0;
    for (int w = 0; w < 16; ++w) {
      x += sh_s[w];
      y += sh_c[w];
    }
    row_s[0] = x;
    row_c[0] = y;
    row_s[1] = mode == 1 ? ps[size - 1] - x : 0;
    row_c[1] = mode == 1 ? pc[size - 1] - y : 0u;
  }
  __syncthreads();
  const long long S0 = row_s[0], S1 = row_s[1];
  const uint32_t C0 = row_c[0], C1 = row_c[1];
  const u64 minls = ts->minls;
  const double inv_scale = scal->inv_scale;
  Best a, b;
  a.score = b.score = -1.0;
  a.t = b.t = 0xFFFFFFFFu;
  for (uint32_t r0 = ch.t0; r0 < t1; r0 += 1024) {
    const uint32_t t = r0 + threadIdx.x;
    long long s = t < t1 ? ss[t] : 0;
    uint32_t cn = t < t1 ? sc[t] : 0u;
    const long long par_s = (mode == 1 && t < t1) ? ps[t] : 0;
    const uint32_t par_c = (mode == 1 && t < t1) ? pc[t] : 0u;
    w_block_scan(s, cn, carry_s, carry_c, sh_s, sh_c);
    if (t < t1) {
      ss[t] = s;
      sc[t] = cn;
      const Best v = slot_gain(s, cn, S0, C0, t, size, minls, inv_scale);
      if (v.score > a.score) a = v;  // ascending t per thread: strict > keeps the first
      if (mode == 1) {
        const long long gs = par_s - s;
        const uint32_t gc = par_c - cn;
        bs[t] = gs;
        bc[t] = gc;
        const Best w = slot_gain(gs, gc, S1, C1, t, size, minls, inv_scale);
        if (w.score > b.score) b = w;
      }
    }
  }
  a = w_block_best(a, sh_b);
  if (mode == 1) b = w_block_best(b, sh_b);
  if (threadIdx.x == 0) {
    cbest[2 * (size_t)blockIdx.x] = a;
    cbest[2 * (size_t)blockIdx.x + 1] = b;
  }
}

__global__ __launch_bounds__(64) void k_wscan_best(
    const QrTreeState *__restrict__ ts, const int mode, const uint32_t *__restrict__ chunk0,
    const uint32_t *__restrict__ woff, const size_t cells, const uint32_t *__restrict__ hcnt,
    const int flocal, const int32_t *__restrict__ lf2gf, const float *__restrict__ thr,
    const Best *__restrict__ cbest, qr_split_t *__restrict__ featrec, float *__restrict__ featthr) {
  int small_slot, big_slot, parent_slot, small_is_left;
  bool active;
  w_scan_setup(ts, mode, small_slot, big_slot, parent_slot, small_is_left, active);
  if (!active) return;
  const int lf = blockIdx.x;
  const uint32_t base = woff[lf], size = woff[lf + 1] - base;
  const uint32_t c0 = chunk0[lf], c1 = chunk0[lf + 1];
  const uint32_t *sc = hcnt + (size_t)small_slot * cells + base;
  const uint32_t *bc = mode == 1 ? hcnt + (size_t)big_slot * cells + base : nullptr;
  for (int which = 0; which < (mode == 1 ? 2 : 1); ++which) {
    Best v;
    v.score = -1.0;
    v.t = 0xFFFFFFFFu;
    for (uint32_t k = c0 + threadIdx.x; k < c1; k += 64) {  // ascending chunks per lane
      const Best w = cbest[2 * (size_t)k + which];
      if (w.score > v.score) v = w;
    }
    const double m = wave_max(v.score);
    const uint32_t t = wave_min_u32(v.score == m && v.t != 0xFFFFFFFFu ? v.t : 0xFFFFFFFFu);
    Best r;
    r.score = t != 0xFFFFFFFFu ? m : -1.0;
    r.t = t;
    const uint32_t *row = which == 0 ? sc : bc;
    const int w = mode == 0 ? 0 : ((which == 0) == (small_is_left != 0) ? 0 : 1);
    w_record(&featrec[(size_t)w * flocal + lf], &featthr[(size_t)w * flocal + lf], r, lf2gf[lf], row,
             row[size - 1], thr + base);
  }
}

// level-wise growth: prefix + sibling for every node of the level (no gains here:
// k_wobl_fill sums them over the level)
__global__ __launch_bounds__(1024) void k_wscan_level(
    const QrTreeState *__restrict__ ts, const uint32_t *__restrict__ woff, const size_t cells,
    long long *__restrict__ hsum, uint32_t *__restrict__ hcnt) {
  __shared__ long long sh_s[16];
  __shared__ uint32_t sh_c[16];
  if (ts->obl_done || (int)blockIdx.y >= ts->l_nodes) return;
  const QrLevelNode &ln = ts->lnode[blockIdx.y];
  if (!ln.active) return;
  const int lf = blockIdx.x;
  const uint32_t base = woff[lf], size = woff[lf + 1] - base;
  long long *ss = hsum + (size_t)ln.small_slot * cells + base;
  uint32_t *sc = hcnt + (size_t)ln.small_slot * cells + base;
  long long *bs = hsum + (size_t)ln.big_slot * cells + base;
  uint32_t *bc = hcnt + (size_t)ln.big_slot * cells + base;
  const long long *ps = hsum + (size_t)ln.parent_slot * cells + base;
  const uint32_t *pc = hcnt + (size_t)ln.parent_slot * cells + base;
  long long carry_s = 0;
  uint32_t carry_c = 0;
  for (uint32_t t0 = 0; t0 < size; t0 += 1024) {
    const uint32_t t = t0 + threadIdx.x;
    long long s = t < size ? ss[t] : 0;
    uint32_t cn = t < size ? sc[t] : 0u;
    w_block_scan(s, cn, carry_s, carry_c, sh_s, sh_c);
    if (t < size) {
      ss[t] = s;
      sc[t] = cn;
      bs[t] = ps[t] - s;
      bc[t] = pc[t] - cn;
    }
  }
}

// fill() + argmax of one level for one feature (ot.cc:177-201, 67-92), any row length:
// the gain of slot t summed over the level's nodes in node order, sticky `invalid`,
// only sums > 0 compete, first maximum
__global__ __launch_bounds__(1024) void k_wobl_fill(
    const QrTreeState *__restrict__ ts, const int level, const uint32_t *__restrict__ woff,
    const size_t cells, const long long *__restrict__ hsum, const uint32_t *__restrict__ hcnt,
    const int flocal, const int32_t *__restrict__ lf2gf, const QrScalars *__restrict__ scal,
    qr_split_t *__restrict__ featrec) {
  __shared__ Best sh_b[16];
  if (ts->obl_done) return;
  const int lf = blockIdx.x;
  const uint32_t base = woff[lf], size = woff[lf + 1] - base;
  const u64 minls = ts->minls;
  const double inv_scale = scal->inv_scale;
  const int lbegin = (1 << level) - 1, lend = (1 << (level + 1)) - 1;
  Best best;
  best.score = -1.0;
  best.t = 0xFFFFFFFFu;
  for (uint32_t t = threadIdx.x; t < size; t += 1024) {
    double sum = 0.0;
    bool invalid = false;
    for (int i = lbegin; i < lend; ++i) {
      const size_t nb = (size_t)ts->nodes[i].hslot * cells + base;
      const long long cs = hsum[nb + t], S = hsum[nb + size - 1];
      const u64 lc = hcnt[nb + t], C = hcnt[nb + size - 1];
      const u64 rc = C - lc;
      if (lc >= minls && rc >= minls) {
        const double s = (double)S * inv_scale;
        const double lsum = (double)cs * inv_scale;
        const double rsum = s - lsum;
        sum += lsum * lsum / (double)lc + rsum * rsum / (double)rc;
      } else
        invalid = true;
    }
    if (!invalid && sum > 0.0 && sum > best.score) {  // NaN fails the comparisons, as in ot.cc:77-78
      best.score = sum;
      best.t = t;
    }
  }
  best = w_block_best(best, sh_b);
  if (threadIdx.x == 0) {
    qr_split_t *o = &featrec[lf];
    o->score = best.score;
    o->feature = best.t == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)lf2gf[lf];
    o->thr_id = best.t;
    o->lcount = o->rcount = 0;
  }
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
static size_t whist_lds(const qr_ctx *c) {
  const size_t s = std::min<size_t>(c->wmax, QR_WLDS_SLOTS);
  return s * 8 + 16;
}

static int whist_attr(qr_ctx *c) {
  size_t &attr = c->attr_whist_lds;
  const size_t lds = whist_lds(c);
  if (lds > attr) {
    QR_CHECK(c, hipFuncSetAttribute((const void *)k_whist, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    if (c->d_wbins16)
      QR_CHECK(c, hipFuncSetAttribute((const void *)k_whist16, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(QR_W16_SLOTS * 128)));
    attr = lds;
  }
  return QR_OK;
}

// (the LDS-tiled kernel holds a whole row group: QR_W16_SLOTS * 128 bytes -- only where a workgroup can have them)
bool qr_k_wide_fast_rows(const qr_ctx *c, size_t max_slots) {
  return max_slots <= QR_W16_SLOTS && c->lds_block >= (size_t)QR_W16_SLOTS * 128;
}
static bool qr_k_wide_fast(const qr_ctx *c) { return c->d_wbins16 && !getenv("QR_WIDE_NO_FAST"); }

// the node histograms of a launch: rows of up to QR_W16_SLOTS slots through the blocked u16
// copy (k_whist16), longer ones feature by feature (k_whist)
static int launch_whist(qr_ctx *c, const int mode, const uint32_t rootn, const int root_buf, const size_t maxn,
                        const unsigned nodes, const QrTreeState *ts = nullptr) {
  if (!ts) ts = c->d_tree;  // (batched growth hands over the copy of the tree state its step wrote)
  // (maxn: the most documents a node of the launch can hold)
  const unsigned chunks = (unsigned)((maxn + (qr_k_wide_fast(c) ? QR_W16_DOCS : QR_WDOCS) - 1) /
                                     (qr_k_wide_fast(c) ? QR_W16_DOCS : QR_WDOCS)) + (mode >= 2 ? 1u : 0u);
  if (qr_k_wide_fast(c)) {
    const uint32_t slots = ((uint32_t)c->wmax + 1u) & ~1u;
    const unsigned groups = (unsigned)((c->flocal + 15) / 16);
    // partial slots: one per (document range, group); level-wise growth packs the nodes' ranges
    const size_t need = ((size_t)chunks + nodes) * groups * slots * 16;
    if (need > c->wpart_cap) {
      QR_CHECK(c, hipStreamSynchronize(c->stream));
      if (c->d_wpart) (void)hipFree(c->d_wpart);
      c->d_wpart = nullptr;
      QR_CHECK(c, hipMalloc((void **)&c->d_wpart, need * 8));
      c->wpart_cap = need;
    }
    hipLaunchKernelGGL(k_whist16, dim3(chunks, groups, nodes), dim3(1024), (size_t)slots * 128, c->stream,
                       ts, mode, rootn, root_buf, (uint32_t)c->N, (uint32_t)c->flocal, c->d_wbins16, c->d_woff,
                       c->wcells, c->d_order[0], c->d_order[1], c->d_lambda, c->d_scalars, (u64 *)c->d_wpart,
                       slots);
    QR_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(k_wreduce16, dim3((slots * 16 + 255) / 256, groups, nodes), dim3(256), 0, c->stream,
                       ts, mode, rootn, root_buf, (uint32_t)c->flocal, c->d_woff, c->wcells,
                       (const u64 *)c->d_wpart, c->d_hsum, c->d_hcnt, slots);
  } else {
    hipLaunchKernelGGL(k_whist, dim3(chunks, (unsigned)c->flocal, nodes), dim3(1024), whist_lds(c), c->stream,
                       ts, mode, rootn, root_buf, (uint32_t)c->N, c->d_wbins, c->d_woff, c->wcells,
                       c->d_order[0], c->d_order[1], c->d_lambda, c->d_scalars, c->d_hsum, c->d_hcnt);
  }
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// ---- document-sharded ranks (round 4): the node histogram of the rank's own documents goes through
// the exchange buffer -- [cells] sums, [cells] counts as int64, 2 f64 bit patterns per rank, ONE sum
// all-reduce, as on the u8 path (k_reduce / k_scan in k_tree.hip) -- and comes back as the histogram
// over ALL ranks' documents, which the scan kernels then treat as on one GPU.
//   k_wd_pack   (before the all-reduce) one workgroup per feature: the raw cells of the directly
//               built child -> the buffer; the rank's own CUMULATIVE counts of that child and of
//               its sibling -> hcnt_loc (where its partition cuts its lists: make_desc);
//               workgroup 0 also leaves the rank's (sum of squares, sum) of the child in its pair
//   k_wd_unpack (after) the summed cells -> the child's slot
__global__ __launch_bounds__(1024) void k_wd_pack(
    const QrTreeState *__restrict__ ts, const int root_mode, const uint32_t *__restrict__ woff, const size_t cells,
    const long long *__restrict__ hsum, const uint32_t *__restrict__ hcnt, uint32_t *__restrict__ hcnt_loc,
    long long *__restrict__ xh, const double *__restrict__ part_ss, const int rank, const int world) {
  __shared__ long long sh_s[16];
  __shared__ uint32_t sh_c[16];
  int small_slot = 0, big_slot = -1, parent_slot = -1;
  if (!root_mode) {
    if (!ts->desc.active) return;
    small_slot = ts->desc.small_slot;
    big_slot = ts->desc.big_slot;
    parent_slot = ts->desc.parent_slot;
  }
  const uint32_t lf = blockIdx.x, base = woff[lf], size = woff[lf + 1] - base;
  if (lf == 0 && threadIdx.x < 64) {  // the tail: this rank's pair, zeros elsewhere (sum == gather)
    double a = 0.0, b = 0.0;
    if (!root_mode) {
      const uint32_t nwg = (ts->desc.end - ts->desc.begin + QR_PART_SLICE - 1) / QR_PART_SLICE;
      for (uint32_t i = threadIdx.x; i < nwg; i += 64) {
        a += part_ss[2 * i];
        b += part_ss[2 * i + 1];
      }
      a = wave_sum(a);
      b = wave_sum(b);
    }
    long long *tail = xh + 2 * cells;
    for (int i = threadIdx.x; i < 2 * world; i += 64) {
      long long v = 0;
      if (i == 2 * rank) v = __double_as_longlong(a);
      if (i == 2 * rank + 1) v = __double_as_longlong(b);
      tail[i] = v;
    }
  }
  const long long *ss = hsum + (size_t)small_slot * cells + base;
  const uint32_t *sc = hcnt + (size_t)small_slot * cells + base;
  uint32_t *ls = hcnt_loc + (size_t)small_slot * cells + base;
  long long carry_s = 0;
  uint32_t carry_c = 0;
  for (uint32_t t0 = 0; t0 < size; t0 += 1024) {
    const uint32_t t = t0 + threadIdx.x;
    long long s = 0;
    uint32_t cn = t < size ? sc[t] : 0u;
    if (t < size) {
      xh[base + t] = ss[t];
      xh[cells + base + t] = (long long)cn;
    }
    w_block_scan(s, cn, carry_s, carry_c, sh_s, sh_c);
    if (t < size) {
      ls[t] = cn;
      if (!root_mode)
        hcnt_loc[(size_t)big_slot * cells + base + t] = hcnt_loc[(size_t)parent_slot * cells + base + t] - cn;
    }
  }
}

__global__ __launch_bounds__(256) void k_wd_unpack(const QrTreeState *__restrict__ ts, const int root_mode,
                                                   const size_t cells, const long long *__restrict__ xh,
                                                   long long *__restrict__ hsum, uint32_t *__restrict__ hcnt) {
  int slot = 0;
  if (!root_mode) {
    if (!ts->desc.active) return;
    slot = ts->desc.small_slot;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (size_t)gridDim.x * 256) {
    hsum[(size_t)slot * cells + i] = xh[i];
    hcnt[(size_t)slot * cells + i] = (uint32_t)xh[cells + i];
  }
}

static int launch_wscan(qr_ctx *c, int mode);

// mode 0: root histogram -> slot 0; mode 1: the directly built child of the split being
// applied (+ its sibling); then the per-feature records for k_decide
int qr_k_whist_scan(qr_ctx *c, int root_mode) {
  if (qr_exact_active(c)) return qr_k_exact_scan(c, root_mode);  // long rows: the pre-sorted lists (k_exact.hip)
  int rc = whist_attr(c);
  if (rc) return rc;
  const int mode = root_mode ? 0 : 1;
  const int hmode = !root_mode && c->dmode ? 4 : mode;  // (a document-sharded rank: the segment of its own lists)
  const uint32_t rootn = (uint32_t)(c->sub_k ? c->sub_n : c->N);
  const size_t maxn = root_mode ? rootn : (c->dmode ? rootn : rootn / 2 + 1);  // the smaller child (of ALL ranks' documents)
  if (!qr_k_wide_fast(c)) {  // (the fast rows' reduce writes every cell: nothing to zero)
    const unsigned zg = (unsigned)std::min<size_t>((c->wcells + 255) / 256, 4096);
    hipLaunchKernelGGL(k_wzero, dim3(zg, 1), dim3(256), 0, c->stream, c->d_tree, hmode, c->wcells, c->d_hsum,
                       c->d_hcnt);
    QR_CHECK(c, hipGetLastError());
  }
  if ((rc = launch_whist(c, hmode, rootn, c->sub_k ? 0 : 2, maxn, 1))) return rc;
  if (c->dmode) {  // the scan follows the all-reduce (qr_k_wscan_doc)
    hipLaunchKernelGGL(k_wd_pack, dim3((unsigned)c->flocal), dim3(1024), 0, c->stream, c->d_tree, root_mode, c->d_woff,
                       c->wcells, (const long long *)c->d_hsum, (const uint32_t *)c->d_hcnt, c->d_hcnt_loc, c->d_xh,
                       c->d_part_ss, c->rank, c->world);
    QR_CHECK(c, hipGetLastError());
    return QR_OK;
  }
  return launch_wscan(c, mode);
}

// document-sharded: the all-reduced cells -> the node's slot, then the scan
int qr_k_wscan_doc(qr_ctx *c, int root_mode) {
  const unsigned zg = (unsigned)std::min<size_t>((c->wcells + 255) / 256, 4096);
  hipLaunchKernelGGL(k_wd_unpack, dim3(zg), dim3(256), 0, c->stream, c->d_tree, root_mode, c->wcells,
                     (const long long *)c->d_xh, c->d_hsum, c->d_hcnt);
  QR_CHECK(c, hipGetLastError());
  return launch_wscan(c, root_mode ? 0 : 1);
}

static int launch_wscan(qr_ctx *c, int mode) {
  const bool no_chunks = getenv("QR_WIDE_NO_CHUNKS") != nullptr;  // (A/B aid: same records either way)
  if (c->wmax <= QR_WCHUNK || no_chunks) {  // short rows: one workgroup per feature
    hipLaunchKernelGGL(k_wscan, dim3((unsigned)c->flocal), dim3(1024), 0, c->stream, c->d_tree, mode,
                       c->d_woff, c->wcells, c->d_hsum, c->d_hcnt, c->flocal, c->d_lf2gf, c->d_wthr,
                       c->d_scalars, c->d_featrec, c->d_featthr, (const double *)nullptr, (double *)nullptr,
                       c->batch_root ? (u64)c->cur_minls : ~0ull);
    QR_CHECK(c, hipGetLastError());
    return QR_OK;
  }
  const unsigned nch = (unsigned)c->wchunks;
  hipLaunchKernelGGL(k_wscan_tot, dim3(nch), dim3(1024), 0, c->stream, c->d_tree, mode,
                     (const WChunk *)c->d_wchunk, c->d_woff, c->wcells, c->d_hsum, c->d_hcnt, c->d_wtot_s,
                     c->d_wtot_c);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_wscan_chunk, dim3(nch), dim3(1024), 0, c->stream, c->d_tree, mode,
                     (const WChunk *)c->d_wchunk, c->d_wchunk0, c->d_woff, c->wcells, c->d_hsum, c->d_hcnt,
                     c->d_wtot_s, c->d_wtot_c, c->d_scalars, (Best *)c->d_wcbest);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_wscan_best, dim3((unsigned)c->flocal), dim3(64), 0, c->stream, c->d_tree, mode,
                     c->d_wchunk0, c->d_woff, c->wcells, c->d_hcnt, c->flocal, c->d_lf2gf, c->d_wthr,
                     (const Best *)c->d_wcbest, c->d_featrec, c->d_featthr);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// batched leaf-wise growth (k_tree.hip, qr_k_tree_fit_batch): the directly built children of the
// step's jobs (ts->lnode[0 .. QR_BATCH), the copy of the tree state the step's control call
// wrote), their siblings and the per-feature records of both -- rows short enough for k_wscan
// (only the LDS-tiled rows: with the general histogram kernel the histograms of the splits
// applied ahead of their turn cost more than the shorter chain saves -- 4096 thresholds on the
// MSLR-shaped set: 1.63 ms per iteration batched against 1.47 over the first trees, 1.78 / 1.78
// over 40)
bool qr_k_wide_batch_ok(const qr_ctx *c) {
  return c->wide && c->d_wbins16 != nullptr && c->wmax <= QR_WCHUNK && !qr_exact_active(c);
}
int qr_k_whist_scan_batch(qr_ctx *c, const QrTreeState *ts, const double *pss) {
  int rc = whist_attr(c);
  if (rc) return rc;
  const uint32_t rootn = (uint32_t)(c->sub_k ? c->sub_n : c->N);
  if (!qr_k_wide_fast(c)) {
    const unsigned zg = (unsigned)std::min<size_t>((c->wcells + 255) / 256, 4096);
    hipLaunchKernelGGL(k_wzero, dim3(zg, QR_BATCH), dim3(256), 0, c->stream, ts, 3, c->wcells, c->d_hsum,
                       c->d_hcnt);
    QR_CHECK(c, hipGetLastError());
  }
  if ((rc = launch_whist(c, 3, rootn, 2, (size_t)rootn / 2 + 1, QR_BATCH, ts))) return rc;
  hipLaunchKernelGGL(k_wscan, dim3((unsigned)c->flocal, QR_BATCH), dim3(1024), 0, c->stream, ts, 3, c->d_woff,
                     c->wcells, c->d_hsum, c->d_hcnt, c->flocal, c->d_lf2gf, c->d_wthr, c->d_scalars,
                     c->d_featrec, c->d_featthr, pss, c->d_jobsum);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// level-wise growth: the level's per-feature sums (before its split is chosen) ...
int qr_k_wobl_fill(qr_ctx *c, int level) {
  hipLaunchKernelGGL(k_wobl_fill, dim3((unsigned)c->flocal), dim3(1024), 0, c->stream, c->d_tree, level,
                     c->d_woff, c->wcells, c->d_hsum, c->d_hcnt, c->flocal, c->d_lf2gf, c->d_scalars,
                     c->d_featrec);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// ... and the histograms of the level's children once it is partitioned
int qr_k_wobl_hist(qr_ctx *c, int nodes) {
  int rc = whist_attr(c);
  if (rc) return rc;
  if (!qr_k_wide_fast(c)) {
    const unsigned zg = (unsigned)std::min<size_t>((c->wcells + 255) / 256, 4096);
    hipLaunchKernelGGL(k_wzero, dim3(zg, (unsigned)nodes), dim3(256), 0, c->stream, c->d_tree, 2, c->wcells,
                       c->d_hsum, c->d_hcnt);
    QR_CHECK(c, hipGetLastError());
  }
  // the directly built child of a node holds at most half of its documents
  if ((rc = launch_whist(c, 2, (uint32_t)c->N, 2, c->N / 2, (unsigned)nodes))) return rc;
  hipLaunchKernelGGL(k_wscan_level, dim3((unsigned)c->flocal, (unsigned)nodes), dim3(1024), 0, c->stream,
                     c->d_tree, c->d_woff, c->wcells, c->d_hsum, c->d_hcnt);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}
