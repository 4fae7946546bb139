// k_exact.hip -- the reference's DEFAULT thresholds (`--num-thresholds 0`: every distinct value of
// a column is a threshold, mart.cc:155-158) at a cost that follows the node's DOCUMENTS, not the
// feature's slots.
//
// With 10^5 - 10^6 slots per real-valued column a node histogram (rtnode_histogram.cc:41-87) is
// 0.8 GB of cells that k_wide.hip zeroes, fills with global atomics and scans whatever the node's
// size: 21.5 ms per boosting iteration on the MSLR-shaped stand-in (profiles/r03_wide.md).  This
// file is the pre-sorted formulation of the same arithmetic -- what the reference itself starts
// from (`sortedsid_`, mart.cc:136-146) before it builds its bin map:
//   * once per data set: every feature's documents sorted by threshold slot (stable: ties keep
//     the document order), as 8-byte entries {slot, document} -- `xroot[f][N]`;
//   * a node is the same segment [begin, end) of EVERY feature's list (and of the document-order
//     list the leaf kernels use); its documents appear in each list in that feature's slot order;
//   * a node's split search is one pass over its segments: the running sum of the fixed-point
//     gradients IS the cumulative histogram `sumlbl[f][t]` at the last entry of slot t's run, the
//     position IS `count[f][t]`; the gain of rt.cc:276-279 is evaluated there, first maximum in
//     slot order (rt.cc:285).  Slots without a document of the node repeat the cumulative values
//     of the slot before them, so the first maximum never sits on one: the run ends are all the
//     candidates there are.  Exact integers, as in the histograms -- same records as k_wscan;
//   * a split is a STABLE partition of every feature's segment by the go-left test of the split
//     feature (`bin[f*][doc] <= t*`  <=>  `x <= threshold`, rt.cc:327-334): children stay sorted.
// One workgroup per (feature, child) streams its segment from begin to end with a running carry,
// so no launch needs another's partial results: k_xpart (F workgroups), k_xtotal (the children's
// gradient totals, which the gain needs before the first candidate), k_xscan (F x 2 workgroups).
// Per tree the lists move (1 + pi) N F entries through the scans and pi N F through the
// partitions (pi = documents partitioned / N, ~3.7 on the stand-in): ~10 GB against the ~150 GB
// of cells the slot-indexed path touches.
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "qr_internal.h"
#include "qr_wave.h"
#include "qr_dev.h"

#define QR_X_E 8u                      /* entries per thread and chunk */
#define QR_X_CHUNK (1024u * QR_X_E)    /* entries per chunk of a 1024-thread workgroup */

__device__ __forceinline__ uint32_t x_id(const u64 e) { return (uint32_t)e; }
__device__ __forceinline__ uint32_t x_slot(const u64 e) { return (uint32_t)(e >> 32); }

__global__ __launch_bounds__(256) void k_xiota(uint32_t *__restrict__ v, const uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) v[i] = i;
}
__global__ __launch_bounds__(256) void k_xpack(const uint32_t *__restrict__ slots, const uint32_t *__restrict__ ids,
                                               const uint32_t n, u64 *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) out[i] = ((u64)slots[i] << 32) | ids[i];
}

// the lists a node's segment lives in: 2 = the pristine root lists, 0 / 1 = the work buffers
__device__ __forceinline__ const u64 *x_lists(const int buf, const u64 *xroot, const u64 *x0, const u64 *x1) {
  return buf == 2 ? xroot : (buf == 0 ? x0 : x1);
}

// exclusive prefix of one value per thread over the 1024 threads of the workgroup, and the total
__device__ __forceinline__ void x_block_scan_u32(const uint32_t v, uint32_t &excl, uint32_t &total, uint32_t *sh) {
  const uint32_t inc = wave_scan_u32(v);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __syncthreads();  // (sh is free again)
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16; ++w) {
    const uint32_t t = sh[w];
    before += w < wave ? t : 0u;
    all += t;
  }
  excl = before + inc - v;
  total = all;
}
__device__ __forceinline__ void x_block_scan_i64(const long long v, long long &excl, long long &total, long long *sh) {
  const long long inc = wave_scan_i64(v);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  long long before = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16; ++w) {
    const long long t = sh[w];
    before += w < wave ? t : 0;
    all += t;
  }
  excl = before + inc - v;
  total = all;
}

// ---------------------------------------------------------------------------
// k_xpart: the split being applied (ts->desc), feature blockIdx.x's segment of the parent,
// stable partition into the children's segments of the destination lists.  The go-left test is
// the split feature's slot of the document (its column of the feature-major bins: 4 bytes per
// document, a few MB -- it stays in the L2 while 136 workgroups read it).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_xpart(const QrTreeState *__restrict__ ts, const u64 *__restrict__ xroot,
                                                u64 *__restrict__ x0, u64 *__restrict__ x1, const size_t N,
                                                const uint32_t *__restrict__ wbins) {
  __shared__ uint32_t sh[16];
  const QrSplitDesc d = ts->desc;
  if (!d.active || d.owner_local < 0) return;
  const size_t fo = (size_t)blockIdx.x * N;
  const u64 *src = x_lists(d.src_buf, xroot, x0, x1) + fo + d.begin;
  u64 *dst = (d.dst_buf == 0 ? x0 : x1) + fo;
  const uint32_t *col = wbins + (size_t)d.owner_local * N;
  const uint32_t n = d.end - d.begin;
  uint32_t lcur = d.begin, rcur = d.begin + d.lcount;
  // the next chunk's entries travel while this one is partitioned
  u64 nx[QR_X_E];
  auto fetch = [&](const uint32_t c0) {
#pragma unroll
    for (uint32_t k = 0; k < QR_X_E; ++k) {
      const uint32_t p = c0 + threadIdx.x * QR_X_E + k;
      nx[k] = src[p < n ? p : n - 1];  // (clamped: unconditional loads)
    }
  };
  if (n == 0) return;
  fetch(0);
  for (uint32_t c0 = 0; c0 < n; c0 += QR_X_CHUNK) {
    u64 e[QR_X_E];
    uint32_t sl[QR_X_E];
#pragma unroll
    for (uint32_t k = 0; k < QR_X_E; ++k) e[k] = nx[k];
#pragma unroll
    for (uint32_t k = 0; k < QR_X_E; ++k) sl[k] = col[x_id(e[k])];
    if (c0 + QR_X_CHUNK < n) fetch(c0 + QR_X_CHUNK);
    const uint32_t p0 = c0 + threadIdx.x * QR_X_E;
    uint32_t nl = 0, nv = 0;
    bool left[QR_X_E];
#pragma unroll
    for (uint32_t k = 0; k < QR_X_E; ++k) {
      const bool in = p0 + k < n;
      left[k] = in && sl[k] <= d.thr_id;
      nl += left[k] ? 1u : 0u;
      nv += in ? 1u : 0u;
    }
    uint32_t lo, ltot;
    x_block_scan_u32(nl, lo, ltot, sh);
    // (entries before this thread's in the chunk: p0 - c0 of them, all valid when any of its own is)
    uint32_t li = lcur + lo, ri = rcur + (threadIdx.x * QR_X_E - lo);
#pragma unroll
    for (uint32_t k = 0; k < QR_X_E; ++k) {
      if (p0 + k < n) {
        if (left[k])
          dst[li++] = e[k];
        else
          dst[ri++] = e[k];
      }
    }
    const uint32_t cn = n - c0 < QR_X_CHUNK ? n - c0 : QR_X_CHUNK;
    lcur += ltot;
    rcur += cn - ltot;
    (void)nv;
  }
}

// ---------------------------------------------------------------------------
// k_xtotal: the exact total of the fixed-point gradients of the node(s) about to be scanned
// (feature 0's segment: every feature's segment holds the same documents).  Integer atomics:
// any order gives the same bits.  mode 0: the root -> tot[0]; mode 1: the children of the split
// being applied -> tot[0] (left), tot[1] (right).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_xtotal(const QrTreeState *__restrict__ ts, const int mode,
                                                const uint32_t rootn, const u64 *__restrict__ xroot,
                                                const u64 *__restrict__ x0, const u64 *__restrict__ x1,
                                                const double *__restrict__ lambda,
                                                const QrScalars *__restrict__ scal, long long *__restrict__ tot) {
  uint32_t begin = 0, n = rootn;
  const u64 *lst = xroot;
  if (mode == 1) {
    const QrSplitDesc d = ts->desc;
    if (!d.active) return;
    begin = blockIdx.y == 0 ? d.begin : d.begin + d.lcount;
    n = blockIdx.y == 0 ? d.lcount : d.end - d.begin - d.lcount;
    lst = d.dst_buf == 0 ? x0 : x1;
  }
  const double scale = scal->scale;
  long long s = 0;
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u)
    s += quantize(lambda[x_id(lst[begin + p])] * scale);
  // wave total (exact), one atomic per wave
  s = wave_scan_i64(s);
  if ((threadIdx.x & 63u) == 63u && s != 0)
    atomicAdd(reinterpret_cast<unsigned long long *>(tot + blockIdx.y), (unsigned long long)s);
}

// ---------------------------------------------------------------------------
// k_xscan: feature blockIdx.x of node blockIdx.y (0: the root / the left child, 1: the right
// child): cumulative sums along the segment, the gain at every run end, the first maximum ->
// featrec[which * flocal + lf] (+ the threshold value), as k_wscan leaves them for k_decide.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_xscan(
    const QrTreeState *__restrict__ ts, const int mode, const uint32_t rootn, const u64 *__restrict__ xroot,
    const u64 *__restrict__ x0, const u64 *__restrict__ x1, const size_t N, const double *__restrict__ lambda,
    const QrScalars *__restrict__ scal, const long long *__restrict__ tot, const uint32_t *__restrict__ woff,
    const int flocal, const int32_t *__restrict__ lf2gf, const float *__restrict__ thr,
    qr_split_t *__restrict__ featrec, float *__restrict__ featthr, const u64 minls_root) {
  __shared__ long long sh_s[16];
  __shared__ uint32_t sh_first[1024];
  __shared__ Best sh_b[16];
  __shared__ uint32_t sh_lc[16];
  const int lf = blockIdx.x, which = blockIdx.y;
  uint32_t begin = 0, n = rootn;
  const u64 *lst = xroot;
  if (mode == 1) {
    const QrSplitDesc d = ts->desc;
    if (!d.active) return;
    begin = which == 0 ? d.begin : d.begin + d.lcount;
    n = which == 0 ? d.lcount : d.end - d.begin - d.lcount;
    lst = d.dst_buf == 0 ? x0 : x1;
  }
  const u64 *src = lst + (size_t)lf * N + begin;
  const u64 minls = (mode == 0 && minls_root != ~0ull) ? minls_root : ts->minls;
  const double scale = scal->scale, inv_scale = scal->inv_scale;
  const long long S = tot[which];
  const uint32_t base = woff[lf], tsize = woff[lf + 1] - base;
  Best best;
  best.score = -1.0;
  best.t = 0xFFFFFFFFu;
  uint32_t best_lc = 0;
  long long carry = 0;
  u64 nx[QR_X_E];
  auto fetch = [&](const uint32_t c0) {
#pragma unroll
    for (uint32_t k = 0; k < QR_X_E; ++k) {
      const uint32_t p = c0 + threadIdx.x * QR_X_E + k;
      nx[k] = src[p < n ? p : (n ? n - 1 : 0)];
    }
  };
  if (n) fetch(0);
  for (uint32_t c0 = 0; c0 < n; c0 += QR_X_CHUNK) {
    u64 e[QR_X_E];
    long long q[QR_X_E];
#pragma unroll
    for (uint32_t k = 0; k < QR_X_E; ++k) e[k] = nx[k];
    const uint32_t p0 = c0 + threadIdx.x * QR_X_E;
    {
      double lam[QR_X_E];
#pragma unroll
      for (uint32_t k = 0; k < QR_X_E; ++k) lam[k] = lambda[x_id(e[k])];
      // the slot that follows the chunk's last entry (the run-end test of its last thread)
      const bool more = c0 + QR_X_CHUNK < n;
      if (more) fetch(c0 + QR_X_CHUNK);
#pragma unroll
      for (uint32_t k = 0; k < QR_X_E; ++k) q[k] = p0 + k < n ? quantize(lam[k] * scale) : 0;
    }
    long long run = 0;
#pragma unroll
    for (uint32_t k = 0; k < QR_X_E; ++k) {
      run += q[k];
      q[k] = run;  // inclusive prefix inside the thread
    }
    sh_first[threadIdx.x] = x_slot(e[0]);
    long long off, ctot;
    x_block_scan_i64(run, off, ctot, sh_s);  // (its barriers also publish sh_first)
    // slot of the entry behind this thread's last one: the next thread's first, or -- last thread
    // -- the next chunk's first (already on its way: thread 0's first prefetched entry)
    uint32_t nslot;
    if (threadIdx.x < 1023u)
      nslot = sh_first[threadIdx.x + 1];
    else
      nslot = 0xFFFFFFFFu;
    {
      // (the next chunk's first slot: thread 0 holds it in nx[0]; hand it to the last thread)
      __syncthreads();
      if (threadIdx.x == 0) sh_first[0] = c0 + QR_X_CHUNK < n ? x_slot(nx[0]) : 0xFFFFFFFFu;
      __syncthreads();
      if (threadIdx.x == 1023u) nslot = sh_first[0];
    }
#pragma unroll
    for (uint32_t k = 0; k < QR_X_E; ++k) {
      const uint32_t p = p0 + k;
      if (p < n) {
        const uint32_t t = x_slot(e[k]);
        const uint32_t tn = k + 1 < QR_X_E ? x_slot(e[k + 1]) : nslot;
        const bool run_end = p + 1 == n || tn != t;
        if (run_end) {
          const Best v = slot_gain(carry + off + q[k], p + 1, S, n, t, tsize, minls, inv_scale);
          if (v.score > best.score) {  // ascending positions per thread: strict > keeps the first
            best = v;
            best_lc = p + 1;
          }
        }
      }
    }
    carry += ctot;
    __syncthreads();  // (sh_first is rewritten by the next round)
  }
  // first maximum over the workgroup: highest score, then lowest slot
  const double m = wave_max(best.score);
  const uint32_t tmin = wave_min_u32(best.score == m && best.t != 0xFFFFFFFFu ? best.t : 0xFFFFFFFFu);
  const unsigned long long holder = __ballot(best.score == m && best.t == tmin && tmin != 0xFFFFFFFFu);
  const uint32_t wlc = holder ? (uint32_t)__builtin_amdgcn_readlane((int)best_lc, __ffsll((long long)holder) - 1) : 0u;
  __syncthreads();
  if ((threadIdx.x & 63u) == 0) {
    Best w;
    w.score = tmin != 0xFFFFFFFFu ? m : -1.0;
    w.t = tmin;
    sh_b[threadIdx.x >> 6] = w;
    sh_lc[threadIdx.x >> 6] = wlc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Best r = sh_b[0];
    uint32_t lc = sh_lc[0];
    for (int i = 1; i < 16; ++i) {
      const Best o = sh_b[i];
      if (o.score > r.score || (o.score == r.score && o.t < r.t)) {
        r = o;
        lc = sh_lc[i];
      }
    }
    qr_split_t *o = &featrec[(size_t)which * flocal + lf];
    const bool none = r.t == 0xFFFFFFFFu;
    o->score = none ? -1.0 : r.score;
    o->feature = none ? 0xFFFFFFFFu : (uint32_t)lf2gf[lf];
    o->thr_id = r.t;
    o->lcount = none ? 0 : lc;
    o->rcount = none ? 0 : n - lc;
    featthr[(size_t)which * flocal + lf] = none ? 0.f : thr[base + r.t];
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
void qr_k_exact_free(qr_ctx *c) {
  if (c->d_xroot) (void)hipFree(c->d_xroot);
  if (c->d_xlist[0]) (void)hipFree(c->d_xlist[0]);
  if (c->d_xlist[1]) (void)hipFree(c->d_xlist[1]);
  if (c->d_xtot) (void)hipFree(c->d_xtot);
  c->d_xroot = c->d_xlist[0] = c->d_xlist[1] = nullptr;
  c->d_xtot = nullptr;
  c->xmode = false;
}

// the sorted lists of the context's data set (after qr_k_wide_binning): one stable radix sort per
// feature of (slot, document) pairs
int qr_k_exact_build(qr_ctx *c) {
  const size_t N = c->N, FL = (size_t)c->flocal;
  if (!N || !FL) return QR_OK;
  uint32_t *d_iota = nullptr, *d_ks = nullptr, *d_vs = nullptr;
  void *d_tmp = nullptr;
  size_t tmp_bytes = 0;
  int bits = 1;
  while (((size_t)1 << bits) < (size_t)c->wmax + 1 && bits < 32) ++bits;
  auto fail = [&](const char *what) {
    if (d_iota) (void)hipFree(d_iota);
    if (d_ks) (void)hipFree(d_ks);
    if (d_vs) (void)hipFree(d_vs);
    if (d_tmp) (void)hipFree(d_tmp);
    qr_k_exact_free(c);
    c->err = what;
    return QR_ERR_HIP;
  };
  if (hipMalloc((void **)&d_iota, N * 4) != hipSuccess || hipMalloc((void **)&d_ks, N * 4) != hipSuccess ||
      hipMalloc((void **)&d_vs, N * 4) != hipSuccess)
    return fail("allocating the sort scratch of the pre-sorted lists failed");
  if (hipMalloc((void **)&c->d_xroot, FL * N * 8) != hipSuccess || hipMalloc((void **)&c->d_xlist[0], FL * N * 8) != hipSuccess ||
      hipMalloc((void **)&c->d_xlist[1], FL * N * 8) != hipSuccess || hipMalloc((void **)&c->d_xtot, 16) != hipSuccess)
    return fail("allocating the pre-sorted lists failed");
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint32_t *)c->d_wbins, d_ks,
                                         (const uint32_t *)d_iota, d_vs, (int)N, 0, bits, c->stream) != hipSuccess ||
      hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16) != hipSuccess)
    return fail("sizing the radix sort of the pre-sorted lists failed");
  const unsigned g = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(k_xiota, dim3(g), dim3(256), 0, c->stream, d_iota, (uint32_t)N);
  for (size_t f = 0; f < FL; ++f) {
    if (hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, (const uint32_t *)c->d_wbins + f * N, d_ks,
                                           (const uint32_t *)d_iota, d_vs, (int)N, 0, bits, c->stream) != hipSuccess)
      return fail("the radix sort of a feature's documents by slot failed");
    hipLaunchKernelGGL(k_xpack, dim3(g), dim3(256), 0, c->stream, d_ks, d_vs, (uint32_t)N,
                       (u64 *)c->d_xroot + f * N);
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess)
    return fail("building the pre-sorted lists failed");
  (void)hipFree(d_iota); (void)hipFree(d_ks); (void)hipFree(d_vs); (void)hipFree(d_tmp);
  c->xmode = true;
  return QR_OK;
}

// the counterpart of qr_k_whist_scan on the pre-sorted lists: mode 0 the root, mode 1 the split
// being applied (partition of every feature's segment, then both children's records)
int qr_k_exact_scan(qr_ctx *c, int root_mode) {
  const uint32_t rootn = (uint32_t)c->N;
  const u64 *xr = (const u64 *)c->d_xroot;
  u64 *x0 = (u64 *)c->d_xlist[0], *x1 = (u64 *)c->d_xlist[1];
  const unsigned F = (unsigned)c->flocal;
  const int mode = root_mode ? 0 : 1;
  if (!root_mode) {
    hipLaunchKernelGGL(k_xpart, dim3(F), dim3(1024), 0, c->stream, c->d_tree, xr, x0, x1, c->N,
                       (const uint32_t *)c->d_wbins);
    QR_CHECK(c, hipGetLastError());
  }
  QR_CHECK(c, hipMemsetAsync(c->d_xtot, 0, 16, c->stream));
  hipLaunchKernelGGL(k_xtotal, dim3(64, root_mode ? 1 : 2), dim3(256), 0, c->stream, c->d_tree, mode, rootn, xr,
                     (const u64 *)x0, (const u64 *)x1, c->d_lambda, c->d_scalars, c->d_xtot);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_xscan, dim3(F, root_mode ? 1 : 2), dim3(1024), 0, c->stream, c->d_tree, mode, rootn, xr,
                     (const u64 *)x0, (const u64 *)x1, c->N, c->d_lambda, c->d_scalars, c->d_xtot, c->d_woff,
                     c->flocal, c->d_lf2gf, c->d_wthr, c->d_featrec, c->d_featthr,
                     c->batch_root ? (u64)c->cur_minls : ~0ull);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}
