// k_lambda.hip -- per-query ranking, metric and LambdaMART pseudo-responses.
//
// Stands behind LambdaMart::compute_pseudoresponses (lambdamart.cc:62-152),
// RankedResults (rankedresults.cc:27-41), QueryResults::indexing_of_sorted_labels
// (queryresults.cc:37-53), Ndcg/Dcg::jacobian (ndcg.cc:60-93, dcg.cc:59-83),
// Dcg::compute_dcg (dcg.cc:33-39), Ndcg::evaluate_result_list (ndcg.cc:49-58)
// and Metric::evaluate_dataset (metric.h:77-106).
//
// One wavefront (64 lanes) per query, everything staged in LDS:
//   1. rank by counting: rank_i = #{j : s_j > s_i or (s_j == s_i and j < i)} --
//      n broadcast LDS reads per lane, no sorting network, and tie detection for
//      free.  With distinct scores every correct sort gives this permutation.
//   2. if ANY two scores of the query tie, the permutation the reference gets
//      is whatever GNU libstdc++ std::sort (introsort: median-of-3 to first,
//      unguarded Hoare partition, threshold 16, depth limit 2*lg n with a
//      heapsort fallback, final insertion sort) leaves -- SURVEY.md Appendix A.
//      Lane 0 then re-ranks the query with a sequential emulation of exactly
//      that algorithm (ties dominate the first boosting iterations, where all
//      scores are 0 and the tie order decides the discounts).
//   3. metric of the current ranking (lane 0, same summation order as
//      dcg.cc:36-38) -- the training NDCG comes for free with the lambdas.
//   4. pair loop: lane <-> rank r2, loop r1 over the top-`cutoff` ranks;
//      the swap-delta is the closed form of ndcg.cc:76-88 (no n^2 jacobian),
//      rho/lambda/delta as lambdamart.cc:129-141; contributions to r1 are
//      reduced across the wave with a fixed shuffle tree (deterministic).
// exp() is OCML's f64 exp: lambdas agree with glibc to ~1e-16 relative, not
// bitwise (SURVEY.md section 7 hard part 3); log2 discounts come from a
// host-built table (glibc values) so the metric itself is bit-exact given the
// ranking.
#include "qr_internal.h"

#define NO_CUTOFF 0xFFFFFFFFu

// ---------------------------------------------------------------------------
// Sequential GNU std::sort emulation on an index array in LDS (lane 0 only).
// comp(a, b) := s[a] > s[b]   (queryresults.cc:42-44)
// ---------------------------------------------------------------------------
struct DescCmp {
  const double *s;
  __device__ __forceinline__ bool operator()(uint32_t a, uint32_t b) const {
    return s[a] > s[b];
  }
};

template <class C>
__device__ void g_push_heap(uint32_t *first, int hole, int top, uint32_t value, C c) {
  int parent = (hole - 1) / 2;
  while (hole > top && c(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

template <class C>
__device__ void g_adjust_heap(uint32_t *first, int hole, int len, uint32_t value, C c) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (c(first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  g_push_heap(first, hole, top, value, c);
}

template <class C>
__device__ void g_heapsort(uint32_t *first, int len, C c) {
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      const uint32_t v = first[parent];
      g_adjust_heap(first, parent, len, v, c);
      if (parent == 0) break;
      parent--;
    }
  }
  int last = len;
  while (last > 1) {
    --last;
    const uint32_t v = first[last];
    first[last] = first[0];
    g_adjust_heap(first, 0, last, v, c);
  }
}

template <class C>
__device__ __forceinline__ void g_linear_insert(uint32_t *a, int last, C c) {
  const uint32_t val = a[last];
  int next = last - 1;
  while (c(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}

template <class C>
__device__ void g_insertion_sort(uint32_t *a, int first, int last, C c) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (c(a[i], a[first])) {
      const uint32_t val = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];
      a[first] = val;
    } else
      g_linear_insert(a, i, c);
  }
}

template <class C>
__device__ void g_gnu_sort(uint32_t *a, int n, C c) {
  if (n == 0) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  // explicit stack replaces the recursion on [cut, last): the ranges are
  // disjoint, so the order in which they are processed does not matter.
  int st_first[64], st_last[64], st_depth[64];
  int sp = 0;
  st_first[0] = 0;
  st_last[0] = n;
  st_depth[0] = 2 * lg;
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = st_first[sp], last = st_last[sp], depth = st_depth[sp];
    while (last - first > 16) {
      if (depth == 0) {
        g_heapsort(a + first, last - first, c);
        break;
      }
      --depth;
      const int mid = first + (last - first) / 2;
      // __move_median_to_first(first, first+1, mid, last-1)
      {
        const int ia = first + 1, ib = mid, ic = last - 1;
        int pick;
        if (c(a[ia], a[ib])) {
          if (c(a[ib], a[ic]))
            pick = ib;
          else if (c(a[ia], a[ic]))
            pick = ic;
          else
            pick = ia;
        } else if (c(a[ia], a[ic]))
          pick = ia;
        else if (c(a[ib], a[ic]))
          pick = ic;
        else
          pick = ib;
        const uint32_t t = a[first];
        a[first] = a[pick];
        a[pick] = t;
      }
      // __unguarded_partition(first+1, last, pivot = first)
      int lo = first + 1, hi = last;
      const uint32_t pv = a[first];
      for (;;) {
        while (c(a[lo], pv)) ++lo;
        --hi;
        while (c(pv, a[hi])) --hi;
        if (!(lo < hi)) break;
        const uint32_t t = a[lo];
        a[lo] = a[hi];
        a[hi] = t;
        ++lo;
      }
      st_first[sp] = lo;
      st_last[sp] = last;
      st_depth[sp] = depth;
      ++sp;
      last = lo;
    }
  }
  if (n > 16) {
    g_insertion_sort(a, 0, 16, c);
    for (int i = 16; i < n; ++i) g_linear_insert(a, i, c);
  } else
    g_insertion_sort(a, 0, n, c);
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
  // fixed butterfly: identical association on every launch
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// pow(2.0, label) of dcg.cc:37 / ndcg.cc:80: exact for the integral relevance
// grades LETOR data carries.
__device__ __forceinline__ double pow2_label(float l) {
  const float t = truncf(l);
  if (t == l && fabsf(l) < 1000.f) return ldexp(1.0, (int)t);
  return pow(2.0, (double)l);
}

// mode 0: lambdas + metric, mode 1: metric only.
__global__ __launch_bounds__(64) void k_lambda(
    const double *__restrict__ scores, const float *__restrict__ labels,
    const uint32_t *__restrict__ qoff, int metric, uint32_t cutoff,
    const double *__restrict__ idcg, const double *__restrict__ lg2,
    double *__restrict__ lambda, double *__restrict__ weight,
    double *__restrict__ qmetric, uint32_t *__restrict__ ranks_out,
    double *__restrict__ ssq, QrScalars *__restrict__ scal, uint32_t nmax,
    uint32_t kacc, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t q = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const uint32_t off = qoff[q];
  const uint32_t n = qoff[q + 1] - off;
  double *s = reinterpret_cast<double *>(smem);          // [nmax] scores by doc
  double *accl = s + nmax;                               // [kacc]
  double *accw = accl + kacc;                            // [kacc]
  float *lab0 = reinterpret_cast<float *>(accw + kacc);  // [nmax] labels by doc
  float *sl = lab0 + nmax;                               // [nmax] labels by rank
  uint32_t *unmap = reinterpret_cast<uint32_t *>(sl + nmax);  // [nmax] pos_of_rank
  if (n == 0) {
    if (lane == 0) {
      qmetric[q] = 0.0;
      if (mode == 0 && ssq) ssq[2 * q] = ssq[2 * q + 1] = 0.0;
    }
    return;
  }
  for (uint32_t i = lane; i < n; i += 64) {
    s[i] = scores[off + i];
    lab0[i] = labels[off + i];
  }
  __syncthreads();
  // ---- 1. rank by counting + tie detection
  bool tie = false;
  for (uint32_t i = lane; i < n; i += 64) {
    const double si = s[i];
    uint32_t r = 0;
    for (uint32_t j = 0; j < n; ++j) {
      const double sj = s[j];
      const bool eq = (sj == si);
      r += (sj > si) || (eq && j < i);
      tie |= eq && (j != i);
    }
    unmap[r] = i;
  }
  const bool anytie = __any(tie);
  __syncthreads();
  // ---- 2. exact std::sort tie order
  if (anytie) {
    if (lane == 0) {
      for (uint32_t i = 0; i < n; ++i) unmap[i] = i;
      DescCmp c{s};
      g_gnu_sort(unmap, (int)n, c);
    }
    __syncthreads();
  }
  for (uint32_t r = lane; r < n; r += 64) {
    const uint32_t d = unmap[r];
    sl[r] = lab0[d];
    if (ranks_out) ranks_out[off + r] = d;
  }
  __syncthreads();
  const uint32_t size = cutoff < n ? cutoff : n;
  const double my_idcg = metric == QR_METRIC_NDCG ? idcg[q] : 1.0;
  // ---- 3. metric of the current ranking (dcg.cc:33-39, ndcg.cc:49-58)
  if (lane == 0) {
    double dcg = 0.0;
    for (uint32_t i = 0; i < size; ++i)
      dcg += (pow2_label(sl[i]) - 1.0) / lg2[i];
    double m = dcg;
    if (metric == QR_METRIC_NDCG) m = my_idcg > 0 ? dcg / my_idcg : 0.0;
    qmetric[q] = m;
  }
  if (mode == 1) return;
  // ---- 4. lambdas
  if (metric == QR_METRIC_NDCG && !(my_idcg > 0.0)) {
    // ndcg.cc:69-70: all-zero jacobian => lambdas and weights stay 0
    for (uint32_t i = lane; i < n; i += 64) {
      lambda[off + i] = 0.0;
      weight[off + i] = 0.0;
    }
    if (lane == 0 && ssq) ssq[2 * q] = ssq[2 * q + 1] = 0.0;
    return;
  }
  for (uint32_t i = lane; i < size; i += 64) {
    accl[i] = 0.0;
    accw[i] = 0.0;
  }
  __syncthreads();
  const uint32_t nbatch = (n + 63) / 64;
  for (uint32_t bt = 0; bt < nbatch; ++bt) {
    const uint32_t r2 = bt * 64 + lane;
    const bool live = r2 < n;
    const float l2 = live ? sl[r2] : 0.f;
    const double p2 = live ? pow2_label(l2) : 0.0;
    const double s2 = live ? s[unmap[r2]] : 0.0;
    const double inv2 = live ? 1.0 / lg2[r2] : 0.0;
    double al = 0.0, aw = 0.0;
    // r1 only needs to reach the ranks below the highest r2 of this batch
    const uint32_t r1_end = size < bt * 64 + 64 ? size : bt * 64 + 64;
    for (uint32_t r1 = 0; r1 < r1_end; ++r1) {
      const float l1 = sl[r1];
      double c1 = 0.0, cw = 0.0;
      if (live && r1 < r2 && l1 != l2) {
        const double p1 = pow2_label(l1);
        const double inv1 = 1.0 / lg2[r1];
        double j;
        if (r2 < size)
          j = (inv2 - inv1) * (p1 - p2);
        else
          j = (-inv1) * (p1 - p2);
        if (metric == QR_METRIC_NDCG) j = j / my_idcg;
        const double d = fabs(j);
        const double s1 = s[unmap[r1]];
        const bool hi1 = l1 > l2;  // the higher label plays "j" in lambdamart.cc:127
        const double diff = hi1 ? s1 - s2 : s2 - s1;
        const double rho = 1.0 / (1.0 + exp(diff));
        const double lam = rho * d;
        const double del = rho * (1.0 - rho) * d;
        c1 = hi1 ? lam : -lam;
        cw = del;
        al += hi1 ? -lam : lam;
        aw += del;
      }
      if (__any(c1 != 0.0 || cw != 0.0)) {
        const double t1 = wave_sum(c1);
        const double tw = wave_sum(cw);
        if (lane == 0) {
          accl[r1] += t1;
          accw[r1] += tw;
        }
      }
    }
    if (live) {
      const uint32_t d = off + unmap[r2];
      lambda[d] = al;
      weight[d] = aw;
    }
  }
  __syncthreads();
  // fold the top-`size` accumulators in (same lane wrote the own part: r % 64)
  double mx = 0.0, sq = 0.0, sm = 0.0;
  for (uint32_t r = lane; r < n; r += 64) {
    const uint32_t d = off + unmap[r];
    double l = lambda[d];
    if (r < size) {
      l += accl[r];
      lambda[d] = l;
      weight[d] += accw[r];
    }
    const double a = fabs(l);
    mx = a > mx ? a : mx;
    sq += l * l;
    sm += l;
  }
  mx = wave_max(mx);
  sq = wave_sum(sq);
  sm = wave_sum(sm);
  if (lane == 0) {
    atomicMax(&scal->maxabs_bits, (unsigned long long)__double_as_longlong(mx));
    if (ssq) {
      ssq[2 * q] = sq;
      ssq[2 * q + 1] = sm;
    }
  }
}

// Mart::compute_pseudoresponses (mart.cc:418-431): label - score
__global__ __launch_bounds__(256) void k_residual(const float *__restrict__ labels,
                                                  const double *__restrict__ scores,
                                                  double *__restrict__ out, uint32_t N,
                                                  double *__restrict__ ssq,
                                                  QrScalars *__restrict__ scal) {
  __shared__ double red[4], redm[4], reds[4];
  const uint32_t base = blockIdx.x * QR_SLICE;
  double sq = 0.0, mx = 0.0, sm = 0.0;
  for (uint32_t k = 0; k < QR_SLICE / 256; ++k) {
    const uint32_t i = base + k * 256 + threadIdx.x;
    if (i < N) {
      const double r = (double)labels[i] - scores[i];
      out[i] = r;
      sq += r * r;
      sm += r;
      const double a = fabs(r);
      mx = a > mx ? a : mx;
    }
  }
  sq = wave_sum(sq);
  sm = wave_sum(sm);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = sq;
    reds[threadIdx.x >> 6] = sm;
    redm[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ssq[2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    ssq[2 * blockIdx.x + 1] = (reds[0] + reds[1]) + (reds[2] + reds[3]);
    double m = redm[0];
    for (int i = 1; i < 4; ++i) m = redm[i] > m ? redm[i] : m;
    atomicMax(&scal->maxabs_bits, (unsigned long long)__double_as_longlong(m));
  }
}

// Fixed-order reduction of the per-query / per-slice partials + the
// quantisation scale for the histogram accumulators (one workgroup).
__global__ __launch_bounds__(1024) void k_prep(const double *__restrict__ ssq,
                                               uint32_t nss,
                                               const double *__restrict__ qmetric,
                                               uint32_t nq,
                                               QrScalars *__restrict__ scal) {
  __shared__ double red[16];
  double a = 0.0, b = 0.0, a2 = 0.0;
  for (uint32_t i = threadIdx.x; i < nss; i += 1024) {
    a += ssq[2 * i];
    a2 += ssq[2 * i + 1];
  }
  for (uint32_t i = threadIdx.x; i < nq; i += 1024) b += qmetric[i];
  a = wave_sum(a);
  a2 = wave_sum(a2);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  double ta = 0.0, ta2 = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < 16; ++i) ta += red[i];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a2;
  __syncthreads();
  if (threadIdx.x == 0)
    for (int i = 0; i < 16; ++i) ta2 += red[i];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tb = 0.0;
    for (int i = 0; i < 16; ++i) tb += red[i];
    if (ssq) {
      scal->root_ss = ta;
      scal->root_sum = ta2;
    }
    if (qmetric) scal->metric_sum = tb;
    if (ssq) {
      const double mx = __longlong_as_double((long long)scal->maxabs_bits);
      int x = 0;
      if (mx > 0.0) frexp(mx, &x);  // mx = m * 2^x, m in [0.5, 1)  =>  mx < 2^x
      const int e = QR_QBITS - x;
      scal->scale_exp = e;
      scal->scale = ldexp(1.0, e);
      scal->inv_scale = ldexp(1.0, -e);
    }
  }
}

// ---------------------------------------------------------------------------
static size_t lambda_lds(size_t nmax, size_t kacc) {
  return nmax * 8 + kacc * 16 + nmax * 12;
}

int qr_k_lambda(qr_ctx *c, int which, int metric, size_t cutoff, int mode) {
  const size_t Q = which ? c->vQ : c->Q;
  const size_t maxq = which ? c->vmaxq : c->maxq;
  if (Q == 0) return QR_OK;
  const uint32_t cut = cutoff == 0 ? NO_CUTOFF : (uint32_t)cutoff;
  size_t kacc = cutoff == 0 || cutoff > maxq ? maxq : cutoff;
  if (kacc == 0) kacc = 1;
  const size_t nmax = (maxq + 1) & ~(size_t)1;
  kacc = (kacc + 1) & ~(size_t)1;
  const size_t lds = lambda_lds(nmax, kacc);
  if (lds > 160 * 1024 - 512)
    QR_FAIL(c, QR_ERR_UNSUPPORTED,
            "query too long for the LDS-resident lambda kernel (max ~7000 docs "
            "with a cutoff, ~4400 without)");
  if (lds > 64 * 1024)
    QR_CHECK(c, hipFuncSetAttribute((const void *)k_lambda,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
  if (which == 0) {
    hipLaunchKernelGGL(k_lambda, dim3((unsigned)Q), dim3(64), lds, c->stream,
                       c->d_scores, c->d_labels, c->d_qoff, metric, cut, c->d_idcg,
                       c->d_lg2, c->d_lambda, c->d_weight, c->d_qmetric, c->d_ranks,
                       mode == 0 ? c->d_ssq : nullptr, c->d_scalars, (uint32_t)nmax,
                       (uint32_t)kacc, mode);
  } else {
    hipLaunchKernelGGL(k_lambda, dim3((unsigned)Q), dim3(64), lds, c->stream,
                       c->d_vscores, c->d_vlabels, c->d_vqoff, metric, cut,
                       c->d_vidcg, c->d_lg2, (double *)nullptr, (double *)nullptr,
                       c->d_vqmetric, (uint32_t *)nullptr, (double *)nullptr,
                       c->d_scalars, (uint32_t)nmax, (uint32_t)kacc, 1);
  }
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_residual(qr_ctx *c) {
  const unsigned grid = (unsigned)((c->N + QR_SLICE - 1) / QR_SLICE);
  hipLaunchKernelGGL(k_residual, dim3(grid), dim3(256), 0, c->stream, c->d_labels,
                     c->d_scores, c->d_lambda, (uint32_t)c->N, c->d_ssq,
                     c->d_scalars);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// nss > 0: reduce ssq[nss] into root_ss and derive the scale; the per-query
// metric of set `which` (encoded in the sign: nss == 0 means metric only).
int qr_k_prep(qr_ctx *c, size_t nss) {
  hipLaunchKernelGGL(k_prep, dim3(1), dim3(1024), 0, c->stream,
                     nss ? c->d_ssq : (const double *)nullptr, (uint32_t)nss,
                     (const double *)nullptr, 0u, c->d_scalars);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_metric_reduce(qr_ctx *c, int which) {
  hipLaunchKernelGGL(k_prep, dim3(1), dim3(1024), 0, c->stream,
                     (const double *)nullptr, 0u,
                     which ? c->d_vqmetric : c->d_qmetric,
                     (uint32_t)(which ? c->vQ : c->Q), c->d_scalars);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}
