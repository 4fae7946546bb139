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
//      The wave then re-ranks the query with a wave-parallel emulation of
//      exactly that algorithm (wave_gnu_sort below; ties dominate the first
//      boosting iterations, where all scores are 0 and the tie order decides the
//      discounts, and never fully disappear).  Only the ranks the metric can see are
//      ordered exactly -- the first `cutoff`; see `limit` at wave_gnu_sort.
//   3. metric of the current ranking (lane 0, same summation order as
//      dcg.cc:36-38) -- the training NDCG comes for free with the lambdas.
//   4. pair loop: lane <-> rank r2, loop r1 over the top-`cutoff` ranks;
//      the swap-delta is the closed form of ndcg.cc:76-88 (no n^2 jacobian),
//      rho/lambda/delta as lambdamart.cc:129-141; contributions to r1 are
//      reduced across the wave with a fixed shuffle tree (deterministic).
// exp() is a table + degree-5 polynomial f64 exp (qr_exp, ~1 ulp): lambdas agree
// with glibc to ~1e-15 relative, not bitwise (SURVEY.md section 7 hard part 3); log2 discounts come from a
// host-built table (glibc values) so the metric itself is bit-exact given the
// ranking.
#include <hip/hip_ext.h>

#include <algorithm>

#include "qr_internal.h"
#include "qr_wave.h"
#include "qr_prep.h"

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

struct PackedCmp {  // comp on packed (key << 16 | doc): key = #strictly greater scores
  __device__ __forceinline__ bool operator()(uint32_t a, uint32_t b) const {
    return (a >> 16) < (b >> 16);
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
// exp(x) in f64: x = k*ln2/64 + r, |r| <= ln2/128;  exp(x) = 2^(k>>6) * T[k&63] *
// (1 + r + r^2/2 + ... + r^5/120), T[j] = 2^(j/64) correctly rounded.  About a
// dozen f64 operations and ~1 ulp (the reference tolerance is 1e-5; OCML's exp
// costs ~10x more instructions and was half of the kernel's time).
// ---------------------------------------------------------------------------
__device__ const double QR_EXP_T[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};

__device__ __forceinline__ double qr_exp(double x, const double *T) {
  x = fmin(x, 709.7);   // (one v_min / v_max each; the arguments are never NaN)
  x = fmax(x, -745.1);
  const double kd = rint(x * 92.33248261689366);                   // 64 / ln2
  const int k = (int)kd;
  double r = fma(-kd, 0x1.62e42fef00000p-7, x);                    // ln2/64, high part
  r = fma(-kd, 1.162596423439437e-12, r);                          //         low part
  double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  const double e = fma(p, r * r, r);                               // exp(r) - 1
  const double t = T[k & 63];
  return ldexp(fma(t, e, t), k >> 6);
}

// 1 / x for x >= 1 (the logistic's denominator): hardware seed + two Newton steps, a few
// ulp -- the lambdas' bar is 1e-5 (north_star), the tests ask 1e-11; a correctly rounded
// f64 division costs four times the instructions
__device__ __forceinline__ double qr_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// Batched wave reductions of the pair sweep (k_lambda, queries of up to 128 documents).
// swap32_add(a, b): lanes 0..31 get a[l] + a[l + 32], lanes 32..63 get b[l - 32] + b[l] -- one
// v_permlane32_swap per dword exchanges a's upper half with b's lower half.
__device__ __forceinline__ double swap32_add(const double a, const double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)__double2loint(a), (uint32_t)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)__double2hiint(a), (uint32_t)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// swap16_add(a, b): even 16-lane rows get a[row] + a[row + 1], odd rows b[row - 1] + b[row]
// (v_permlane16_swap: a's odd rows against b's even rows)
__device__ __forceinline__ double swap16_add(const double a, const double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)__double2loint(a), (uint32_t)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)__double2hiint(a), (uint32_t)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// every lane its 16-lane row's total (the first four steps of wave_sum)
__device__ __forceinline__ double row_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return v;
}

// One value per lane, sorted in DESCENDING order over the wave's lanes (bitonic network on
// lane exchanges; max / min keep the multiset whatever the signs of zero).
// K chunks at once: a stage's lane exchanges are LDS-pipe operations of ~100 cycles each and every
// stage waits for the one before it -- four chunks' networks side by side hide three quarters of
// that (a 200-document query's counting rank: 27.9 k -> see profiles/r05_lambda_ragged.md).
template <int K>
__device__ __forceinline__ void wave_sort_desc(double (&v)[K], const uint32_t lane) {
#pragma unroll
  for (uint32_t k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      double o[K];
#pragma unroll
      for (int c = 0; c < K; ++c) {
        const int lo = __shfl_xor(__double2loint(v[c]), (int)j), hi = __shfl_xor(__double2hiint(v[c]), (int)j);
        o[c] = __hiloint2double(hi, lo);
      }
      // descending blocks (lane & k) == 0 keep the larger value in the lower lane
      const bool take_max = ((lane & j) == 0) == ((lane & k) == 0);
#pragma unroll
      for (int c = 0; c < K; ++c) v[c] = take_max ? fmax(v[c], o[c]) : fmin(v[c], o[c]);
    }
}

// pow(2.0, label) of dcg.cc:37 / ndcg.cc:80: exact for the integral relevance
// grades LETOR data carries.
__device__ __forceinline__ double pow2_label(float l) {
  const float t = truncf(l);
  if (t == l && fabsf(l) < 1000.f) return ldexp(1.0, (int)t);
  return pow(2.0, (double)l);
}

// ---------------------------------------------------------------------------
// Wave-parallel emulation of GNU std::sort on a packed (key << 16 | doc) array
// in LDS, key = number of strictly greater scores, so that
//     comp(a, b) = score[a] > score[b]  <=>  key(a) < key(b).
// The permutation a sort leaves depends only on its comparison outcomes, so the
// emulation may run the same algorithm with a different schedule:
//  * introsort loop: ranges are processed from an explicit stack (disjoint
//    ranges commute); median-of-3 and the swap to `first` are uniform scalar
//    work; the unguarded Hoare partition is done by the whole wave -- the
//    sequential scan pairs the k-th position from the left with key >= pivot
//    ("LB") with the k-th position from the right with key <= pivot ("RB") and
//    swaps them while LB[k] < RB[k]; both lists and the number of swaps K come
//    from ballots/prefix counts, and the returned cut is
//    K ? min(LB[K], RB[K-1]) : LB[0]   (the left scan stops at the next
//    untouched >= pivot position or at the last swapped-in element).
//  * depth limit 2*lg(n): the heapsort fallback runs sequentially on lane 0
//    (only adversarial inputs reach it).
//  * final insertion sort: linear insertion with a strict comparison never
//    moves an element past an equal one, so it equals a STABLE sort by key of
//    the arrangement the partition phase left -- done by counting.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk(uint32_t v) { return v >> 16; }

//  * a range whose keys are pairwise distinct is left alone: whatever arrangement
//    the partition phase would give it, the final stable sort by key puts its
//    elements in the same places (ranges are ordered by key among themselves, so
//    an equal key in ANOTHER range is on a known side).  `dupk[key]` != 0 marks the
//    keys that occur more than once; with a few tied pairs in a query only the
//    ranges on the way down to them are partitioned.
// The partition phase is ONE wave's work.  BS = true: the workgroup is that wave (its
// __syncthreads are wave barriers); BS = false: wave 0 of a larger workgroup runs it alone
// while the others wait at the caller's barrier -- inside a wave the LDS operations are
// already in program order, so a compiler-level fence is all that is needed.
template <bool BS>
__device__ __forceinline__ void sort_sync() {
  if (BS) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

//  * `limit`: only the first `limit` positions of the result are asked for (the metric's
//    cutoff: ranks beyond it carry no discount, so neither the metric nor a lambda depends
//    on the ORDER of the documents ranked there -- only on which documents they are).  The
//    partition phase leaves the ranges ordered by key among themselves, and the final
//    stable sort by key never moves an element out of its range; so position q of the
//    result comes from the range that covers q, and a range that starts at or beyond
//    `limit` need not be partitioned at all.  limit = n gives the whole permutation.
template <bool BS>
__device__ void wave_gnu_sort(uint32_t *a, const int n, uint32_t *LB, uint32_t *RB,
                              int *stk, const uint8_t *dupk, const int limit) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  if (n > 16) {
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    int sp = 0;
    if (lane == 0) {
      stk[0] = 0;
      stk[1] = n;
      stk[2] = 2 * lg;
    }
    sp = 1;
    sort_sync<BS>();
    while (sp > 0) {
      --sp;
      int first = stk[3 * sp], last = stk[3 * sp + 1], depth = stk[3 * sp + 2];
      sort_sync<BS>();
      while (last - first > 16) {
        {
          bool dup = false;
          for (int x = first + lane; x < last; x += 64) dup |= dupk[pk(a[x])] != 0;
          if (!__any(dup)) break;
        }
        if (depth == 0) {
          if (lane == 0) g_heapsort(a + first, last - first, PackedCmp());
          sort_sync<BS>();
          break;
        }
        --depth;
        // __move_median_to_first(first, first+1, mid, last-1): uniform
        uint32_t p;  // the pivot's key (the element that ends up at `first`)
        {
          const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
          const uint32_t ka = pk(a[ia]), kb = pk(a[ib]), kc = pk(a[ic]);
          int pick;
          if (ka < kb) {
            if (kb < kc) pick = ib;
            else if (ka < kc) pick = ic;
            else pick = ia;
          } else if (ka < kc) pick = ia;
          else if (kb < kc) pick = ic;
          else pick = ib;
          p = pick == ia ? ka : (pick == ib ? kb : kc);
          sort_sync<BS>();
          if (lane == 0) {
            const uint32_t t = a[first];
            a[first] = a[pick];
            a[pick] = t;
          }
          sort_sync<BS>();
        }
        const int lo0 = first + 1;
        int nl = 0, nr = 0, K = 0;
        uint32_t cut;
        if (last - lo0 <= 128) {
          // Ranges of at most two 64-lane chunks (every range of a 100-document
          // query): each position is read once, the four ballots stay in scalar
          // registers, and the lanes keep their (LB[k], RB[k]) pair for the swap.
          const unsigned long long gt = lane == 63 ? 0ull : (~0ull << (lane + 1));
          const int x0 = lo0 + lane, x1 = lo0 + 64 + lane;
          const bool in0 = x0 < last, in1 = x1 < last;
          const uint32_t k0 = pk(a[in0 ? x0 : first]), k1 = pk(a[in1 ? x1 : first]);
          const bool lb0 = in0 && !(k0 < p), lb1 = in1 && !(k1 < p);  // key >= p
          const bool rb0 = in0 && !(p < k0), rb1 = in1 && !(p < k1);  // key <= p
          const unsigned long long mlb0 = __ballot(lb0), mlb1 = __ballot(lb1);
          const unsigned long long mrb0 = __ballot(rb0), mrb1 = __ballot(rb1);
          nl = __popcll(mlb0) + __popcll(mlb1);
          nr = __popcll(mrb0) + __popcll(mrb1);
          // LB ascending, RB descending (index = flagged positions to my right)
          if (lb0) LB[__popcll(mlb0 & lt)] = (uint32_t)x0;
          if (lb1) LB[__popcll(mlb0) + __popcll(mlb1 & lt)] = (uint32_t)x1;
          if (rb0) RB[__popcll(mrb1) + __popcll(mrb0 & gt)] = (uint32_t)x0;
          if (rb1) RB[__popcll(mrb1 & gt)] = (uint32_t)x1;
          sort_sync<BS>();
          const int np = nl < nr ? nl : nr;
          const int q0 = lane, q1 = lane + 64;
          const uint32_t L0 = q0 < np ? LB[q0] : 0u, R0 = q0 < np ? RB[q0] : 0u;
          const uint32_t L1 = q1 < np ? LB[q1] : 0u, R1 = q1 < np ? RB[q1] : 0u;
          const bool v0 = q0 < np && L0 < R0, v1 = q1 < np && L1 < R1;
          const unsigned long long mv0 = __ballot(v0), mv1 = __ballot(v1);
          // the pairs cross once: true ... true false ... false
          K = mv0 == ~0ull ? 64 + __popcll(mv1) : __popcll(mv0);
          if (K > 0) {
            const uint32_t c1 = K < nl ? LB[K] : 0xFFFFFFFFu;
            const uint32_t c2 = RB[K - 1];
            cut = c1 < c2 ? c1 : c2;
          } else
            cut = nl > 0 ? LB[0] : (uint32_t)last;
          if (v0) {
            const uint32_t t = a[L0];
            a[L0] = a[R0];
            a[R0] = t;
          }
          if (v1 && mv0 == ~0ull) {
            const uint32_t t = a[L1];
            a[L1] = a[R1];
            a[R1] = t;
          }
          sort_sync<BS>();
        } else {
          // LB: ascending positions with key >= p
          for (int base = lo0; base < last; base += 64) {
            const int x = base + lane;
            const bool in = x < last;
            const bool lb = in && !(pk(a[in ? x : first]) < p);
            const unsigned long long m = __ballot(lb);
            if (lb) LB[nl + __popcll(m & lt)] = (uint32_t)x;
            nl += __popcll(m);
          }
          // RB: descending positions with key <= p
          for (int top = last; top > lo0; top -= 64) {
            const int x = top - 1 - lane;
            const bool in = x >= lo0;
            const bool rb = in && !(p < pk(a[in ? x : first]));
            const unsigned long long m = __ballot(rb);
            if (rb) RB[nr + __popcll(m & lt)] = (uint32_t)x;
            nr += __popcll(m);
          }
          sort_sync<BS>();
          const int np = nl < nr ? nl : nr;
          for (int base = 0; base < np; base += 64) {
            const int k = base + lane;
            const bool v = k < np && LB[k] < RB[k];
            const unsigned long long m = __ballot(v);
            K += __popcll(m);
            if (m != ~0ull) break;
          }
          if (K > 0) {
            const uint32_t c1 = K < nl ? LB[K] : 0xFFFFFFFFu;
            const uint32_t c2 = RB[K - 1];
            cut = c1 < c2 ? c1 : c2;
          } else
            cut = nl > 0 ? LB[0] : (uint32_t)last;
          sort_sync<BS>();
          for (int k = lane; k < K; k += 64) {
            const uint32_t x = LB[k], y = RB[k];
            const uint32_t t = a[x];
            a[x] = a[y];
            a[y] = t;
          }
          sort_sync<BS>();
        }
        // recurse on [cut, last) (pushed, if any of it is asked for), loop on [first, cut)
        if ((int)cut < limit) {
          if (lane == 0) {
            stk[3 * sp] = (int)cut;
            stk[3 * sp + 1] = last;
            stk[3 * sp + 2] = depth;
          }
          ++sp;
        }
        last = (int)cut;
        sort_sync<BS>();
      }
    }
  }
}

// The final insertion sort of std::sort == a stable sort by key of the arrangement the
// partition phase left, by every thread of the workgroup (T = its size).
// PACKED: the query is ONE wave's work inside a larger workgroup (k_lambda_u): `tid` is the lane,
// and the wave's own LDS operations are in program order (qsync<true> = a fence, no s_barrier).
template <bool PACKED>
__device__ __forceinline__ void qsync() {
  if (PACKED) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  } else
    __syncthreads();
}

template <int W, bool PACKED>
__device__ __forceinline__ void sort_placement(const uint32_t *a, const int n, uint32_t *out,
                                               const uint8_t *dupk, uint32_t *cnt, uint32_t *pos,
                                               const uint32_t tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  // final insertion sort == stable sort by key of the current arrangement
  // (a[n .. n4) is padded with key 0xFFFF by the caller: never counted)
  // The key IS the number of strictly greater scores, so the elements of key k end
  // up in positions k, k+1, ...: only the equal-key elements standing before x in
  // the current arrangement have to be counted.
  // A key that occurs once is its own position.  The duplicated keys are handled one
  // distinct value at a time (uniform loop): the ballots of "my key == k" over the 64-
  // position chunks give every holder the number of equal keys standing before it.
  // Early boosting iterations have a handful of distinct scores per query, later ones
  // few duplicates: either way far fewer steps than comparing every pair of positions.
  if (n <= 128) {
    if (wave != 0) return;
    const int x0 = lane, x1 = lane + 64;
    const bool in0 = x0 < n, in1 = x1 < n;
    const uint32_t v0 = in0 ? a[x0] : 0xFFFFFFFFu, v1 = in1 ? a[x1] : 0xFFFFFFFFu;
    const uint32_t k0 = pk(v0), k1 = pk(v1);
    uint32_t r0 = k0, r1 = k1;
    unsigned long long todo0 = __ballot(in0 && dupk[in0 ? k0 : 0] != 0);
    unsigned long long todo1 = __ballot(in1 && dupk[in1 ? k1 : 0] != 0);
    while (todo0 | todo1) {
      const int src = todo0 ? __ffsll((long long)todo0) - 1 : __ffsll((long long)todo1) - 1;
      const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)(todo0 ? k0 : k1), src);
      const unsigned long long m0 = __ballot(in0 && k0 == k), m1 = __ballot(in1 && k1 == k);
      if (in0 && k0 == k) r0 = k + (uint32_t)__popcll(m0 & lt);
      if (in1 && k1 == k) r1 = k + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1 & lt);
      todo0 &= ~m0;
      todo1 &= ~m1;
    }
    if (in0) out[r0] = v0 & 0xFFFFu;
    if (in1) out[r1] = v1 & 0xFFFFu;
  } else {
    // Long queries: the holders of a duplicated key k take the slots of pos[k .. k + c_k)
    // in whatever order their atomics land, and the ORDER is then restored from the
    // arrangement positions stored there -- deterministic whatever the atomics did.  A
    // small group (fewer than 64 holders) is ranked by its own members (each counts the
    // members standing before it); the few big ones (at most n / 64 of them) by one sweep
    // of ballots over the arrangement each, dealt to the waves in turn.  Either way a few
    // thousand instructions per wave, where comparing every holder with every position
    // took 100 us on a 1200-document query of tied scores.
    constexpr int T = 64 * W;
    for (int x = (int)tid; x < n; x += T) cnt[x] = 0;
    qsync<PACKED>();
    for (int x = (int)tid; x < n; x += T) {
      const uint32_t v = a[x], k = pk(v);
      if (!dupk[k])
        out[k] = v & 0xFFFFu;
      else
        pos[k + atomicAdd(&cnt[k], 1u)] = (uint32_t)x;
    }
    qsync<PACKED>();
    for (int x = (int)tid; x < n; x += T) {
      const uint32_t v = a[x], k = pk(v);
      if (!dupk[k]) continue;
      const uint32_t c = cnt[k];
      if (c >= 64) continue;
      uint32_t r = k;
      for (uint32_t t = 0; t < c; ++t) r += pos[k + t] < (uint32_t)x;
      out[r] = v & 0xFFFFu;
    }
    int j = 0;
    for (int kb = 0; kb < n; kb += 64) {
      unsigned long long km = __ballot(kb + lane < n && cnt[kb + lane < n ? kb + lane : 0] >= 64);
      while (km) {
        const int bit = __ffsll((long long)km) - 1;
        km &= km - 1ull;
        if ((j++ % W) != wave) continue;
        const uint32_t k = (uint32_t)(kb + bit);
        uint32_t before = 0;
        for (int base = 0; base < n; base += 64) {
          const int x = base + lane;
          const uint32_t v = x < n ? a[x] : 0xFFFFFFFFu;
          const bool e = pk(v) == k;
          const unsigned long long m = __ballot(e);
          if (e) out[k + before + (uint32_t)__popcll(m & lt)] = v & 0xFFFFu;
          before += (uint32_t)__popcll(m);
        }
      }
    }
  }
}

// mode 0: lambdas + metric, mode 1: metric only.
// lg2[r] = log2(r + 2) and ilg2[r] = 1.0 / lg2[r] are host-built tables (glibc
// log2, IEEE division): the same values the reference computes inline.
// LONG = false: one workgroup per query, working set in LDS; queries flagged in
// `long_flag` (too long for the LDS) are left to the LONG = true launch, which
// runs the same code on a per-query slice of a global scratch buffer
// (`long_list[blockIdx.x]` names the query) -- slow, but any length up to 65535.
// W = waves per workgroup: 1 for queries of up to 256 documents, 4 up to 512, 16 beyond
// (the launch's size class decides), whose O(n^2 / 64) phases -- the counting rank, the
// pair sweep -- and the placement of tied keys are then shared by the waves (the partition
// phase of the sort emulation stays one wave's work).  The per-rank sums of a W > 1 query
// are added per wave and then pairwise over the waves in a fixed order: deterministic, and
// within an ulp or two of the one-wave order.
// One query: `smem` its working set (LDS, or a global scratch slice), `tid` the thread's
// index among the 64 * W threads that work on it.  PACKED (W == 1 only): the query is one
// wave's work inside a larger workgroup -- no s_barrier anywhere on its way (qsync).
constexpr int QR_LAMBDA_RR = 5;  // ranks r1 per round of the pair sweep (their sums are reduced together)
struct QrLambdaArgs {
  double *scores;  // (written too when a score update rides in this pass: upd_leaf below)
  const float *labels;
  const uint32_t *qoff;
  int metric;
  uint32_t cutoff;
  const double *idcg, *lg2, *ilg2;
  double *lambda, *weight, *qmetric;
  uint32_t *ranks_out;
  double *ssq, *qmax;
  unsigned long long *qslot;
  int mode;
  const uint8_t *present;
  int exact_tail;
  // Mart::update_modelscores of the tree before (mart.cc:464-467), folded into this pass's
  // load (lambdamart.cc:70 reads the scores it would have written): score += shrinkage *
  // leaf_value[leaf of the document], the same two operations as k_score_update_leaf.
  // upd_leaf == nullptr: the scores are up to date.
  const uint8_t *upd_leaf;
  const double *upd_value;
  double upd_shrinkage;
};

// SMALL: every query of the launch has at most 128 documents -- the paths of longer queries
// (sorted-chunk rank, the pair sweep's rounds over LDS) are compiled out, and with them their
// registers: 72 VGPRs instead of 105, seven waves per SIMD instead of four (the uniform bench
// set's launch: 45.6 us, 51.2 with the long paths compiled in).
template <int W, bool PACKED, bool SMALL = false>
__device__ __forceinline__ void lambda_query(const QrLambdaArgs &A, const uint32_t q, char *smem,
                                             const uint32_t nmax, const uint32_t kacc, const uint32_t tid,
                                             double (*sh_part)[W][2 * QR_LAMBDA_RR], double (*sh_red)[3],
                                             double *expt_shared) {
  static_assert(!PACKED || W == 1, "a packed query is one wave's");
  double *scores = A.scores;
  const float *__restrict__ labels = A.labels;
  const uint32_t *__restrict__ qoff = A.qoff;
  const int metric = A.metric;
  const uint32_t cutoff = A.cutoff;
  const double *__restrict__ idcg = A.idcg;
  const double *__restrict__ lg2 = A.lg2;
  const double *__restrict__ ilg2 = A.ilg2;
  double *__restrict__ lambda = A.lambda;
  double *__restrict__ weight = A.weight;
  double *__restrict__ qmetric = A.qmetric;
  uint32_t *__restrict__ ranks_out = A.ranks_out;
  double *__restrict__ ssq = A.ssq;
  double *__restrict__ qmax = A.qmax;
  unsigned long long *__restrict__ qslot = A.qslot;
  const int mode = A.mode;
  const uint8_t *__restrict__ present = A.present;
  const int exact_tail = A.exact_tail;
  constexpr uint32_t T = 64 * W;
  const uint32_t lane = tid & 63, wave = tid >> 6;
#ifdef QR_LAMBDA_TIMING
  long long tq[8];
  tq[0] = clock64();
#define QR_T(i) tq[i] = clock64()
#elif defined(QR_LAMBDA_STOP)  // ablation builds (scripts/lambda_ablation.sh): leave after section i
#define QR_T(i) if (QR_LAMBDA_STOP == (i)) return
#else
#define QR_T(i)
#endif
#ifdef QR_LAMBDA_STOP  // a section boundary of the ablation builds only
#define QR_TS(i) QR_T(i)
#else
#define QR_TS(i)
#endif
  const uint32_t off = qoff[q];
  const uint32_t n_full = qoff[q + 1] - off;
  uint32_t n = n_full;
  double *s = reinterpret_cast<double *>(smem);          // [nmax] scores by doc
  double *sr = s + nmax;                                 // [nmax] scores by rank
  double *accl = sr + nmax;                              // [kacc] contributions to top ranks
  double *accw = accl + kacc;                            // [kacc]
  double *ownl = accw + kacc;                            // [nmax] own accumulators by rank
  double *ownw = ownl + nmax;                            // [nmax]
  float *lab0 = reinterpret_cast<float *>(ownw + nmax);  // [nmax] labels by doc
  float *sl = lab0 + nmax;                               // [nmax] labels by rank
  uint32_t *unmap = reinterpret_cast<uint32_t *>(sl + nmax);  // [nmax] pos_of_rank
  // sort scratch aliases the own accumulators (16 * nmax bytes, used before them)
  uint32_t *pa = reinterpret_cast<uint32_t *>(ownl);     // packed (key << 16 | doc)
  uint32_t *LB = pa + nmax;
  uint32_t *RB = LB + nmax;
  int *stk = reinterpret_cast<int *>(unmap + nmax);      // [3 * 40]: at most 2 lg n + 1 ranges wait
  double *ilt = reinterpret_cast<double *>(stk + 3 * 40);  // [kacc] 1/log2(r+2), top ranks
  // [64] 2^(j/64): the query's own copy, or (packed queries) one table for the workgroup
  double *expt = expt_shared ? expt_shared : ilt + kacc;
  uint8_t *dupk = reinterpret_cast<uint8_t *>(expt_shared ? ilt + kacc : expt + 64);  // [nmax, padded to 8] key occurs more than once
  // [nmax] cleaned -> original doc; only there when a sample is drawn (`present`)
  uint32_t *cmap = reinterpret_cast<uint32_t *>(dupk + ((nmax + 7) & ~7u));
  // --subsample (lambdamart.cc:85-102): the query is "cleaned" of the documents
  // that are not in this iteration's sample; everything below then runs on the
  // cleaned list, in its own numbering, exactly as on a shorter query.  Documents
  // outside the sample get lambda = weight = 0.
  if (present) {
    uint32_t cnt = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t base = 0; base < n_full; base += 64) {
      const uint32_t i = base + lane;
      const bool in = i < n_full && present[off + i] != 0;
      const unsigned long long m = __ballot(in);
      if (in) cmap[cnt + __popcll(m & lt)] = i;
      if (i < n_full && !in && mode == 0) {
        lambda[off + i] = 0.0;
        weight[off + i] = 0.0;
      }
      cnt += __popcll(m);
    }
    n = cnt;
    qsync<PACKED>();
  }
  if (n == 0) {
    if (tid == 0) {
      qmetric[q] = 0.0;
      if (mode == 0 && ssq) ssq[2 * q] = ssq[2 * q + 1] = qmax[q] = 0.0;
    }
    return;
  }
  if (A.upd_leaf) {  // (never with a sample: every document of the query is here)
    for (uint32_t i = tid; i < n; i += T) {
      const double v = scores[off + i] + A.upd_shrinkage * A.upd_value[A.upd_leaf[off + i]];
      scores[off + i] = v;
      s[i] = v;
      lab0[i] = labels[off + i];
    }
  } else
  for (uint32_t i = tid; i < n; i += T) {
    const uint32_t di = present ? cmap[i] : i;
    s[i] = scores[off + di];
    lab0[i] = labels[off + di];
  }
  // NaN padding to a multiple of 4: compares false, so it never counts
  const uint32_t n4 = (n + 3) & ~3u;
  if (tid < n4 - n) s[n + tid] = __longlong_as_double(0x7ff8000000000000LL);
  qsync<PACKED>();
  QR_T(1);
  // ---- 1. rank by counting: g = #docs with a strictly greater score.  Two docs
  //         per lane per sweep share the broadcast read of s[j]; one f64 compare
  //         and one add per (j, doc).  Without ties g is a permutation of 0..n-1;
  //         a tie makes two docs collide on the same slot, which the check below
  //         detects (the loser does not find itself in unmap[g]).
  for (uint32_t r = tid; r < n; r += T) {
    unmap[r] = 0xFFFFFFFFu;
    dupk[r] = 0;
  }
  qsync<PACKED>();
  bool tie = false;
  if (!SMALL && n > 128) {
    // Longer queries: the same count from sorted chunks.  Every 64 scores are sorted in a
    // wave's registers (descending; the scores by rank are not live yet: `sr` holds the chunks),
    // and a document counts the scores above its own chunk by chunk with a binary search --
    // seven probes per chunk instead of sixty-four compares.  g is an exact integer either way.
    double *srt = sr;
    const uint32_t nch = (n + 63) >> 6;
    if (W == 1) {  // (one wave: four chunks' networks side by side)
      for (uint32_t c0 = 0; c0 < nch; c0 += 4) {
        double v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          const uint32_t i = (c0 + u) * 64 + lane;
          v[u] = i < n ? s[i] : -__builtin_inf();
        }
        wave_sort_desc<4>(v, lane);
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          const uint32_t i = (c0 + u) * 64 + lane;
          if (i < n) srt[i] = v[u];  // (the padding sorts to the chunk's end: beyond n)
        }
      }
    } else {
      for (uint32_t c0 = wave; c0 < nch; c0 += 2 * W) {  // (a wave's chunks: c0 and c0 + W together)
        double v[2];
#pragma unroll
        for (uint32_t u = 0; u < 2; ++u) {
          const uint32_t i = (c0 + u * W) * 64 + lane;
          v[u] = i < n ? s[i] : -__builtin_inf();
        }
        wave_sort_desc<2>(v, lane);
#pragma unroll
        for (uint32_t u = 0; u < 2; ++u) {
          const uint32_t i = (c0 + u * W) * 64 + lane;
          if (i < n) srt[i] = v[u];
        }
      }
    }
    qsync<PACKED>();
    for (uint32_t i = tid; i < n; i += T) {
      const double a = s[i];
      uint32_t g = 0;
      // (four chunks side by side: a probe waits for the one before it in ITS chunk only; a
      // chunk beyond the query has no scores above anything)
      constexpr uint32_t G = 4;
      for (uint32_t c0 = 0; c0 < nch; c0 += G) {
        uint32_t pos[G];  // the chunk's scores [0, pos) are above a
#pragma unroll
        for (uint32_t u = 0; u < G; ++u) pos[u] = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) {
          double e[G];
#pragma unroll
          for (uint32_t u = 0; u < G; ++u) {
            const uint32_t t = (c0 + u) * 64 + pos[u] + step;  // (position + 1 in the query)
            e[u] = t <= n ? srt[t - 1] : a;
          }
#pragma unroll
          for (uint32_t u = 0; u < G; ++u) pos[u] += e[u] > a ? step : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < G; ++u) {
          const uint32_t t = (c0 + u) * 64 + 64;
          if (pos[u] == 63 && t <= n && srt[t - 1] > a) pos[u] = 64;
          g += pos[u];
        }
      }
      unmap[g] = i;
      pa[i] = (g << 16) | i;
    }
  } else
  for (uint32_t ib = tid; ib < n; ib += 2 * T) {
    const uint32_t i0 = ib, i1 = ib + T;
    const bool has1 = i1 < n;
    const double a0 = s[i0], a1 = has1 ? s[i1] : 0.0;
    uint32_t g0 = 0, g1 = 0;
    if (W == 1 || __any(has1)) {
#pragma unroll 4
      for (uint32_t j = 0; j < n4; ++j) {
        const double sj = s[j];
        g0 += sj > a0;
        g1 += sj > a1;
      }
    } else {  // (a wave of a long query's last sweep: one document per lane)
#pragma unroll 4
      for (uint32_t j = 0; j < n4; ++j) g0 += s[j] > a0;
    }
    unmap[g0] = i0;            // exact whenever no two scores tie
    pa[i0] = (g0 << 16) | i0;  // identity arrangement, as queryresults.cc:50-51
    if (has1) {
      unmap[g1] = i1;
      pa[i1] = (g1 << 16) | i1;
    }
  }
  qsync<PACKED>();
  for (uint32_t i = tid; i < n; i += T) {
    const uint32_t key = pa[i] >> 16;
    const bool lost = unmap[key] != i;  // somebody else holds my slot: the key is shared
    if (lost) dupk[key] = 1;
    tie |= lost;
  }
  if (tid < n4 - n) pa[n + tid] = 0xFFFFFFFFu;
  const bool anytie = W == 1 ? (bool)__any(tie) : (bool)__syncthreads_or(tie);
  qsync<PACKED>();
  QR_T(2);
  // ---- 2. with ties the permutation is what GNU std::sort leaves: the partition phase by
  //         one wave, the stable placement that ends it by all
  if (anytie) {
    const int limit = exact_tail ? (int)n : (int)(cutoff < n ? cutoff : n);
    if (W == 1) {
      wave_gnu_sort<!PACKED>(pa, (int)n, LB, RB, stk, dupk, limit);
    } else {
      if (wave == 0) wave_gnu_sort<false>(pa, (int)n, LB, RB, stk, dupk, limit);
      qsync<PACKED>();
    }
    QR_TS(8);
    sort_placement<W, PACKED>(pa, (int)n, unmap, dupk, LB, RB, tid);
    qsync<PACKED>();
  }
  QR_T(3);
  for (uint32_t r = tid; r < n; r += T) {
    const uint32_t d = unmap[r];
    sl[r] = lab0[d];
    sr[r] = s[d];
    if (ranks_out) ranks_out[off + r] = present ? cmap[d] : d;
  }
  qsync<PACKED>();
  const uint32_t size = cutoff < n ? cutoff : n;
  double my_idcg = metric == QR_METRIC_NDCG ? idcg[q] : 1.0;
  if (present && metric == QR_METRIC_NDCG) {
    // Ndcg::compute_idcg (ndcg.cc:35-47) of the cleaned list: labels in descending
    // order of their integer part (equal integer parts keep the list order -- for
    // the usual integer grades any order gives the same value), then dcg.cc:33-39.
    // ownl is free here (the sort is over, the accumulators are not live yet).
    double *sorted_gain = ownl;
    for (uint32_t j = lane; j < n; j += 64) {
      const int lj = (int)lab0[j];
      uint32_t r = 0;
      for (uint32_t i = 0; i < n; ++i) {
        const int li = (int)lab0[i];
        r += (li > lj) || (li == lj && i < j);
      }
      sorted_gain[r] = pow2_label(lab0[j]) - 1.0;
    }
    qsync<PACKED>();
    double v = 0.0;
    if (lane == 0)
      for (uint32_t i = 0; i < size; ++i) v += sorted_gain[i] / lg2[i];
    my_idcg = readlane_f64(v, 0);
    qsync<PACKED>();
  }
  // ---- 3. metric of the current ranking (dcg.cc:33-39, ndcg.cc:49-58)
  if (W == 1 || wave == 0) {
    // the terms in parallel (one IEEE division per lane), the sum in rank order -- the
    // summation order of dcg.cc:36-38, so the value is bit for bit the sequential one
    double dcg = 0.0;
    for (uint32_t base = 0; base < size; base += 64) {
      const uint32_t i = base + lane;
      const double term = i < size ? (pow2_label(sl[i]) - 1.0) / lg2[i] : 0.0;
      const uint32_t m = size - base < 64 ? size - base : 64;
      for (uint32_t k = 0; k < m; ++k) dcg += readlane_f64(term, (int)k);
    }
    if (lane == 0) {
      double m = dcg;
      if (metric == QR_METRIC_NDCG) m = my_idcg > 0 ? dcg / my_idcg : 0.0;
      qmetric[q] = m;
    }
  }
  QR_T(4);
  if (mode == 1) return;
  // ---- 4. lambdas
  if (metric == QR_METRIC_NDCG && !(my_idcg > 0.0)) {
    // ndcg.cc:69-70: all-zero jacobian => lambdas and weights stay 0
    for (uint32_t i = tid; i < n_full; i += T) {
      lambda[off + i] = 0.0;
      weight[off + i] = 0.0;
    }
    if (tid == 0 && ssq) ssq[2 * q] = ssq[2 * q + 1] = qmax[q] = 0.0;
    return;
  }
  for (uint32_t i = tid; i < n; i += T) {
    ownl[i] = 0.0;
    ownw[i] = 0.0;
    s[i] = pow2_label(sl[i]);  // 2^label by rank (scores by doc are no longer needed)
  }
  for (uint32_t i = tid; i < size; i += T) ilt[i] = ilg2[i];
  if (tid < 64) expt[tid] = QR_EXP_T[tid];
  qsync<PACKED>();
  const double *pw = s;
  const uint32_t nbatch = (n + 63) / 64;
  const double inv_idcg = metric == QR_METRIC_NDCG ? 1.0 / my_idcg : 1.0;  // (ndcg.cc:81: / idcg, to an ulp)
  // one pair term (lambdamart.cc:120-141 with the closed form of ndcg.cc:76-88): rank r1
  // (uniform over the wave) against this lane's rank r2 > r1
  // `flip(v, f)`: f ? -v : v, as one integer op on the sign bit
  auto flip = [](const double v, const bool f) {
    return __hiloint2double(__double2hiint(v) ^ (f ? (int)0x80000000 : 0), __double2loint(v));
  };
  // one pair term (lambdamart.cc:120-141 with the closed form of ndcg.cc:76-88): rank r1
  // (uniform over the wave) against this lane's rank r2 > r1.  slam = the pair's lambda with
  // the sign it enters rank r1's sum with (the higher label plays "j", lambdamart.cc:127).
  auto pair_term = [&](const float l1, const double p1, const double inv1, const double s1,
                       const float l2, const double p2, const double il2, const double s2, double &slam,
                       double &del) {
    const double j = (il2 - inv1) * (p1 - p2);   // il2 == 0 beyond the cutoff (ndcg.cc:84-86)
    const double d = fabs(j * inv_idcg);
    const bool lo1 = l1 < l2;
    const double diff = flip(s1 - s2, lo1);      // s_hi - s_lo
    const double rho = qr_rcp(1.0 + qr_exp(diff, expt));
    slam = flip(rho * d, lo1);
    del = rho * (1.0 - rho) * d;
  };
  // The same term from per-document exponentials e = exp(score - max score of the query):
  // 1 / (1 + exp(s_hi - s_lo)) = e_lo / (e_lo + e_hi), `lo` the document with the lower
  // label -- n exponentials per query instead of one per pair (to a few ulp the same value;
  // the lambdas are held to 1e-11).  Used when the query's scores span less than 690, so
  // that no e underflows.
  auto pair_term_e = [&](const float l1, const double p1, const double inv1, const double e1,
                         const float l2, const double p2, const double il2, const double e2, double &slam,
                         double &del) {
    const double j = (il2 - inv1) * (p1 - p2);
    const double d = fabs(j * inv_idcg);
    const bool lo1 = l1 < l2;
    const double r = qr_rcp(e1 + e2);
    const double rho = (lo1 ? e1 : e2) * r;
    slam = flip(rho * d, lo1);
    del = rho * (1.0 - rho) * d;
  };
  if (SMALL || n <= 128) {
    if (W == 1 || wave == 0) {
    // Every lane keeps its two ranks (lane, lane + 64) -- label, 2^label, discount,
    // score and the two accumulators -- in registers for the whole sweep; the only
    // memory traffic of a step is the broadcast read of rank r1's four values, and the
    // two ranks' terms are independent instruction chains.  Same additions in the same
    // order as the general loop below.
    const uint32_t ra = lane, rb = lane + 64;
    const bool ina = ra < n, inb = rb < n;
    const float la = ina ? sl[ra] : 0.f, lb = inb ? sl[rb] : 0.f;
    const double pa_ = ina ? pw[ra] : 0.0, pb_ = inb ? pw[rb] : 0.0;
    const double ia = ra < size ? ilt[ra] : 0.0, ib = rb < size ? ilt[rb] : 0.0;
    // (ranks are in non-increasing score order: the query's scores span sr[0] - sr[n - 1])
    const bool use_e = sr[0] - sr[n - 1] <= 690.0;
    double sa = ina ? sr[ra] : 0.0, sb = inb ? sr[rb] : 0.0;
    if (use_e) {
      const double smax = sr[0];
      sa = qr_exp(sa - smax, expt);
      sb = qr_exp(sb - smax, expt);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (every lane has read sr[0], sr[n - 1])
      __builtin_amdgcn_wave_barrier();
      if (ina && ra < size) sr[ra] = sa;  // what the broadcast of rank r1 reads below
      if (inb && rb < size) sr[rb] = sb;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    double ola = 0.0, owa = 0.0, olb = 0.0, owb = 0.0;
    // Five ranks r1 per round: their ten per-lane sums (c1, cw) are reduced TOGETHER -- a
    // reduce-scatter over the wave's halves (v_permlane32_swap: the lower half carries on with
    // the c1's, the upper with the cw's), one over the row pairs (v_permlane16_swap: even rows
    // the even ranks of the round, odd rows the odd ones), then three row totals on DPP -- 60
    // vector instructions per round where ten wave_sum calls took ~260 (a query's twenty wave
    // sums were a third of the pair sweep).  Fixed association, as before; the
    // totals land in lanes 0 / 16 / 32 / 48.
    constexpr int RR = 5;
    for (uint32_t rbase = 0; rbase < size; rbase += RR) {
      double c1v[RR], cwv[RR];
#pragma unroll
      for (int u = 0; u < RR; ++u) {
        const uint32_t r1 = rbase + u;
        double c1 = 0.0, cw = 0.0;
        if (r1 < size) {  // (wave-uniform)
          const float l1 = sl[r1];
          const double p1 = pw[r1], inv1 = ilt[r1], s1 = sr[r1];
          const bool va = ina && ra > r1 && l1 != la, vb = inb && rb > r1 && l1 != lb;
          if (va) {
            double slam, del;
            if (use_e)
              pair_term_e(l1, p1, inv1, s1, la, pa_, ia, sa, slam, del);
            else
              pair_term(l1, p1, inv1, s1, la, pa_, ia, sa, slam, del);
            c1 += slam;
            cw += del;
            ola -= slam;
            owa += del;
          }
          if (vb) {
            double slam, del;
            if (use_e)
              pair_term_e(l1, p1, inv1, s1, lb, pb_, ib, sb, slam, del);
            else
              pair_term(l1, p1, inv1, s1, lb, pb_, ib, sb, slam, del);
            c1 += slam;
            cw += del;
            olb -= slam;
            owb += del;
          }
        }
        c1v[u] = c1;
        cwv[u] = cw;
      }
      // halves: lanes 0..31 x[u] = c1v[u][l] + c1v[u][l + 32], lanes 32..63 the same of cwv[u]
      double x[RR];
#pragma unroll
      for (int u = 0; u < RR; ++u) x[u] = swap32_add(c1v[u], cwv[u]);
      // row pairs: even rows (x[0], x[2], x[4]), odd rows (x[1], x[3], nothing)
      double z0 = swap16_add(x[0], x[1]), z1 = swap16_add(x[2], x[3]), z2 = swap16_add(x[4], 0.0);
      z0 = row_sum(z0);
      z1 = row_sum(z1);
      z2 = row_sum(z2);
      if ((lane & 15u) == 0) {
        double *acc = lane < 32 ? accl : accw;   // rows 0, 1: the lambdas' sums; rows 2, 3: the weights'
        const uint32_t odd = (lane >> 4) & 1u;   // rows 1, 3: the odd ranks of the round
        if (rbase + odd < size) acc[rbase + odd] = z0;
        if (rbase + 2 + odd < size) acc[rbase + 2 + odd] = z1;
        if (!odd && rbase + 4 < size) acc[rbase + 4] = z2;
      }
    }
    if (ina) {
      ownl[ra] = ola;
      ownw[ra] = owa;
    }
    if (inb) {
      ownl[rb] = olb;
      ownw[rb] = owb;
    }
    }
  } else {
  // (longer queries: the same per-document exponentials, in place of the scores by rank)
  const bool use_e = sr[0] - sr[n - 1] <= 690.0;
  {
    const double smax = sr[0];
    qsync<PACKED>();  // (every thread has read sr[0], sr[n - 1])
    if (use_e)
      for (uint32_t r = tid; r < n; r += T) sr[r] = qr_exp(sr[r] - smax, expt);
    qsync<PACKED>();
  }
  // Rounds of five ranks r1, as above: a lane's batch position r2 meets the round's five ranks
  // in turn (its own accumulators stay in registers for the round: the same subtractions in
  // the same order as one rank at a time), the round's ten per-lane sums are reduced together,
  // and with several waves ONE barrier ends the round (their sums meet in `sh_part`, added
  // pairwise in wave order by ten threads) -- where one rank at a time took a barrier, two
  // wave sums and a single thread's additions per rank.
  constexpr int RR = QR_LAMBDA_RR;
  for (uint32_t rbase = 0; rbase < size; rbase += RR) {
    double c1v[RR], cwv[RR];
#pragma unroll
    for (int u = 0; u < RR; ++u) c1v[u] = cwv[u] = 0.0;
    for (uint32_t bt = rbase / 64 + (W > 1 ? wave : 0u); bt < nbatch; bt += W) {  // a wave's batches
      const uint32_t r2 = bt * 64 + lane;
      if (r2 < n && r2 > rbase) {
        const float l2 = sl[r2];
        const double p2 = pw[r2], il2 = r2 < size ? ilt[r2] : 0.0, s2 = sr[r2];
        double ol = ownl[r2], ow = ownw[r2];  // only this lane touches rank r2 in this round
#pragma unroll
        for (int u = 0; u < RR; ++u) {
          const uint32_t r1 = rbase + u;
          if (r1 >= size) break;  // (uniform)
          // rank r1's four values: broadcast LDS reads, not kept in registers over the round
          // (thirty-five registers that cost every launch of this kernel a wave per SIMD)
          const float l1 = sl[r1];
          if (r2 > r1 && l1 != l2) {
            double slam, del;
            if (use_e)
              pair_term_e(l1, pw[r1], ilt[r1], sr[r1], l2, p2, il2, s2, slam, del);
            else
              pair_term(l1, pw[r1], ilt[r1], sr[r1], l2, p2, il2, s2, slam, del);
            c1v[u] += slam;
            cwv[u] += del;
            ol -= slam;
            ow += del;
          }
        }
        ownl[r2] = ol;
        ownw[r2] = ow;
      }
    }
    double x[RR];
#pragma unroll
    for (int u = 0; u < RR; ++u) x[u] = swap32_add(c1v[u], cwv[u]);
    double z0 = swap16_add(x[0], x[1]), z1 = swap16_add(x[2], x[3]), z2 = swap16_add(x[4], 0.0);
    z0 = row_sum(z0);
    z1 = row_sum(z1);
    z2 = row_sum(z2);
    const uint32_t odd = (lane >> 4) & 1u;  // rows 1, 3: the odd ranks of the round
    if (W == 1) {
      if ((lane & 15u) == 0) {
        double *acc = lane < 32 ? accl : accw;  // rows 0, 1: the lambdas' sums; rows 2, 3: the weights'
        if (rbase + odd < size) acc[rbase + odd] = z0;
        if (rbase + 2 + odd < size) acc[rbase + 2 + odd] = z1;
        if (!odd && rbase + 4 < size) acc[rbase + 4] = z2;
      }
    } else {  // the waves' sums, added in wave order (two buffers: one barrier per round)
      const uint32_t par = (rbase / RR) & 1u;
      if ((lane & 15u) == 0) {
        double *dst = &sh_part[par][wave][lane < 32 ? 0 : RR];
        dst[odd] = z0;
        dst[2 + odd] = z1;
        if (!odd) dst[4] = z2;
      }
      qsync<PACKED>();
      if (tid < 2 * RR) {
        const uint32_t u = tid % RR, r1 = rbase + u;
        if (r1 < size) {
          // pairwise over the waves: ((w0 + w1) + (w2 + w3)) + ...
          double v[W];
#pragma unroll
          for (int w = 0; w < W; ++w) v[w] = sh_part[par][w][tid];
#pragma unroll
          for (int st = 1; st < W; st *= 2)
#pragma unroll
            for (int w = 0; w < W; w += 2 * st) v[w] += v[w + st];
          (tid < RR ? accl : accw)[r1] = v[0];
        }
      }
    }
  }
  }
  qsync<PACKED>();
  QR_T(5);
  double mx = 0.0, sq = 0.0, sm = 0.0;
  for (uint32_t r = tid; r < n; r += T) {
    const uint32_t d = off + (present ? cmap[unmap[r]] : unmap[r]);
    double l = ownl[r], w = ownw[r];
    if (r < size) {
      l += accl[r];
      w += accw[r];
    }
    lambda[d] = l;
    weight[d] = w;
    const double a = fabs(l);
    mx = a > mx ? a : mx;
    sq += l * l;
    sm += l;
  }
  mx = wave_max(mx);
  sq = wave_sum(sq);
  sm = wave_sum(sm);
  if (W > 1) {  // over the waves, in wave order
    if (lane == 0) {
      sh_red[wave][0] = mx;
      sh_red[wave][1] = sq;
      sh_red[wave][2] = sm;
    }
    qsync<PACKED>();
    double vm[W], vq[W], vs[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      vm[w] = sh_red[w][0];
      vq[w] = sh_red[w][1];
      vs[w] = sh_red[w][2];
    }
#pragma unroll
    for (int st = 1; st < W; st *= 2)
#pragma unroll
      for (int w = 0; w < W; w += 2 * st) {
        vm[w] = fmax(vm[w], vm[w + st]);
        vq[w] += vq[w + st];
        vs[w] += vs[w + st];
      }
    mx = vm[0];
    sq = vq[0];
    sm = vs[0];
  }
  // (the query's max |lambda| goes to its own slot and k_prep takes the maximum: ten
  // thousand atomicMax on one address queued up for half of this launch's duration)
  if (tid == 0 && ssq) {
    ssq[2 * q] = sq;
    ssq[2 * q + 1] = sm;
    qmax[q] = mx;
    // ... and into the iteration's slot set (qr_prep.h), for the launches that need the scale
    // before the scalars are finished: one atomic that returns nothing, on one of 64 words
    if (qslot && mx > 0.0)
      (void)__hip_atomic_fetch_max(&qslot[q % QR_PREP_SLOTS], (unsigned long long)__double_as_longlong(mx),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef QR_LAMBDA_TIMING
  QR_T(6);
  if (tid == 0 && (q % 211 == 0 || n > 540))
    printf("k_lambda q=%u n=%u tie=%d: load %lld count %lld sort %lld rank/metric %lld pairs %lld out %lld total %lld\n",
           q, n, (int)anytie, tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4],
           tq[6] - tq[5], tq[6] - tq[0]);
#endif
}

// (SMALL allocates 73 VGPRs since the score update's operands ride along: six waves per SIMD.
// Held at 72 = seven waves by amdgpu_waves_per_eu it spills three dwords and is no faster:
// 0.4110 / 0.4109 ms per iteration against 0.4086 / 0.4109 without the attribute.)
template <bool LONG, int W, bool SMALL = false>
__global__ __launch_bounds__(64 * W) void k_lambda(const QrLambdaArgs A, const uint32_t nmax, const uint32_t kacc,
                                                   const uint32_t *__restrict__ long_list,
                                                   char *__restrict__ lscratch, const size_t lstride) {
  extern __shared__ __attribute__((aligned(16))) char lds_mem[];
  // (LONG = false: `long_list`, when given, is the launch's size class -- the queries whose
  // working set fits THIS launch's LDS)
  const uint32_t q = long_list ? long_list[blockIdx.x] : blockIdx.x;
  char *smem = LONG ? lscratch + (size_t)blockIdx.x * lstride : lds_mem;
  __shared__ double sh_part[2][W][2 * QR_LAMBDA_RR], sh_red[W][3];
  lambda_query<W, false, SMALL>(A, q, smem, nmax, kacc, threadIdx.x, sh_part, sh_red, (double *)nullptr);
}

// ONE launch for a ragged query set (config 1's shape): workgroups of eight waves in three
// roles, dispatched longest queries first --
//   [0, nC)   a query of more than 512 documents, the eight waves together;
//   then      three queries of 257 .. 512 documents, one wave each (the other waves leave: a
//             query's sort emulation is one wave's work anyway, and a wave of its own costs the
//             chip an eighth of the slots eight waves waiting for it do);
//   then      six queries of 129 .. 256 documents, one wave each;
//   the rest  eight queries of up to 128 documents, one wave each
// -- in place of one launch per size class side by side on auxiliary streams (fork, three or
// four launches, join: the join alone cost ~18 us, and each class waited for its own longest
// query).  `list`: the queries in dispatch order.  LDS: every role carves the same dynamic
// block (two workgroups of 76 KB per CU).
// (Measured and left out: amdgpu_waves_per_eu(6) -- 80 VGPRs with 60 bytes of scratch per lane --
// and 52 KB of LDS per workgroup, i.e. three workgroups per CU: 0.483 ms per iteration on the
// MSLR-shaped set against 0.456 with 95 VGPRs and two workgroups of 76 KB.)
#define QR_LU_DYN_DEFAULT 77824
__global__ __launch_bounds__(64 * QR_LU_W) void k_lambda_u(const QrLambdaArgs A, const QrLambdaPlanDev P,
                                                           const uint32_t *__restrict__ list) {
  extern __shared__ __attribute__((aligned(16))) char lds_mem[];
  __shared__ double sh_part[2][QR_LU_W][2 * QR_LAMBDA_RR], sh_red[QR_LU_W][3], sh_expt[64];
  const uint32_t b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  if (b < P.nC) {
    // the launch's critical path: the longest queries, first in and ahead of their neighbours
    __builtin_amdgcn_s_setprio(3);
    lambda_query<QR_LU_W, false>(A, list[b], lds_mem, P.nmaxC, P.kaccC, tid, sh_part, sh_red, sh_expt);
    return;
  }
  // packed roles: one wave per query, `per` queries per workgroup (the other waves leave)
  uint32_t bb = b - P.nC, r = 0;
  while (r + 1 < QR_LU_ROLES && bb >= P.pk[r].blocks) bb -= P.pk[r++].blocks;
  if (wave >= P.pk[r].per) return;
  uint32_t idx = bb * P.pk[r].per + wave;
  if (idx >= P.pk[r].count) return;
  idx += P.pk[r].first;
  const uint32_t nmax = P.pk[r].nmax, kacc = P.pk[r].kacc;
  char *smem = lds_mem + (size_t)wave * P.pk[r].slice;
  // (every wave fills the workgroup's exponential table itself, with the same values, before
  // its own fence: no wave waits for another)
  lambda_query<1, true>(A, list[idx], smem, nmax, kacc, tid & 63u, (double(*)[1][2 * QR_LAMBDA_RR]) nullptr,
                        (double(*)[3]) nullptr, sh_expt);
}

// Mart::compute_pseudoresponses (mart.cc:418-431): label - score
__global__ __launch_bounds__(256) void k_residual(const float *__restrict__ labels,
                                                  const double *__restrict__ scores,
                                                  double *__restrict__ out, uint32_t N,
                                                  double *__restrict__ ssq,
                                                  double *__restrict__ qmax) {
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
    qmax[blockIdx.x] = m;
  }
}

// the scalars into the pinned host block, every field EXCEPT `pad`: that word is the
// launch's sequence number and only ever moves forward (stored last, by the caller, behind a
// system-scope fence); a whole-struct copy would first put the device's 0 over it
// The per-iteration scalars in a launch of their own (qr_prep.h: prep_body; the batched and
// level-wise growth paths let the same sixteen workgroups ride in their root scan launch
// instead, k_tree.hip).
__global__ __launch_bounds__(64) void k_prep(const QrPrepJob j) { prep_body<16>(j, blockIdx.x, gridDim.x); }

// ---------------------------------------------------------------------------
static size_t lambda_lds(size_t nmax, size_t kacc, bool sampled, bool own_expt = true) {
  // s/sr[nmax] f64, accl/accw[kacc] f64, ownl/ownw[nmax] f64 (aliased by the sort
  // scratch), lab0/sl f32, unmap u32, stk, ilt[kacc] f64, expt, dupk, (cmap u32)
  return nmax * 16 + kacc * 16 + nmax * 16 + nmax * 12 + 3 * 40 * 4 + kacc * 8 + (own_expt ? 64 * 8 : 0) +
         ((nmax + 7) & ~(size_t)7) + (sampled ? nmax * 4 : 0);
}

// size classes of the LDS-resident launches: a query goes to the first class that holds it
// (measured on the MSLR-shaped set: {128, 512, 2048} with four waves from 129 documents on is
// 10 us slower per iteration, a 192 or 64 bound more, sixteen waves from 257 on 40 us)
static const uint32_t kClassBound[] = {128, 256, 512, 2048, 0xFFFFFFFFu};

#ifndef QR_LAMBDA_W4_FROM
#define QR_LAMBDA_W4_FROM 256  /* classes whose longest query is longer than this take four waves per query */
#endif
#ifndef QR_LAMBDA_AUX
#define QR_LAMBDA_AUX 2  /* auxiliary streams of a ragged set's size-class launches (at most 4) */
#endif
int qr_k_lambda(qr_ctx *c, int which, int metric, size_t cutoff, int mode) {
  const size_t Q = which ? c->vQ : c->Q;
  const size_t maxq = which ? c->vmaxq : c->maxq;
  if (Q == 0) return QR_OK;
  if (maxq > 65535)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "queries of more than 65535 documents are not supported");
  const uint32_t cut = cutoff == 0 ? NO_CUTOFF : (uint32_t)cutoff;
  size_t kacc = cutoff == 0 || cutoff > maxq ? maxq : cutoff;
  if (kacc == 0) kacc = 1;
  kacc = (kacc + 1) & ~(size_t)1;
  // (the static part: sh_part [2][W][10] + sh_red [W][3] doubles = 2.9 KB with sixteen waves)
  const size_t limit = 160 * 1024 - 4096;
  const bool sampled = which == 0 && c->sub_k != 0;
  // the longest query the LDS holds; longer ones run out of a global scratch slice each
  // (LONG launch); an unbounded cutoff on a long query would also need the per-rank
  // accumulators there, so it takes the same route
  size_t nmax_lds = (maxq + 3) & ~(size_t)3;
  while (nmax_lds > 4 && lambda_lds(nmax_lds, std::min(kacc, nmax_lds), sampled) > limit) nmax_lds -= 4;
  const std::vector<uint64_t> &qo = which ? c->h_vqoff : c->h_qoff;
  int &tag = which ? c->long_tag[which] : c->long_tag[0];
  std::vector<uint32_t> &llist = which ? c->h_long_list[1] : c->h_long_list[0];
  std::vector<qr_ctx::QClass> &classes = c->h_qclass[which];
  // The single launch of a ragged set (k_lambda_u): its roles carve one dynamic LDS block per
  // workgroup -- eight packed working sets of up to 128 documents, six of up to 256, or one
  // query of up to `capC` documents -- sized so that two workgroups share a CU.  Queries
  // beyond capC keep their own launches (size class / global scratch) beside it.
  auto a16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  // a packed role's capacity (a role whose bound repeats the next one's stays empty: by default
  // no query of more than 256 documents is one wave's work -- measured on the MSLR-shaped set:
  // {512, 256, 128} 0.471 ms per iteration, {256, 256, 128} 0.456)
  static const uint32_t kPackBound[QR_LU_ROLES] = {256, 256, 128};
  size_t pk_kacc[QR_LU_ROLES], pk_slice[QR_LU_ROLES], pk_per[QR_LU_ROLES];
  // 160 KB / 2 (the kernel's VGPRs allow two workgroups of eight waves per CU) less its static
  // 2 KB (sh_part, sh_red, sh_expt); a packed role takes as many queries per workgroup as fit
  // (three, six and eight at cutoff 10)
  size_t lu_dyn = (size_t)QR_LU_DYN_DEFAULT;
  for (int r = 0; r < QR_LU_ROLES; ++r) {
    pk_kacc[r] = std::min<size_t>(kacc, kPackBound[r]);
    pk_slice[r] = a16(lambda_lds(kPackBound[r], pk_kacc[r], false, false));
    lu_dyn = std::max(lu_dyn, (r == QR_LU_ROLES - 1 ? (size_t)QR_LU_W : (size_t)2) * pk_slice[r]);
  }
  for (int r = 0; r < QR_LU_ROLES; ++r) pk_per[r] = std::min<size_t>(QR_LU_W, lu_dyn / pk_slice[r]);
  size_t capC = std::min<size_t>(nmax_lds, 4096);
  while (capC > kPackBound[0] && lambda_lds(capC, std::min(kacc, capC), false, false) > lu_dyn) capC -= 4;
  bool lu_try = !sampled && lu_dyn <= limit && capC > kPackBound[0];
  const int want_tag = (int)nmax_lds + ((int)capC << 16) + (lu_try ? 1 << 30 : 0);
  // (the plan's slices hold `kacc` top ranks: another cutoff is another plan, whatever capC says)
  if (tag != want_tag || c->lu_kacc[which] != kacc) {  // (re)build the classes and the long list for this capacity
    c->lu_kacc[which] = kacc;
    std::vector<std::vector<uint32_t>> by(sizeof(kClassBound) / sizeof(kClassBound[0]));
    std::vector<uint32_t> cmax(by.size(), 0);
    std::vector<uint32_t> role[1 + QR_LU_ROLES];  // beyond the packed roles, up to capC documents: eight waves together; then the packed roles
    for (int pass = 0; pass < 2; ++pass) {
      llist.clear();
      for (auto &v : by) v.clear();
      for (auto &v : role) v.clear();
      std::fill(cmax.begin(), cmax.end(), 0u);
      for (size_t q = 0; q < Q; ++q) {
        const size_t n = qo[q + 1] - qo[q];
        if (n > nmax_lds) {
          llist.push_back((uint32_t)q);
          continue;
        }
        if (lu_try && n <= capC) {
          int r = 0;
          while (r < QR_LU_ROLES && n <= kPackBound[r]) ++r;  // the smallest role that holds it
          role[r].push_back((uint32_t)q);
          continue;
        }
        size_t k = 0;
        while (n > kClassBound[k]) ++k;
        by[k].push_back((uint32_t)q);
        cmax[k] = std::max(cmax[k], (uint32_t)n);
      }
      // a set that one launch of one kind serves anyway (every query in the same role and
      // nothing beside it -- the uniform bench set) keeps that launch: no list, no roles
      int kinds = 0;
      for (auto &v : role) kinds += v.empty() ? 0 : 1;
      size_t others = llist.empty() ? 0 : 1;
      for (auto &v : by) others += v.empty() ? 0 : 1;
      if (!lu_try || kinds + others >= 2) break;
      lu_try = false;
    }
    c->lu_on[which] = lu_try;
    std::vector<uint32_t> ulist;
    if (lu_try) {
      QrLambdaPlanDev &P = c->lu_plan[which];
      for (auto &v : role)  // longest first (equal lengths in query order)
        std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return qo[a + 1] - qo[a] > qo[b + 1] - qo[b]; });
      uint32_t nmaxC = role[0].empty() ? 0u : (uint32_t)(qo[role[0][0] + 1] - qo[role[0][0]]);
      nmaxC = std::max<uint32_t>((nmaxC + 3) & ~3u, kPackBound[0] + 4);
      P.nC = (uint32_t)role[0].size();
      P.nmaxC = nmaxC;
      P.kaccC = (uint32_t)std::min<size_t>(kacc, nmaxC);
      ulist = role[0];
      P.blocks = P.nC;
      for (int r = 0; r < QR_LU_ROLES; ++r) {
        const std::vector<uint32_t> &v = role[1 + r];  // role[1 + r]: the queries packed role r holds (largest capacity first)
        P.pk[r].first = (uint32_t)ulist.size();
        P.pk[r].count = (uint32_t)v.size();
        P.pk[r].per = (uint32_t)pk_per[r];
        P.pk[r].blocks = (uint32_t)((v.size() + pk_per[r] - 1) / pk_per[r]);
        P.pk[r].nmax = kPackBound[r];
        P.pk[r].kacc = (uint32_t)pk_kacc[r];
        P.pk[r].slice = (uint32_t)pk_slice[r];
        P.blocks += P.pk[r].blocks;
        ulist.insert(ulist.end(), v.begin(), v.end());
      }
      c->lu_dyn[which] = std::max(lu_dyn, lambda_lds(nmaxC, P.kaccC, false, false));
    }
    // (Measured and left out: the longest queries first inside a class -- the 16-wave launch of the
    // MSLR-shaped set stays at 74 us: its time is its longest query's, not a dispatch-order tail.)
    std::vector<uint32_t> flat;
    classes.clear();
    for (size_t k = 0; k < by.size(); ++k)
      if (!by[k].empty()) {
        classes.push_back({(uint32_t)flat.size(), (uint32_t)by[k].size(), cmax[k]});
        flat.insert(flat.end(), by[k].begin(), by[k].end());
      }
    // small classes join their larger neighbour: a launch is worth ~5 us
    for (size_t k = 0; k + 1 < classes.size();)
      if (classes[k].count < 64) {
        classes[k + 1].first = classes[k].first;
        classes[k + 1].count += classes[k].count;
        classes.erase(classes.begin() + k);
      } else
        ++k;
    c->qclass_identity[which] = classes.size() == 1 && classes[0].count == Q && !lu_try;
    QR_CHECK(c, hipStreamSynchronize(c->stream));
    uint32_t *&dq = c->d_qclass[which];
    uint32_t *&dl = c->d_long_list[which];
    uint32_t *&du = c->d_lu_list[which];
    if (dq) (void)hipFree(dq);
    if (dl) (void)hipFree(dl);
    if (du) (void)hipFree(du);
    dq = nullptr;
    dl = nullptr;
    du = nullptr;
    QR_CHECK(c, hipMalloc((void **)&dq, (flat.size() + 1) * 4));
    QR_CHECK(c, hipMalloc((void **)&dl, (llist.size() + 1) * 4));
    QR_CHECK(c, hipMalloc((void **)&du, (ulist.size() + 1) * 4));
    QR_CHECK(c, hipMemcpy(dq, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    QR_CHECK(c, hipMemcpy(dl, llist.data(), llist.size() * 4, hipMemcpyHostToDevice));
    QR_CHECK(c, hipMemcpy(du, ulist.data(), ulist.size() * 4, hipMemcpyHostToDevice));
    tag = want_tag;
  }
  const size_t nlong = llist.size();
  size_t lstride = 0, nmax_long = 0;
  if (nlong) {
    nmax_long = (maxq + 3) & ~(size_t)3;
    lstride = (lambda_lds(nmax_long, std::min(kacc, nmax_long), sampled) + 255) & ~(size_t)255;
    if (nlong * lstride > c->lscratch_bytes) {
      QR_CHECK(c, hipStreamSynchronize(c->stream));
      if (c->d_lscratch) (void)hipFree(c->d_lscratch);
      c->d_lscratch = nullptr;
      QR_CHECK(c, hipMalloc((void **)&c->d_lscratch, nlong * lstride));
      c->lscratch_bytes = nlong * lstride;
    }
  }
  double *sc = which ? c->d_vscores : c->d_scores;
  const float *lb = which ? c->d_vlabels : c->d_labels;
  const uint32_t *qoffd = which ? c->d_vqoff : c->d_qoff;
  const double *idcg = which ? c->d_vidcg : c->d_idcg;
  double *lam = which ? nullptr : c->d_lambda, *wgt = which ? nullptr : c->d_weight;
  double *qm = which ? c->d_vqmetric : c->d_qmetric;
  uint32_t *ranks = which ? nullptr : c->d_ranks;
  double *ssq = (!which && mode == 0) ? c->d_ssq : nullptr;
  if (ssq) c->nqmax = c->Q;
  // the iteration's slot set (qr_prep.h): the sets alternate from one lambda pass to the next
  // (document-sharded contexts exchange the maximum between the ranks first: no slots there)
  unsigned long long *qslot = nullptr;
  if (ssq && !c->dmode) {
    c->prep_parity ^= 1;
    qslot = qr_prep_slots(c, c->prep_parity);
  }
  const int md = which ? 1 : mode;
  const uint8_t *present = (!which && mode == 0 && c->sub_k) ? c->d_present : nullptr;
  // One launch per size class (its LDS sized for the class's longest query), the launches
  // side by side on auxiliary streams when there are several: the few long queries of a
  // ragged set no longer dictate the occupancy of the many short ones.
  // a score update left pending by qr_scores_update rides in this pass (training set, lambdas)
  const bool upd = !which && mode == 0 && c->lazy_scores;
  if (c->lazy_scores && !which && !upd) {
    const int frc = qr_k_scores_flush(c);
    if (frc) return frc;
  }
  c->lazy_scores = upd ? false : c->lazy_scores;
  const QrLambdaArgs A = {sc,  lb, qoffd, metric, cut,       idcg,  c->d_lg2, c->d_ilg2, lam,
                          wgt, qm, ranks, ssq,    c->d_qmax, qslot, md,       present,   c->exact_tail,
                          upd ? c->d_leafb : (const uint8_t *)nullptr,
                          upd ? c->d_tree->leaf_value : (const double *)nullptr, upd ? c->lazy_shrinkage : 0.0};
  const bool lu_on = c->lu_on[which];
  const size_t nlaunch = classes.size() + (nlong ? 1 : 0) + (lu_on ? 1 : 0);
  const bool fork = nlaunch > 1;
  if (fork) {
    if (!c->aux_fork) QR_CHECK(c, hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
    QR_CHECK(c, hipEventRecord(c->aux_fork, c->stream));
  }
  size_t li = 0;  // launch index: 0 stays on the context's stream
  auto stream_for = [&](size_t i, hipStream_t *out) -> int {
    if (i == 0) {
      *out = c->stream;
      return QR_OK;
    }
    // TWO auxiliary streams: with the context's own that is three hardware queues, and HIP puts a
    // third auxiliary stream on the queue of the second anyway (rocprofv3 trace of the MSLR-shaped
    // set: the fourth launch waited for the third, 70 + 27 us in a row while the 51 us launch's
    // queue sat empty).  The launches go out longest class first, so the fourth -- the short
    // queries -- lines up behind the second: 51 + 27 us next to 70 and 70.
    const size_t a = (i - 1) % QR_LAMBDA_AUX;
    if (!c->aux_stream[a]) {
      // one auxiliary stream above, one below the context's priority: streams of different
      // priorities never share a hardware queue, whatever else the process has created (with
      // plain streams the MSLR-shaped set ran at 0.53 ms per iteration in a process of its own
      // and at 0.57 inside bench.py, whose earlier contexts had shifted the round-robin of
      // streams over hardware queues; with priorities 0.53 in both)
      int lo = 0, hi = 0;
      QR_CHECK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
      QR_CHECK(c, hipStreamCreateWithPriority(&c->aux_stream[a], hipStreamNonBlocking,
                                              a == 0 ? hi : (a == 1 ? lo : 0)));
      QR_CHECK(c, hipEventCreateWithFlags(&c->aux_join[a], hipEventDisableTiming));
    }
    if (i <= QR_LAMBDA_AUX) QR_CHECK(c, hipStreamWaitEvent(c->aux_stream[a], c->aux_fork, 0));
    *out = c->aux_stream[a];
    return QR_OK;
  };
  if (lu_on) {  // the ragged set's single launch, on the context's own stream
    const QrLambdaPlanDev &P = c->lu_plan[which];
    const size_t lds = c->lu_dyn[which];
    if (lds > c->attr_lambda_u_lds) {
      QR_CHECK(c, hipFuncSetAttribute((const void *)k_lambda_u, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      c->attr_lambda_u_lds = lds;
    }
    hipStream_t st;
    int rc = stream_for(li++, &st);
    if (rc) return rc;
    const unsigned grid = P.blocks;
    if (c->prof_on && c->prof_lambda && !fork && !which && mode == 0) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      QR_CHECK(c, hipEventCreate(&e0));
      QR_CHECK(c, hipEventCreate(&e1));
      hipExtLaunchKernelGGL(k_lambda_u, dim3(grid), dim3(64 * QR_LU_W), lds, st, e0, e1, 0, A, P,
                            (const uint32_t *)c->d_lu_list[which]);
      c->prof_events_child.push_back({e0, e1});
    } else
      hipLaunchKernelGGL(k_lambda_u, dim3(grid), dim3(64 * QR_LU_W), lds, st, A, P,
                         (const uint32_t *)c->d_lu_list[which]);
    QR_CHECK(c, hipGetLastError());
  }
  // the longest-running launches first: the largest class, then down
  for (size_t k = classes.size(); k-- > 0;) {
    const qr_ctx::QClass &cl = classes[k];
    const size_t nmax = ((size_t)cl.nmax + 3) & ~(size_t)3;
    const size_t kshort = std::min(kacc, nmax);  // nmax is a multiple of 4: stays even
    const size_t lds = lambda_lds(nmax, kshort, sampled);
    if (lds > c->attr_lambda_lds) {
      QR_CHECK(c, hipFuncSetAttribute((const void *)k_lambda<false, 1>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      QR_CHECK(c, hipFuncSetAttribute((const void *)k_lambda<false, 1, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      QR_CHECK(c, hipFuncSetAttribute((const void *)k_lambda<false, 4>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      QR_CHECK(c, hipFuncSetAttribute((const void *)k_lambda<false, 16>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      c->attr_lambda_lds = lds;
    }
    hipStream_t st;
    int rc = stream_for(li++, &st);
    if (rc) return rc;
    const uint32_t *qlist = c->qclass_identity[which] ? (const uint32_t *)nullptr
                                                      : (const uint32_t *)(c->d_qclass[which] + cl.first);
    // long queries: sixteen / four waves each (not with a sample: the cleaning is one wave's code)
    const dim3 g(cl.count);
    if (cl.nmax > 512 && !sampled)
      hipLaunchKernelGGL((k_lambda<false, 16>), g, dim3(1024), lds, st, A, (uint32_t)nmax, (uint32_t)kshort, qlist,
                         (char *)nullptr, (size_t)0);
    else if (cl.nmax > QR_LAMBDA_W4_FROM && !sampled)
      hipLaunchKernelGGL((k_lambda<false, 4>), g, dim3(256), lds, st, A, (uint32_t)nmax, (uint32_t)kshort, qlist,
                         (char *)nullptr, (size_t)0);
    else if (c->prof_on && c->prof_lambda && !fork && !which && mode == 0) {
      // bench.py's roofline_lambda: HIP events on the launch itself (qr_prof_enable bit 2; read
      // with qr_prof_get_child -- the two timings share the slot and exclude each other)
      hipEvent_t e0 = nullptr, e1 = nullptr;
      QR_CHECK(c, hipEventCreate(&e0));
      QR_CHECK(c, hipEventCreate(&e1));
      if (cl.nmax <= 128)
        hipExtLaunchKernelGGL((k_lambda<false, 1, true>), g, dim3(64), lds, st, e0, e1, 0, A, (uint32_t)nmax,
                              (uint32_t)kshort, qlist, (char *)nullptr, (size_t)0);
      else
        hipExtLaunchKernelGGL((k_lambda<false, 1>), g, dim3(64), lds, st, e0, e1, 0, A, (uint32_t)nmax,
                              (uint32_t)kshort, qlist, (char *)nullptr, (size_t)0);
      c->prof_events_child.push_back({e0, e1});
    } else if (cl.nmax <= 128)  // (short queries only: the variant without the long queries' paths)
      hipLaunchKernelGGL((k_lambda<false, 1, true>), g, dim3(64), lds, st, A, (uint32_t)nmax, (uint32_t)kshort, qlist,
                         (char *)nullptr, (size_t)0);
    else
      hipLaunchKernelGGL((k_lambda<false, 1>), g, dim3(64), lds, st, A, (uint32_t)nmax, (uint32_t)kshort, qlist,
                         (char *)nullptr, (size_t)0);
    QR_CHECK(c, hipGetLastError());
  }
  if (nlong) {
    hipStream_t st;
    int rc = stream_for(li++, &st);
    if (rc) return rc;
    hipLaunchKernelGGL((k_lambda<true, 1>), dim3((unsigned)nlong), dim3(64), 0, st, A, (uint32_t)nmax_long,
                       (uint32_t)std::min(kacc, nmax_long), (const uint32_t *)c->d_long_list[which], c->d_lscratch,
                       lstride);
    QR_CHECK(c, hipGetLastError());
  }
  if (fork)  // join: the context's stream carries on when every launch is done
    for (size_t a = 0; a < std::min<size_t>(QR_LAMBDA_AUX, li - 1); ++a) {
      QR_CHECK(c, hipEventRecord(c->aux_join[a], c->aux_stream[a]));
      QR_CHECK(c, hipStreamWaitEvent(c->stream, c->aux_join[a], 0));
    }
  return QR_OK;
}

int qr_k_residual(qr_ctx *c) {
  const unsigned grid = (unsigned)((c->N + QR_SLICE - 1) / QR_SLICE);
  hipLaunchKernelGGL(k_residual, dim3(grid), dim3(256), 0, c->stream, c->d_labels,
                     c->d_scores, c->d_lambda, (uint32_t)c->N, c->d_ssq, c->d_qmax);
  c->nqmax = grid;
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// nss > 0: reduce ssq[nss] into root_ss and derive the scale; the per-query
// metric of set `which` (encoded in the sign: nss == 0 means metric only).
// the arguments of the prep workgroups for the scalars of this iteration (`nss` entries of
// d_ssq / d_qmax; `with_metric`: the per-query metric too; `publish`: into the pinned block)
void qr_k_prep_job(qr_ctx *c, size_t nss, int with_metric, int publish, QrPrepJob *j) {
  j->ssq = nss ? c->d_ssq : (const double *)nullptr;
  j->nss = (uint32_t)nss;
  j->qmetric = with_metric ? c->d_qmetric : (const double *)nullptr;
  j->nq = with_metric ? (uint32_t)c->Q : 0u;
  j->qmax = nss ? c->d_qmax : (const double *)nullptr;
  j->nmx = nss ? (uint32_t)c->nqmax : 0u;
  j->scal = c->d_scalars;
  j->reset_max = c->dmode ? 0 : 1;
  j->host_copy = publish ? &c->d_pin->scal : (QrScalars *)nullptr;
  j->seq = publish ? qr_next_scal_seq(c) : 0;
  j->part = c->d_prep_part;
  j->ticket = reinterpret_cast<uint32_t *>(c->d_prep_part + 64);
  // (whoever finishes an iteration's scalars clears the set the NEXT lambda pass fills)
  j->zero_slots = c->dmode ? (unsigned long long *)nullptr : qr_prep_slots(c, c->prep_parity ^ 1);
  j->slots = nullptr;
  j->nwg = 16;
}

int qr_k_prep(qr_ctx *c, size_t nss, int with_metric, int publish) {
  QrPrepJob j;
  qr_k_prep_job(c, nss, with_metric, publish, &j);
  hipLaunchKernelGGL(k_prep, dim3(16), dim3(64), 0, c->stream, j);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// a lambda pass whose scalars have not been finished yet (qr_lambda_compute defers them to
// the root scan launch of the tree that follows): finish them now, in a launch of their own
int qr_k_prep_flush(qr_ctx *c) {
  if (!c->prep_deferred) return QR_OK;
  c->prep_deferred = false;
  return qr_k_prep(c, c->prep_nss, c->prep_with_metric, c->prep_publish);
}

// ---------------------------------------------------------------------------
// Document-sharded contexts: the four per-iteration scalars of every rank are
// exchanged through a [world][4] int64 buffer (own row filled, zeros elsewhere:
// a sum all-reduce is then an all-gather of the bit patterns) and combined in
// rank order, so that every rank derives the same quantisation scale, root
// statistics and metric.
// ---------------------------------------------------------------------------
__global__ void k_scal_pack(const QrScalars *__restrict__ scal, long long *__restrict__ x,
                            const int rank, const int world) {
  const int i = threadIdx.x;
  if (i >= 4 * world) return;
  long long v = 0;
  if (i == 4 * rank) v = (long long)scal->maxabs_bits;
  if (i == 4 * rank + 1) v = __double_as_longlong(scal->root_ss);
  if (i == 4 * rank + 2) v = __double_as_longlong(scal->root_sum);
  if (i == 4 * rank + 3) v = __double_as_longlong(scal->metric_sum);
  x[i] = v;
}

__global__ void k_scal_global(QrScalars *__restrict__ scal, const long long *__restrict__ x,
                              const int world, QrScalars *__restrict__ host_copy, const int seq) {
  if (threadIdx.x != 0) return;
  double mx = 0.0, ss = 0.0, sm = 0.0, ms = 0.0;
  for (int r = 0; r < world; ++r) {
    const double m = __longlong_as_double(x[4 * r]);
    mx = m > mx ? m : mx;
    ss += __longlong_as_double(x[4 * r + 1]);
    sm += __longlong_as_double(x[4 * r + 2]);
    ms += __longlong_as_double(x[4 * r + 3]);
  }
  scal->root_ss = ss;
  scal->root_sum = sm;
  scal->metric_gsum = ms;
  scal->maxabs_bits = 0;  // ready for the next iteration's atomicMax
  int e = 0;
  if (mx > 0.0) frexp(mx, &e);
  e = QR_QBITS - e;
  scal->scale_exp = e;
  scal->scale = ldexp(1.0, e);
  scal->inv_scale = ldexp(1.0, -e);
  scalars_to_host(host_copy, scal, (int)seq);
  __threadfence_system();
  __hip_atomic_store(&host_copy->pad, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// local reductions (sum of squares / sum / metric) + pack for the exchange
int qr_k_prep_pack(qr_ctx *c) {
  if (4 * c->world > 1024) QR_FAIL(c, QR_ERR_UNSUPPORTED, "world too large");
  hipLaunchKernelGGL(k_scal_pack, dim3(1), dim3(4 * c->world), 0, c->stream, c->d_scalars,
                     c->d_xscal, c->rank, c->world);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_prep_global(qr_ctx *c) {
  hipLaunchKernelGGL(k_scal_global, dim3(1), dim3(64), 0, c->stream, c->d_scalars,
                     c->d_xscal, c->world, &c->d_pin->scal, qr_next_scal_seq(c));
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_metric_reduce(qr_ctx *c, int which) {
  QrPrepJob j;
  qr_k_prep_job(c, 0, 0, 0, &j);
  j.qmetric = which ? c->d_vqmetric : c->d_qmetric;
  j.nq = (uint32_t)(which ? c->vQ : c->Q);
  j.reset_max = 0;
  j.zero_slots = nullptr;
  hipLaunchKernelGGL(k_prep, dim3(16), dim3(64), 0, c->stream, j);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}
