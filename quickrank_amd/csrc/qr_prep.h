// qr_prep.h -- the per-iteration scalars (quantisation scale, the root's sum and sum of
// squares, the ranking's metric) and the two places they are finished in:
//   * k_prep (k_lambda.hip): a launch of its own, sixteen one-wave workgroups and a ticket;
//   * the same sixteen workgroups riding at the end of the ROOT k_redscan launch of batched
//     and level-wise growth (k_tree.hip), when the lambda pass left the maximum in the
//     iteration's slots (below): one launch less on the chain of every boosting iteration.
// Same additions in the same order either way: the values are bit for bit the same.
#pragma once
#include "qr_internal.h"
#include "qr_wave.h"

// The iteration's max |pseudo-response| in QR_PREP_SLOTS words (bits of non-negative
// doubles, which order like the integers): query q's workgroup of k_lambda takes the
// maximum into word q % QR_PREP_SLOTS with one atomic that returns nothing (ten thousand
// atomics on ONE address queued for half of that launch; on 64 they are ~150 each),
// every wave that needs the scale before the scalars are finished -- the root histogram
// launch, the root scan -- reads the 64 words (one per lane) and derives it itself.  Two
// sets: the workgroup that finishes the scalars of iteration t clears the set of t + 1
// (nobody reads or writes it by then; the readers of set t run beside that workgroup).
#define QR_PREP_SLOTS 64

// 2^e with e = QR_QBITS - (exponent of the maximum): what k_prep derives (mart.cc has no
// counterpart: the fixed-point histogram cells are the device's own, qr_internal.h)
__device__ __forceinline__ int qr_scale_exp(const double mx) {
  int x = 0;
  if (mx > 0.0) frexp(mx, &x);  // mx = m * 2^x, m in [0.5, 1)  =>  mx < 2^x
  return QR_QBITS - x;
}
// (every lane passes the slot word it loaded: slots[lane])
__device__ __forceinline__ int qr_slot_scale_exp(const unsigned long long lane_word) {
  return qr_scale_exp(wave_max(__longlong_as_double((long long)lane_word)));
}

__device__ __forceinline__ void scalars_to_host(QrScalars *__restrict__ host_copy, const QrScalars *__restrict__ scal,
                                                const int seq) {
  host_copy->maxabs_bits = scal->maxabs_bits;
  host_copy->scale_exp = scal->scale_exp;
  host_copy->scale = scal->scale;
  host_copy->inv_scale = scal->inv_scale;
  host_copy->root_ss = scal->root_ss;
  host_copy->root_sum = scal->root_sum;
  host_copy->metric_sum = scal->metric_sum;
  host_copy->metric_gsum = scal->metric_gsum;
  host_copy->tag = qr_scal_tag((unsigned long long)__double_as_longlong(scal->metric_sum),
                               (unsigned long long)__double_as_longlong(scal->metric_gsum), seq);
}

// what a prep workgroup needs (kernel argument of k_prep and of k_redscan)
struct QrPrepJob {
  const double *ssq;        // [nss][2] per-query / per-slice (sum of squares, sum), or null
  const double *qmetric;    // [nq] per-query metric, or null
  const double *qmax;       // [nmx] per-query / per-slice max |pseudo-response|
  QrScalars *scal;
  QrScalars *host_copy;     // pinned block the finished scalars are published to, or null
  double *part;             // [16][4] workgroup partials
  uint32_t *ticket;
  unsigned long long *zero_slots;  // the NEXT iteration's slot set (cleared), or null
  const unsigned long long *slots; // this iteration's slot set (the scan workgroups' scale), or null
  uint32_t nss, nq, nmx;
  int32_t reset_max, seq;
  int32_t nwg;              // prep workgroups in the launch (0: none ride in this launch)
};

// Fixed-order reduction of the per-query / per-slice partials + the quantisation scale for
// the histogram accumulators.  The order is the one a single workgroup of 1024 threads
// gives -- thread t adds elements t, t + 1024, ..., a wave adds its lanes, the 16 waves'
// sums are added in wave order -- but every "wave" is a workgroup of its own here (16
// workgroups of 64 threads: one CU took 10 us to pull 10,000 queries' values through its
// memory pipe, 34 us for 80,000), and the one that arrives last at the ticket adds the 16
// partials and finishes the scalars.  Bit for bit the single-workgroup values.
// DEPTH rounds of loads leave together, then their additions in the usual order (16 in the
// launch of its own; 8 inside k_redscan, whose 1024-thread workgroups have 128 registers).
// Called by ONE wave (threads 0..63 of the workgroup; the others have left).
template <int DEPTH>
__device__ __forceinline__ void prep_body(const QrPrepJob &j, const uint32_t bid, const uint32_t nb) {
  const double *__restrict__ ssq = j.ssq;
  const double *__restrict__ qmetric = j.qmetric;
  const double *__restrict__ qmax = j.qmax;
  QrScalars *__restrict__ scal = j.scal;
  const uint32_t nss = j.nss, nq = j.nq, nmx = j.nmx;
  double a = 0.0, b = 0.0, a2 = 0.0;
  double m = 0.0;  // max |pseudo-response| over the per-query / per-slice maxima
  // one loop, so that the three arrays' loads are in flight together (same additions in
  // the same order per accumulator as three loops)
  const uint32_t nall = nss > nq ? (nss > nmx ? nss : nmx) : (nq > nmx ? nq : nmx);
  const uint32_t slot = bid * 64 + threadIdx.x;
  for (uint32_t i0 = slot; i0 < nall; i0 += DEPTH * 1024) {
    double2 v[DEPTH];
    double w[DEPTH], x[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      const uint32_t i = i0 + k * 1024;
      v[k] = i < nss ? *reinterpret_cast<const double2 *>(ssq + 2 * i) : make_double2(0.0, 0.0);
      w[k] = i < nq ? qmetric[i] : 0.0;
      x[k] = i < nmx ? qmax[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      const uint32_t i = i0 + k * 1024;
      if (i < nss) {
        a += v[k].x;
        a2 += v[k].y;
      }
      if (i < nq) b += w[k];
      if (i < nmx) m = fmax(m, x[k]);
    }
  }
  m = wave_max(m);
  a = wave_sum(a);
  a2 = wave_sum(a2);
  b = wave_sum(b);
  __shared__ uint32_t sh_last;
  double *part = j.part;
  if (threadIdx.x == 0) {
    // (agent scope: the sixteen workgroups sit on different XCDs, one L2 each)
    __hip_atomic_store(&part[4 * bid], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&part[4 * bid + 1], a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&part[4 * bid + 2], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&part[4 * bid + 3], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_last = __hip_atomic_fetch_add(j.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nb - 1;
  }
  __syncthreads();
  if (!sh_last) return;
  if (j.zero_slots && threadIdx.x < QR_PREP_SLOTS) j.zero_slots[threadIdx.x] = 0ull;
  if (threadIdx.x == 0) {
    __hip_atomic_store(j.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
    double ta = 0.0, ta2 = 0.0, tb = 0.0, tm = 0.0;
    for (uint32_t i = 0; i < nb; ++i) {
      ta += __hip_atomic_load(&part[4 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ta2 += __hip_atomic_load(&part[4 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tb += __hip_atomic_load(&part[4 * i + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tm = fmax(tm, __hip_atomic_load(&part[4 * i + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (ssq) {
      scal->root_ss = ta;
      scal->root_sum = ta2;
    }
    if (qmetric) scal->metric_sum = tb;
    if (ssq) {
      // (maxabs_bits: what qr_pseudo_set or a document-sharded exchange put there)
      double mx = fmax(__longlong_as_double((long long)scal->maxabs_bits), tm);
      scal->maxabs_bits = (unsigned long long)__double_as_longlong(mx);
      const int e = qr_scale_exp(mx);
      scal->scale_exp = e;
      scal->scale = ldexp(1.0, e);
      scal->inv_scale = ldexp(1.0, -e);
      // ready for the next iteration's atomicMax (document-sharded contexts still
      // have to pack it for the exchange: k_scal_global clears it there)
      if (j.reset_max) scal->maxabs_bits = 0;
    }
    // read-back without a copy launch: the finished scalars go straight into the pinned
    // host block; `pad` = the launch's sequence number, stored LAST behind a system-scope
    // fence: the host polls it (wait_seq_impl in qr_api.hip) instead of waiting for an event
    if (j.host_copy) {
      scalars_to_host(j.host_copy, scal, (int)j.seq);
      __threadfence_system();
      __hip_atomic_store(&j.host_copy->pad, j.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
