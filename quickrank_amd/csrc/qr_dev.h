// qr_dev.h -- small device functions shared by the tree kernels (k_tree.hip: u8 bins,
// LDS histograms) and the wide-bin kernels (k_wide.hip: more than 255 thresholds).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;

__device__ __forceinline__ long long quantize(double x) {
  // round-to-nearest-even of |x| < 2^51 via the 1.5*2^52 trick
  const double magic = 6755399441055744.0;
  return __double_as_longlong(x + magic) - __double_as_longlong(magic);
}

struct Best {
  double score;
  uint32_t t;
};

__device__ __forceinline__ Best best_pick(Best a, Best b) {
  // first max: higher score wins, equal scores -> lower slot (rt.cc:285)
  if (b.score > a.score || (b.score == a.score && b.t < a.t)) return b;
  return a;
}

// gain of slot t of a node: rt.cc:268-291
__device__ __forceinline__ Best slot_gain(long long cs, uint32_t cc, long long S,
                                          uint32_t C, uint32_t t, uint32_t tsize,
                                          u64 minls, double inv_scale) {
  Best b;
  b.score = -1.0;
  b.t = 0xFFFFFFFFu;
  const u64 lc = cc, rc = (u64)C - cc;
  if (t < tsize && lc >= minls && rc >= minls) {
    const double s = (double)S * inv_scale;
    const double lsum = (double)cs * inv_scale;
    const double rsum = s - lsum;
    const double score = lsum * lsum / (double)lc + rsum * rsum / (double)rc;
    if (score > -1.0) {
      b.score = score;
      b.t = t;
    }
  }
  return b;
}

