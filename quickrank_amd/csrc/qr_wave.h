// qr_wave.h -- wave64 reductions and scans on DPP (gfx950): no LDS round trip, no
// ds_bpermute latency chain.  Every helper has a fixed association order, so f64
// results are deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// quad butterflies, row_half_mirror, row_mirror give every lane its 16-lane row
// total; the four row totals are combined in a fixed order through readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
  double o = dpp_f64<0xB1>(v);
  v = o > v ? o : v;
  o = dpp_f64<0x4E>(v);
  v = o > v ? o : v;
  o = dpp_f64<0x141>(v);
  v = o > v ? o : v;
  o = dpp_f64<0x140>(v);
  v = o > v ? o : v;
  const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32),
               d = readlane_f64(v, 48);
  const double ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);
  v = o < v ? o : v;
  o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);
  v = o < v ? o : v;
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0),
                 b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                 c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32),
                 d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}

// Inclusive scan over the 64 lanes: row_shr 1/2/4/8 inside each 16-lane row
// (lanes shifted in from outside the row contribute the identity 0), then the
// carries of the three preceding rows through readlane.
template <int CTRL>
__device__ __forceinline__ long long dpp_shr_i64(long long v) {
  int lo = (int)(uint32_t)(unsigned long long)v, hi = (int)(uint32_t)((unsigned long long)v >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);  // bound_ctrl: out-of-row reads 0
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return (long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ long long readlane_i64(long long v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(unsigned long long)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)v >> 32), l);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long wave_scan_i64(long long v) {
  v += dpp_shr_i64<0x111>(v);  // row_shr:1
  v += dpp_shr_i64<0x112>(v);  // row_shr:2
  v += dpp_shr_i64<0x114>(v);  // row_shr:4
  v += dpp_shr_i64<0x118>(v);  // row_shr:8
  const long long r0 = readlane_i64(v, 15), r1 = readlane_i64(v, 31), r2 = readlane_i64(v, 47);
  const int row = (threadIdx.x & 63) >> 4;
  return v + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
}
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15),
                 r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                 r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
  const int row = (threadIdx.x & 63) >> 4;
  return v + (row > 0 ? r0 : 0u) + (row > 1 ? r1 : 0u) + (row > 2 ? r2 : 0u);
}
