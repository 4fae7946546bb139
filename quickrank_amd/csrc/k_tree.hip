// k_tree.hip -- regression-tree construction on the device (gfx950).
//
// Stands behind RTNodeHistogram::update / child ctor / sibling subtraction
// (rtnode_histogram.cc:41-87, 172-217), RegressionTree::fit + split
// (rt.cc:49-90, 209-362), MaxHeap (maxheap.h:58-88), RTNode stats
// (rtnode.h:97-107), RTNode::save_leaves (rtnode.cc:34-46),
// RegressionTree::update_output (rt.cc:165-207) and Mart::update_modelscores
// (mart.cc:447-468).
//
// Kernels (one stream, no host round trip inside a tree):
//   k_hist       histogram build of one node over the rank's 64-feature blocks:
//                LDS-resident [bin][feature] cells, one ds_add_u64 per
//                (doc, feature) carrying gradient AND count (fixed point, see
//                qr_internal.h), bank-conflict-free by construction
//   k_reduce     sum of the workgroup partials in their native cell order
//   k_scan       per feature: prefix-sum over the 256 slots, sibling (parent -
//                child), split gain of every slot, first best one
//   k_decide     control: merge of the per-feature bests (two waves), node
//                statistics, max-deviance heap growth, next split descriptor
//                (k_merge publishes the rank's bests instead on feature-sharded runs)
//   k_partition  single-pass stable partition of the node's document list with a
//                look-back chain (k_mask: the owner's go-left bits on feature-sharded
//                runs; document-sharded runs take the rank's own left count from
//                the prefix of its local counts, k_scan)
//   k_finish / k_leaf_sums / k_leaf_final (k_leaf_global) / k_score_update(_walk) /
//   k_valid_update: leaves, leaf outputs, score updates
//   k_obl_fill / k_obl_plan + k_partition_level / k_hist_level /
//   k_redscan_level: level-batched oblivious growth (ot.cc:32-201)
//   k_decide_part (k_decide_batch + k_partition_batch for larger trees) / k_hist_batch /
//   k_redscan: leaf-wise growth on one GPU, up to QR_BATCH splits per step; control step
//   and partition in one launch, reduce and scan in one launch
#include <hip/hip_ext.h>

#include <algorithm>

#include "qr_internal.h"
#include "qr_wave.h"
#include "qr_dev.h"
#include "qr_prep.h"

// ===========================================================================
// k_hist
// ===========================================================================
// Bank-conflict-free LDS accumulation.
//   lane = (doc slot dsub = lane / CH, 16-feature chunk c = lane % CH)
//   each lane loads the 16 bin ids (16 B) of its chunk of its doc, then issues
//   16 ds_add_u64.  At step k lane L touches column 16c + ((k + L) & 15):
//   the 16 lanes of a hardware LDS lane group (L & ~15 .. +15) therefore hit 16
//   distinct (column mod 16) values; with the [bin][fw] cell layout (fw a
//   multiple of 16, 8-byte cells) that is 16 distinct bank pairs whatever the
//   bins are.  Same-cell collisions between groups / waves are what the
//   atomic is for.

#ifndef QR_HIST_SETS
#define QR_HIST_SETS 3
#endif
// (the child launches' variant keeps its column offsets packed and has room for a deeper pipeline)
#ifndef QR_HIST_SETS_CHILD
#define QR_HIST_SETS_CHILD 3
#endif

// the control lane's helpers: inlined, or (-DQR_CTRL_NOINLINE, an experiment) shared calls
#ifdef QR_CTRL_NOINLINE
#define QR_CTRL_FN __noinline__
#else
#define QR_CTRL_FN __forceinline__
#endif

// -DQR_DEBUG_CHECKS (tests/tools/abort_hunt.py --debug builds the library with
// it): every indexed store of the growth kernels that is not bounded by construction in plain
// sight -- the partition's scatter, the node / heap / split-log entries the control lane creates,
// the partial slot a histogram workgroup flushes into, the node-histogram slot a scan writes --
// is checked against the capacity the host allocated; the first violation is recorded (code,
// two values) and the store SKIPPED, and qr_debug_check reports it after the launch.  The
// product build compiles the checks out.
#ifdef QR_DEBUG_CHECKS
__device__ unsigned long long qr_dbg_word[4];   // [0] first failure, [1] failures, [2], [3] its values
__device__ unsigned long long qr_dbg_caps[8];   // [0] d_partials [1] its slots [2] d_lpartials [3] its slots [4] histogram slots
__device__ __noinline__ bool qr_dbg_fail(const unsigned code, const unsigned long long a, const unsigned long long b) {
  if (atomicCAS(&qr_dbg_word[0], 0ull, ((unsigned long long)code << 32) | (blockIdx.x & 0xffffffffu)) == 0ull) {
    qr_dbg_word[2] = a;
    qr_dbg_word[3] = b;
  }
  atomicAdd(&qr_dbg_word[1], 1ull);
  return false;
}
#define QR_DBG_OK(cond, code, a, b) ((cond) ? true : qr_dbg_fail((code), (unsigned long long)(a), (unsigned long long)(b)))
// the capacities as they are NOW: called where a tree starts (its buffers have just been sized)
static int qr_dbg_caps_upload(qr_ctx *c) {
  const unsigned long long caps[8] = {(unsigned long long)c->d_partials, c->partial_slots, (unsigned long long)c->d_lpartials,
                                      c->lslots_cap, c->hist_slots, 0, 0, 0};
  QR_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(qr_dbg_caps), caps, sizeof(caps)));
  return QR_OK;
}
#define QR_DBG_CAPS(c) do { const int drc_ = qr_dbg_caps_upload(c); if (drc_) return drc_; } while (0)
#else
#define QR_DBG_CAPS(c) do {} while (0)
#define QR_DBG_OK(cond, code, a, b) true
#endif

// -DQR_WG_JITTER (round 6; tests/tools/abort_hunt.py --jitter builds the library with it): the
// inter-workgroup race stress.  One workgroup in eight of EVERY launch of this file -- which ones
// changes from launch to launch -- starts ~30 us late (s_sleep), the way workgroups of one launch
// drift apart when several processes share the GPU (the r06 hunt met the intermittent mismatch of
// rounds 4-5 fifty times more often with eight processes on the device).  A workgroup that reads
// what an earlier-finishing workgroup of the SAME launch has already overwritten then does so every
// time, not once in thousands of runs.  The product build compiles it out.
#ifdef QR_WG_JITTER
__device__ unsigned int qr_jitter_seq;
__device__ __noinline__ void qr_jitter_wait() {
  for (int i = 0; i < 9; ++i) __builtin_amdgcn_s_sleep(127);
}
#define QR_JITTER()                                                                              \
  do {                                                                                           \
    const unsigned int jq_ = __builtin_amdgcn_readfirstlane(qr_jitter_seq);                      \
    if ((((blockIdx.x + blockIdx.y * 7u + jq_) * 2654435761u) >> 29) == 3u) qr_jitter_wait();   \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) qr_jitter_seq = jq_ + 1u;        \
  } while (0)
#else
#define QR_JITTER() do {} while (0)
#endif

// -DQR_STEP_TIMING (scripts/step_timing.py): clock64 stamps of one histogram workgroup's
// sections, with the load queue drained at the first two so that they show the dependent
// round trips (descriptor -> ids -> rows) one by one
#ifdef QR_STEP_TIMING
__device__ long long qr_ht[8];
#define QR_HT(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) qr_ht[i] = clock64(); } while (0)
#define QR_HT_WAIT(i) do { __builtin_amdgcn_s_waitcnt(0); QR_HT(i); } while (0)
#else
#define QR_HT(i)
#define QR_HT_WAIT(i)
#endif

// SUMS: the lanes of chunk 0 also add up their documents' pseudo-responses and squares
// (sq, sm): the directly built child's `sum` / `squares_sum_` (rtnode.h:97-107) come out of
// the pass that reads every lambda[id] of the child anyway, instead of a second gather of
// them (8 bytes out of a cold 64-byte line each) in the partition.  The two accumulators do
// not fit next to the sixteen precomputed column offsets in the 128 registers of a
// 1024-thread workgroup, so this variant keeps the columns packed, four to a register, and
// extracts the one a step needs (one more VALU instruction under an LDS-bound loop; the
// empty asm keeps the compiler from hoisting the extractions back into sixteen registers).
template <int CH, bool IDENTITY, bool SUMS = false>
__device__ __forceinline__ void hist_accumulate(
    u64 *__restrict__ hist, const uint8_t *__restrict__ bins_b,
    const uint32_t *__restrict__ order, const uint32_t seg_begin, const uint32_t r0,
    const uint32_t r1, const double *__restrict__ lambda, const double scale_in, double *sq = nullptr,
    double *sm = nullptr, const u64 slot_word = 0, const bool use_slot = false) {
  // (the caller has NOT zeroed `hist`: that happens below, behind the first loads)
  // use_slot (the root launch when the iteration's scalars are not finished yet, qr_prep.h):
  // the scale comes from the slot word this lane loaded at the top of the kernel; it is
  // derived below, behind the first tiles' requests, so that no load waits for it
  double scale = scale_in;
  constexpr int FW = 16 * CH;
  constexpr int DW = 64 / CH;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int dsub = lane / CH;
  const int c = lane - dsub * CH;
  const bool lane_ok = dsub < DW;
  const int r = lane & 15;
  uint32_t colk[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) colk[k] = 16 * c + ((k + r) & 15);
  const int dr = r >> 2;       // dword rotation
  const uint32_t br = r & 3;   // byte rotation inside a dword
  // (SUMS: the sixteen column indices packed into four registers, a byte each.  Round 6 measured the
  // packing in EVERY variant -- no spill in k_hist_root, 90 VGPRs -- and the root launch got slower,
  // 39.4 -> 40.9 us: profiles/r06_child_hist_bound.md)
  constexpr bool PACKED = SUMS;
  uint32_t cp[4] = {0, 0, 0, 0};
  if (PACKED) {
#pragma unroll
    for (int k = 0; k < 16; ++k) cp[k >> 2] |= colk[k] << (8 * (k & 3));
  }
  auto process = [&](const uint4 &row, const double lam) {
    uint32_t c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
    if (PACKED) asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));  // (not hoisted back into sixteen registers)
    if (SUMS) {
      if (c == 0) {
        *sq += lam * lam;
        *sm += lam;
      }
    }
    const uint32_t cpk[4] = {c0, c1, c2, c3};
    const u64 addend = (1ull << QR_SB) + (u64)quantize(lam * scale);
    // rotate the 16 bytes right by r so that byte k of R is byte (k+r)&15
    const uint32_t t0 = (dr & 1) ? row.y : row.x;
    const uint32_t t1 = (dr & 1) ? row.z : row.y;
    const uint32_t t2 = (dr & 1) ? row.w : row.z;
    const uint32_t t3 = (dr & 1) ? row.x : row.w;
    const uint32_t u0 = (dr & 2) ? t2 : t0;
    const uint32_t u1 = (dr & 2) ? t3 : t1;
    const uint32_t u2 = (dr & 2) ? t0 : t2;
    const uint32_t u3 = (dr & 2) ? t1 : t3;
    uint32_t R[4];
    R[0] = __builtin_amdgcn_alignbyte(u1, u0, br);
    R[1] = __builtin_amdgcn_alignbyte(u2, u1, br);
    R[2] = __builtin_amdgcn_alignbyte(u3, u2, br);
    R[3] = __builtin_amdgcn_alignbyte(u0, u3, br);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t bin = (R[k >> 2] >> (8 * (k & 3))) & 0xffu;
      const uint32_t col = PACKED ? ((cpk[k >> 2] >> (8 * (k & 3))) & 0xffu) : colk[k];
      atomicAdd(&hist[bin * FW + col], addend);
    }
  };
  // Software pipeline: NS register sets rotate, so NS-1 tiles of loads stay in
  // flight behind the tile whose 16 LDS atomics are issuing; in the gather case
  // each set also keeps the document id of its NEXT tile in flight.  No register
  // copies between stages (a copy would wait for the load it forwards); the set
  // index is a compile-time constant everywhere, so the sets live in registers.
  constexpr int NS = SUMS ? QR_HIST_SETS_CHILD : QR_HIST_SETS;
  const uint32_t step = nw * DW;
  const uint32_t p0 = r0 + wave * DW + dsub;
  auto valid = [&](uint32_t p) { return lane_ok && p < r1; };
  auto get_id = [&](uint32_t p) -> uint32_t {
    return IDENTITY ? seg_begin + p : order[seg_begin + p];
  };
  auto load_row = [&](uint32_t id) -> uint4 {
    return *reinterpret_cast<const uint4 *>(bins_b + (size_t)id * FW + 16 * c);
  };
  // Loads are unconditional (positions clamped into the range) so that no
  // exec-masked branch surrounds a VMEM instruction: the compiler then emits
  // counted s_waitcnt vmcnt(N) instead of draining the queue at every stage.
  const uint32_t plast = r1 - 1;
  auto clampp = [&](uint32_t p) { return p < plast ? p : plast; };
  uint4 row[NS];
  double lam[NS];
  uint32_t id[NS];
  bool v[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    v[i] = valid(p0 + i * step);
    id[i] = get_id(clampp(p0 + i * step));
  }
  QR_HT_WAIT(1);   // (timing builds only: the ids have arrived)
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    row[i] = load_row(id[i]);
    lam[i] = lambda[id[i]];
  }
  QR_HT_WAIT(2);   // (timing builds only: the first rows have arrived)
#pragma unroll
  for (int i = 0; i < NS; ++i) id[i] = get_id(clampp(p0 + (NS + i) * step));
  // zero the cells while the first tiles are on their way
  for (uint32_t i = threadIdx.x * 2; i < 256u * FW; i += blockDim.x * 2) {
    hist[i] = 0;
    hist[i + 1] = 0;
  }
  if (use_slot) scale = ldexp(1.0, qr_slot_scale_exp(slot_word));
  __syncthreads();
  QR_HT(3);
  uint32_t pos = p0;
  for (uint32_t tile = r0 + wave * DW; tile < r1; tile += NS * step, pos += NS * step) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (v[i]) process(row[i], lam[i]);
      v[i] = valid(pos + (NS + i) * step);
      row[i] = load_row(id[i]);
      lam[i] = lambda[id[i]];
      id[i] = get_id(clampp(pos + (2 * NS + i) * step));
    }
  }
}

template <bool SUMS = false>
__device__ __forceinline__ void hist_run(
    u64 *hist, const uint32_t seg_begin, const uint32_t r0, const uint32_t r1, const int buf,
    const int b, const size_t slot0, const QrBlock *__restrict__ blocks,
    const uint8_t *__restrict__ bins, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, const double *__restrict__ lambda,
    const double scale, u64 *__restrict__ partials, const bool tr = false, const int fw_known = 0,
    const size_t off_known = 0, double *__restrict__ sums_out = nullptr, const u64 slot_word = 0,
    const bool use_slot = false);

// One workgroup's share of a node histogram: workgroup `wg` of the `G` that the
// plan hands to a node of n documents; partial slots start at `slot_base`.
__device__ __forceinline__ void hist_body(
    u64 *hist, const uint32_t seg_begin, const uint32_t n, const int buf, const uint32_t q,
    const int wg, const size_t slot_base, const QrBlock *__restrict__ blocks, const int nblocks,
    const uint8_t *__restrict__ bins, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, const double *__restrict__ lambda,
    const double scale, u64 *__restrict__ partials, const bool tr = false, const u64 slot_word = 0,
    const bool use_slot = false) {
  __shared__ QrPlan plan;
  if (threadIdx.x == 0) qr_make_plan(n, nblocks, blocks, q, &plan);
  __syncthreads();
  int b = -1;
  for (int i = 0; i < nblocks; ++i)
    if (wg >= plan.wg_start[i] && wg < plan.wg_start[i + 1]) b = i;
  if (b < 0) return;
  const uint32_t j = wg - plan.wg_start[b];
  const uint32_t per = plan.per[b];
  const uint32_t r0 = j * per;
  const uint32_t r1 = (r0 + per < n) ? r0 + per : n;
  hist_run(hist, seg_begin, r0, r1, buf, b, slot_base + (size_t)wg * plan.kmax, blocks, bins, order0,
           order1, lambda, scale, partials, tr, 0, 0, nullptr, slot_word, use_slot);
}

// positions [r0, r1) of the segment at seg_begin, for feature block b; partial slots
// slot0, slot0 + 1, ... (one per QR_DPW documents).  SUMS (k_hist_batch: document lists
// only): every workgroup also adds up its documents' pseudo-responses and their squares
// and leaves the pair at sums_out[2 * slot0] (see hist_accumulate).
template <bool SUMS>
__device__ __forceinline__ void hist_run(
    u64 *hist, const uint32_t seg_begin, const uint32_t r0, const uint32_t r1, const int buf,
    const int b, const size_t slot0, const QrBlock *__restrict__ blocks,
    const uint8_t *__restrict__ bins, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, const double *__restrict__ lambda,
    const double scale, u64 *__restrict__ partials, const bool tr, const int fw_known,
    const size_t off_known, double *__restrict__ sums_out, const u64 slot_word, const bool use_slot) {
  // (batched growth hands the block's geometry over with the workgroup's share: one
  // dependent read less before the first bins can be requested)
  const int fw = fw_known ? fw_known : blocks[b].fw;
  const uint8_t *bins_b = bins + (fw_known ? off_known : blocks[b].off);
  const uint32_t *order = buf == 0 ? order0 : order1;
  const bool identity = buf == 2;
  const uint32_t cells = 256u * fw;
  uint32_t k = 0;
  double sq = 0.0, sm = 0.0;  // (sums_out) this lane's documents of chunk 0, in list order
  for (uint32_t s0 = r0; s0 < r1; s0 += QR_DPW, ++k) {
    const uint32_t s1 = (s0 + QR_DPW < r1) ? s0 + QR_DPW : r1;
    if (SUMS) {
      switch (fw) {
        case 16: hist_accumulate<1, false, true>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, &sq, &sm); break;
        case 32: hist_accumulate<2, false, true>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, &sq, &sm); break;
        case 48: hist_accumulate<3, false, true>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, &sq, &sm); break;
        default: hist_accumulate<4, false, true>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, &sq, &sm); break;
      }
    } else
    if (identity) {
      switch (fw) {
        case 16: hist_accumulate<1, true>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, nullptr, nullptr, slot_word, use_slot); break;
        case 32: hist_accumulate<2, true>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, nullptr, nullptr, slot_word, use_slot); break;
        case 48: hist_accumulate<3, true>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, nullptr, nullptr, slot_word, use_slot); break;
        default: hist_accumulate<4, true>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, nullptr, nullptr, slot_word, use_slot); break;
      }
    } else {
      switch (fw) {
        case 16: hist_accumulate<1, false>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, nullptr, nullptr, slot_word, use_slot); break;
        case 32: hist_accumulate<2, false>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, nullptr, nullptr, slot_word, use_slot); break;
        case 48: hist_accumulate<3, false>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, nullptr, nullptr, slot_word, use_slot); break;
        default: hist_accumulate<4, false>(hist, bins_b, order, seg_begin, s0, s1, lambda, scale, nullptr, nullptr, slot_word, use_slot); break;
      }
    }
    __syncthreads();
    QR_HT(4);
    u64 *dst = partials + (slot0 + k) * (256u * 64u);
#ifdef QR_DEBUG_CHECKS
    {
      const u64 cap = (u64)partials == qr_dbg_caps[0] ? qr_dbg_caps[1] : ((u64)partials == qr_dbg_caps[2] ? qr_dbg_caps[3] : ~0ull);
      if (!QR_DBG_OK(slot0 + k < cap, 2, slot0 + k, cap)) {  // (workgroup-uniform)
        __syncthreads();
        continue;
      }
    }
#endif
    if (tr) {
      // feature-major slot [column][256 bins] for k_redscan, which reads one column
      // of every slot.  A wave moves a tile of 16 columns x 8 bins: lane = (column
      // f, bin pair b); the 16 lanes of an LDS lane group read 16 consecutive
      // columns of one bin (conflict-free), and the 4 lanes of a column write 64
      // contiguous bytes.
      const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
      const uint32_t nct = (uint32_t)fw / 16;
      for (uint32_t tile = wave; tile < 32u * nct; tile += nw) {
        const uint32_t col = (tile % nct) * 16 + (lane & 15);
        const uint32_t bin = (tile / nct) * 8 + (lane >> 4) * 2;
        ulonglong2 v;
        v.x = hist[bin * fw + col];
        v.y = hist[(bin + 1) * fw + col];
        *reinterpret_cast<ulonglong2 *>(dst + col * 256u + bin) = v;
      }
    } else
    for (uint32_t i = threadIdx.x * 2; i < cells; i += blockDim.x * 2) {
      ulonglong2 v;
      v.x = hist[i];
      v.y = hist[i + 1];
      *reinterpret_cast<ulonglong2 *>(dst + i) = v;
    }
    __syncthreads();
    QR_HT_WAIT(5);
  }
  if (SUMS) {
    // a lane's documents in list order, the lanes of a wave by wave_sum, the sixteen waves
    // in wave order: fixed, whatever the launch looks like.  One pair per workgroup, at its
    // first partial slot (k_redscan adds the workgroups' pairs in slot order).
    __shared__ double sh_sq[16], sh_sm[16];
    sq = wave_sum(sq);
    sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) {
      sh_sq[threadIdx.x >> 6] = sq;
      sh_sm[threadIdx.x >> 6] = sm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0.0, b2 = 0.0;
      for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
        a += sh_sq[w];
        b2 += sh_sm[w];
      }
      sums_out[2 * slot0] = a;
      sums_out[2 * slot0 + 1] = b2;
    }
  }
}

__global__ __launch_bounds__(1024) void k_hist(
    const QrTreeState *__restrict__ ts, const int root_mode, const uint32_t N,
    const QrBlock *__restrict__ blocks, const int nblocks,
    const uint8_t *__restrict__ bins, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, const double *__restrict__ lambda,
    const QrScalars *__restrict__ scal, u64 *__restrict__ partials, const int docmode,
    const int root_buf) {
  QR_JITTER();
  extern __shared__ __attribute__((aligned(16))) u64 hist[];
  uint32_t seg_begin, n;
  int buf;
  if (root_mode) {  // N documents: all of them (identity, buffer 2) or the sample's list
    seg_begin = 0;
    n = N;
    buf = root_buf;
  } else {
    if (!ts->desc.active) return;
    // document-sharded: the rank's own part of the directly built child
    seg_begin = docmode ? ts->loc.small_begin : ts->desc.small_begin;
    n = docmode ? ts->loc.small_n : ts->desc.small_n;
    buf = ts->desc.dst_buf;
  }
  const uint32_t q =
      qr_plan_quantum((unsigned long long)n * qr_plan_wsum(nblocks, blocks), (int)gridDim.x - nblocks);
  hist_body(hist, seg_begin, n, buf, q, (int)blockIdx.x, 0, blocks, nblocks, bins, order0, order1,
            lambda, scal->scale, partials);
}

// The root launch under its own name, so that a kernel trace lists it apart from
// the (much smaller) child launches: it is the kernel bench.py's roofline is about.
__global__ __launch_bounds__(1024) void k_hist_root(
    const uint32_t N, const QrBlock *__restrict__ blocks, const int nblocks,
    const uint8_t *__restrict__ bins, const uint32_t *__restrict__ order0,
    const double *__restrict__ lambda, const QrScalars *__restrict__ scal,
    u64 *__restrict__ partials, const int root_buf, const int tr,
    const unsigned long long *__restrict__ slots, const QrHistWg *__restrict__ wgs) {
  QR_JITTER();
  extern __shared__ __attribute__((aligned(16))) u64 hist[];
  // `slots`: the iteration's scalars are finished by workgroups of the scan launch BEHIND this
  // one (qr_prep.h); the scale comes from the slot set the lambda pass filled, a word per lane
  const u64 slot_word = slots ? slots[threadIdx.x & (QR_PREP_SLOTS - 1)] : 0ull;
  const double scale = slots ? 0.0 : scal->scale;
  // `wgs`: every workgroup's share of the root ready-made -- the root's plan depends on the
  // number of documents and the blocks only, so the host makes it once (root_shares) and the
  // workgroup asks for its first rows after ONE load, instead of thread 0 planning, a barrier
  // and the block's geometry from memory
  if (wgs) {
    const QrHistWg d = wgs[blockIdx.x];
    if (d.count == 0) return;
    hist_run<false>(hist, d.begin, 0, d.count, d.buf, d.block, d.slot, blocks, bins, order0, order0, lambda, scale,
                    partials, tr != 0, (int)d.fw, (size_t)d.off256 << 8, nullptr, slot_word, slots != nullptr);
    return;
  }
  const uint32_t q =
      qr_plan_quantum((unsigned long long)N * qr_plan_wsum(nblocks, blocks), (int)gridDim.x - nblocks);
  hist_body(hist, 0, N, root_buf, q, (int)blockIdx.x, 0, blocks, nblocks, bins, order0, order0, lambda,
            scale, partials, tr != 0, slot_word, slots != nullptr);
}

// level-wise (oblivious) growth: the directly built children of ALL nodes of the
// level in one launch; workgroup w serves node map[w] >> 16 as its workgroup
// map[w] & 0xffff
__global__ __launch_bounds__(1024) void k_hist_level(
    const QrTreeState *__restrict__ ts, const uint32_t *__restrict__ map,
    const QrBlock *__restrict__ blocks, const int nblocks,
    const uint8_t *__restrict__ bins, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, const double *__restrict__ lambda,
    const QrScalars *__restrict__ scal, u64 *__restrict__ partials) {
  QR_JITTER();
  extern __shared__ __attribute__((aligned(16))) u64 hist[];
  if (ts->obl_done || blockIdx.x >= ts->l_hist_wgs) return;
  const uint32_t m = map[blockIdx.x];
  const QrLevelNode &ln = ts->lnode[m >> 16];
  hist_body(hist, ln.small_begin, ln.small_n, ln.dst_buf, ln.q, (int)(m & 0xffffu), ln.slot_base,
            blocks, nblocks, bins, order0, order1, lambda, scal->scale, partials, true);
}

// batched leaf-wise growth: the directly built children of the batch's nodes
__global__ __launch_bounds__(1024) void k_hist_batch(
    const QrHistWg *__restrict__ wgs, const QrBlock *__restrict__ blocks,
    const uint8_t *__restrict__ bins, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, const double *__restrict__ lambda,
    const QrScalars *__restrict__ scal, u64 *__restrict__ partials, double *__restrict__ histsum) {
  QR_JITTER();
  extern __shared__ __attribute__((aligned(16))) u64 hist[];
  QR_HT(0);
  const QrHistWg d = wgs[blockIdx.x];
  if (d.count == 0) return;
  // (every workgroup also adds up its documents' pseudo-responses -- k_redscan reads the pairs
  // of feature block 0's workgroups; the others' are the same numbers and cost nothing)
  hist_run<true>(hist, d.begin, 0, d.count, d.buf, d.block, d.slot, blocks, bins, order0, order1, lambda,
                 scal->scale, partials, true, (int)d.fw, (size_t)d.off256 << 8, histsum);
#ifdef QR_STEP_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("hist_batch wg0 (%u docs, grid %u): desc+ids %lld rows %lld zero %lld accumulate %lld flush %lld cycles\n",
           d.count, gridDim.x, qr_ht[1] - qr_ht[0], qr_ht[2] - qr_ht[1], qr_ht[3] - qr_ht[2], qr_ht[4] - qr_ht[3],
           qr_ht[5] - qr_ht[4]);
#endif
}

// ===========================================================================
// k_reduce: sum the workgroup partials of one histogram launch, in the native
// [bin][fw] cell order (fully coalesced 8-byte reads), unpacking count and sum.
// One workgroup = 64 consecutive cells x 4 slot groups; exact integers, so the
// reduction order is free.
// ===========================================================================
// sum of the partial slots [slot_base, ...) of a node of n documents planned with
// G workgroups, for 64 consecutive cells starting at cellblock * 64
__device__ __forceinline__ void reduce_body(
    const uint32_t n, const uint32_t q, const size_t slot_base, const uint32_t cellblock,
    const QrBlock *__restrict__ blocks, const int nblocks, const u64 *__restrict__ partials,
    long long *__restrict__ red_sum, uint32_t *__restrict__ red_cnt, const uint32_t cs,
    uint32_t *__restrict__ red_cnt_loc, const QrPlan *__restrict__ given = nullptr) {
  __shared__ long long sh_s[512];
  __shared__ uint32_t sh_c[512];
  __shared__ QrPlan sh_plan;
  if (!given) {
    if (threadIdx.x == 0) qr_make_plan(n, nblocks, blocks, q, &sh_plan);
    __syncthreads();
  }
  const QrPlan &plan = given ? *given : sh_plan;
  // which block does this workgroup's cell range belong to?
  uint32_t cell0 = cellblock * 64u;  // over the concatenation of 256*fw cells per block
  int b = -1;
  uint32_t base = 0;
  for (int i = 0; i < nblocks; ++i) {
    const uint32_t cells = 256u * blocks[i].fw;
    if (b < 0 && cell0 < base + cells) {
      b = i;
      cell0 -= base;
    }
    if (b < 0) base += cells;
  }
  if (b < 0) return;
  const uint32_t c = threadIdx.x & 63, g = threadIdx.x >> 6;  // 8 slot groups
  const uint32_t per = plan.per[b];
  const int W = plan.wg_start[b + 1] - plan.wg_start[b];
  const int kmax = plan.kmax;
  const int total = W * kmax;
  const u64 *src = partials + (slot_base + (size_t)plan.wg_start[b] * kmax) * (256u * 64u) + cell0 + c;
  long long s = 0;
  uint32_t cn = 0;
#pragma unroll 4
  for (int idx = (int)g; idx < total; idx += 8) {
    bool valid = true;
    if (kmax > 1) {  // workgroup j flushed only ceil(docs_j / QR_DPW) slots
      const int j = idx / kmax, k = idx - j * kmax;
      const uint32_t r0 = j * per;
      const uint32_t r1 = (r0 + per < n) ? r0 + per : n;
      valid = r0 < n && k < (int)((r1 - r0 + QR_DPW - 1) / QR_DPW);
    }
    if (valid) {
      const u64 cell = src[(size_t)idx * (256u * 64u)];
      const u64 cnt = (cell + (1ull << (QR_SB - 1))) >> QR_SB;
      s += (long long)(cell - (cnt << QR_SB));
      cn += (uint32_t)cnt;
    }
  }
  sh_s[threadIdx.x] = s;
  sh_c[threadIdx.x] = cn;
  __syncthreads();
  if (g == 0) {
    s = 0;
    cn = 0;
    for (int i = 0; i < 8; ++i) {
      s += sh_s[i * 64 + c];
      cn += sh_c[i * 64 + c];
    }
    red_sum[base + cell0 + c] = s;
    red_cnt[(size_t)(base + cell0 + c) * cs] = cn;
    if (red_cnt_loc) red_cnt_loc[base + cell0 + c] = cn;  // survives the all-reduce
  }
}

__global__ __launch_bounds__(512) void k_reduce(
    const QrTreeState *__restrict__ ts, const int root_mode, const uint32_t N,
    const QrBlock *__restrict__ blocks, const int nblocks, const int G,
    const u64 *__restrict__ partials, long long *__restrict__ red_sum,
    uint32_t *__restrict__ red_cnt, const int docmode, const double *__restrict__ part_ss,
    long long *__restrict__ tail, const int rank, const int world,
    uint32_t *__restrict__ red_cnt_loc) {
  QR_JITTER();
  uint32_t n;
  if (root_mode) {
    n = N;
  } else {
    if (!ts->desc.active) return;
    n = docmode ? ts->loc.small_n : ts->desc.small_n;
  }
  // document-sharded: counts are int64 cells of the exchange buffer (low word
  // written here, high word stays 0), and the rank's (sum of squares, sum) of the
  // directly built child rides in its own slot of the tail -- zeros elsewhere, so
  // the sum all-reduce doubles as an all-gather of the bit patterns.
  const uint32_t cs = docmode ? 2u : 1u;
  if (docmode && blockIdx.x == 0 && threadIdx.x < 64) {
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
    for (int i = threadIdx.x; i < 2 * world; i += 64) {
      long long v = 0;
      if (i == 2 * rank) v = __double_as_longlong(a);
      if (i == 2 * rank + 1) v = __double_as_longlong(b);
      tail[i] = v;
    }
  }
  const uint32_t q =
      qr_plan_quantum((unsigned long long)n * qr_plan_wsum(nblocks, blocks), G - nblocks);
  reduce_body(n, q, 0, blockIdx.x, blocks, nblocks, partials, red_sum, red_cnt, cs, red_cnt_loc);
}

// ===========================================================================
// k_scan
// ===========================================================================
__device__ __forceinline__ Best block_best(Best v, Best *sh) {
  // lanes are in slot order, so the first lane holding the wave maximum is the
  // first maximum (rt.cc:285); scores are >= 0 or the -1 of "no valid slot"
  const double m = wave_max(v.score);
  const unsigned long long hit = __ballot(v.score == m && v.t != 0xFFFFFFFFu);
  Best w;
  w.score = hit ? m : -1.0;
  w.t = hit ? (threadIdx.x & ~63u) + (uint32_t)__ffsll((long long)hit) - 1u : 0xFFFFFFFFu;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
  __syncthreads();
  Best r = sh[0];
  for (int i = 1; i < 4; ++i) r = best_pick(r, sh[i]);
  return r;
}

__device__ __forceinline__ void scan_core(
    const int root_mode, const int small_slot, const int big_slot, const int parent_slot,
    const int small_is_left, const u64 minls, const int lf, long long s, uint32_t cn, uint32_t cl,
    long long *__restrict__ hsum, uint32_t *__restrict__ hcnt, const int flocal,
    const uint32_t *__restrict__ thr_size, const int32_t *__restrict__ lf2gf,
    const QrScalars *__restrict__ scal, qr_split_t *__restrict__ featrec,
    uint32_t *__restrict__ hcnt_loc, const float my_thr, float *__restrict__ featthr,
    const bool pre = false, const long long pre_s = 0, const uint32_t pre_c = 0,
    const int pre_gf = -1, const uint32_t pre_tsize = 0, const double pre_inv = 0.0);

// One feature of one node: workgroup of 256 threads, thread = slot.
__device__ __forceinline__ void scan_body(
    const int root_mode, const int small_slot, const int big_slot, const int parent_slot,
    const int small_is_left, const u64 minls, const int lf,
    const QrBlock *__restrict__ blocks, const int nblocks,
    const long long *__restrict__ red_sum, const uint32_t *__restrict__ red_cnt,
    long long *__restrict__ hsum, uint32_t *__restrict__ hcnt, const int flocal,
    const uint32_t *__restrict__ thr_size, const int32_t *__restrict__ lf2gf,
    const QrScalars *__restrict__ scal, qr_split_t *__restrict__ featrec, const uint32_t cs,
    const uint32_t *__restrict__ red_cnt_loc, uint32_t *__restrict__ hcnt_loc,
    const float *__restrict__ thr = nullptr, float *__restrict__ featthr = nullptr) {
  // the slot's threshold value travels with the record (requested now, used at the end)
  const float my_thr = featthr ? thr[(size_t)lf2gf[lf] * QR_MAX_BINS + threadIdx.x] : 0.f;
  int b = 0;
  uint32_t base = 0, mybase = 0;
  for (int i = 0; i < nblocks; ++i) {
    if (lf >= blocks[i].lf0 && lf < blocks[i].lf0 + blocks[i].nreal) {
      b = i;
      mybase = base;
    }
    base += 256u * blocks[i].fw;
  }
  const int col = lf - blocks[b].lf0;
  const int fw = blocks[b].fw;
  const uint32_t t = threadIdx.x;
  const long long s = red_sum[mybase + t * fw + col];
  const uint32_t cn = red_cnt[(size_t)(mybase + t * fw + col) * cs];
  // document-sharded: the same prefix over THIS rank's counts (kept aside by
  // k_reduce before the all-reduce) tells the partition how many of the rank's own
  // documents go left at any (feature, slot) -- no counting pass
  const uint32_t cl = hcnt_loc ? red_cnt_loc[mybase + t * fw + col] : 0u;
  scan_core(root_mode, small_slot, big_slot, parent_slot, small_is_left, minls, lf, s, cn, cl, hsum,
            hcnt, flocal, thr_size, lf2gf, scal, featrec, hcnt_loc, my_thr, featthr);
}

// thread t of 256 holds the reduced cell (s, cn[, cl]) of slot t of feature lf
__device__ __forceinline__ void scan_core(
    const int root_mode, const int small_slot, const int big_slot, const int parent_slot,
    const int small_is_left, const u64 minls, const int lf, long long s, uint32_t cn, uint32_t cl,
    long long *__restrict__ hsum, uint32_t *__restrict__ hcnt, const int flocal,
    const uint32_t *__restrict__ thr_size, const int32_t *__restrict__ lf2gf,
    const QrScalars *__restrict__ scal, qr_split_t *__restrict__ featrec,
    uint32_t *__restrict__ hcnt_loc, const float my_thr, float *__restrict__ featthr,
    const bool pre, const long long pre_s, const uint32_t pre_c, const int pre_gf,
    const uint32_t pre_tsize, const double pre_inv) {
  __shared__ long long sh_s[4];
  __shared__ uint32_t sh_c[4], sh_l[4];
  __shared__ long long tot_s[2];
  __shared__ uint32_t tot_c[2];
  __shared__ Best sh_b[4];
  const uint32_t t = threadIdx.x;
  // inclusive scan over the 256 slots (exact integers: any association)
  const int lane = t & 63, wave = t >> 6;
  s = wave_scan_i64(s);
  cn = wave_scan_u32(cn);
  if (hcnt_loc) cl = wave_scan_u32(cl);
  if (lane == 63) {
    sh_s[wave] = s;
    sh_c[wave] = cn;
    sh_l[wave] = cl;
  }
  __syncthreads();
  for (int w = 0; w < wave; ++w) {
    s += sh_s[w];
    cn += sh_c[w];
    cl += sh_l[w];
  }
  const size_t hidx = ((size_t)small_slot * flocal + lf) * 256 + t;
  hsum[hidx] = s;
  hcnt[hidx] = cn;
  if (hcnt_loc) hcnt_loc[hidx] = cl;
  long long bs = 0;
  uint32_t bc = 0;
  if (!root_mode) {
    const size_t pidx = ((size_t)parent_slot * flocal + lf) * 256 + t;
    bs = (pre ? pre_s : hsum[pidx]) - s;  // (pre: the caller fetched the parent's cell early)
    bc = (pre ? pre_c : hcnt[pidx]) - cn;
    const size_t bidx = ((size_t)big_slot * flocal + lf) * 256 + t;
    hsum[bidx] = bs;
    hcnt[bidx] = bc;
    if (hcnt_loc) hcnt_loc[bidx] = hcnt_loc[pidx] - cl;
  }
  if (t == 255) {
    tot_s[0] = s;
    tot_c[0] = cn;
    tot_s[1] = bs;
    tot_c[1] = bc;
  }
  __syncthreads();
  // (pre_gf >= 0: the caller requested these at its top -- here, behind the barriers, the
  // chain lf2gf -> thr_size would be two exposed round trips)
  const int gf = pre_gf >= 0 ? pre_gf : lf2gf[lf];
  const uint32_t tsize = pre_gf >= 0 ? pre_tsize : thr_size[gf];
  const double inv_scale = pre_gf >= 0 ? pre_inv : scal->inv_scale;
  {
    Best v = slot_gain(s, cn, tot_s[0], tot_c[0], t, tsize, minls, inv_scale);
    v = block_best(v, sh_b);
    const int which = root_mode ? 0 : (small_is_left ? 0 : 1);
    qr_split_t *o = &featrec[(size_t)which * flocal + lf];
    if (t == 0) {
      o->score = v.score;
      o->feature = v.t == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)gf;
      o->thr_id = v.t;
    }
    if (t == (v.t == 0xFFFFFFFFu ? 0u : v.t)) {  // the winning slot knows its counts
      o->lcount = v.t == 0xFFFFFFFFu ? 0 : cn;
      o->rcount = v.t == 0xFFFFFFFFu ? 0 : tot_c[0] - cn;
      if (featthr) featthr[(size_t)which * flocal + lf] = my_thr;
    }
  }
  if (!root_mode) {
    Best v = slot_gain(bs, bc, tot_s[1], tot_c[1], t, tsize, minls, inv_scale);
    v = block_best(v, sh_b);
    const int which = small_is_left ? 1 : 0;
    qr_split_t *o = &featrec[(size_t)which * flocal + lf];
    if (t == 0) {
      o->score = v.score;
      o->feature = v.t == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)gf;
      o->thr_id = v.t;
    }
    if (t == (v.t == 0xFFFFFFFFu ? 0u : v.t)) {
      o->lcount = v.t == 0xFFFFFFFFu ? 0 : bc;
      o->rcount = v.t == 0xFFFFFFFFu ? 0 : tot_c[1] - bc;
      if (featthr) featthr[(size_t)which * flocal + lf] = my_thr;
    }
  }
}

__global__ __launch_bounds__(256) void k_scan(
    const QrTreeState *__restrict__ ts, const int root_mode, const uint32_t N,
    const QrBlock *__restrict__ blocks, const int nblocks,
    const long long *__restrict__ red_sum, const uint32_t *__restrict__ red_cnt,
    long long *__restrict__ hsum, uint32_t *__restrict__ hcnt, const int flocal,
    const uint32_t *__restrict__ thr_size, const int32_t *__restrict__ lf2gf,
    const QrScalars *__restrict__ scal, qr_split_t *__restrict__ featrec, const uint32_t cs,
    const uint32_t *__restrict__ red_cnt_loc, uint32_t *__restrict__ hcnt_loc,
    const float *__restrict__ thr, float *__restrict__ featthr) {
  QR_JITTER();
  int small_slot, big_slot = -1, parent_slot = -1, small_is_left = 1;
  if (root_mode) {
    small_slot = 0;
  } else {
    if (!ts->desc.active) return;
    small_slot = ts->desc.small_slot;
    big_slot = ts->desc.big_slot;
    parent_slot = ts->desc.parent_slot;
    small_is_left = ts->desc.small_is_left;
  }
  scan_body(root_mode, small_slot, big_slot, parent_slot, small_is_left, ts->minls, blockIdx.x,
            blocks, nblocks, red_sum, red_cnt, hsum, hcnt, flocal, thr_size, lf2gf, scal, featrec,
            cs, red_cnt_loc, hcnt_loc, thr, featthr);
}

// Sum of one cell over the partial slots idx = g, g + 4, ... < total of a node of n
// documents (feature-major slots, `src` points at the cell in slot 0).  Eight slots
// per round: all eight requests leave before the first cell is used (unconditional
// loads of clamped slots, so that no branch separates them).
__device__ __forceinline__ void column_sum(const u64 *__restrict__ src, const int total,
                                           const int kmax, const uint32_t per, const uint32_t n,
                                           const int g, long long &s, uint32_t &cn) {
  for (int base = g; base < total; base += 32) {
    u64 cell[8];
    bool ok[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
      const int idx = base + 4 * k8;
      bool valid = idx < total;
      if (valid && kmax > 1) {  // workgroup j flushed only ceil(docs_j / QR_DPW) slots
        const int j = idx / kmax, k = idx - j * kmax;
        const uint32_t r0 = j * per;
        const uint32_t r1 = (r0 + per < n) ? r0 + per : n;
        valid = r0 < n && k < (int)((r1 - r0 + QR_DPW - 1) / QR_DPW);
      }
      ok[k8] = valid;
      cell[k8] = src[(size_t)(valid ? idx : 0) * (256u * 64u)];
    }
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
      if (ok[k8]) {
        const u64 cnt = (cell[k8] + (1ull << (QR_SB - 1))) >> QR_SB;
        s += (long long)(cell[k8] - (cnt << QR_SB));
        cn += (uint32_t)cnt;
      }
    }
  }
}

// level-wise growth: the reset of the tree state (k_obl_reset) as one more workgroup of the
// root scan launch, which does not look at the tree state
struct QrResetJob {
  QrTreeState *ts;
  u64 minls;
  int32_t maxnodes;  // 0: no reset workgroup in this launch
  int32_t pad;
};
__device__ __forceinline__ void obl_reset_body(QrTreeState *ts, const int i, const int maxnodes, const u64 minls) {
  if (i < maxnodes) {
    ts->nodes[i].feature = -2;  // absent
    ts->nodes[i].left = ts->nodes[i].right = -1;
    ts->nodes[i].count = 0;
    ts->nodes[i].value = 0.0;
    ts->nodes[i].deviance = 0.0;
    ts->nodes[i].threshold = 0.f;
    ts->nodes[i].thr_id = -1;
  }
  if (i == 0) {
    ts->nleaves_req = 0;
    ts->nnodes = 0;
    ts->taken = 0;
    ts->done = 0;
    ts->step = 0;
    ts->nsplits = 0;
    ts->minls = minls;
    ts->heap_size = 0;
    ts->desc.active = 0;
    ts->nleaves = 0;
    ts->obl_done = 0;
    ts->obl_level = 0;
    ts->incomplete = 0;
    ts->real_steps = 0;
  }
}

// Batched leaf-wise growth: k_reduce and k_scan in one launch.  Grid = (features,
// nodes of the batch), 1024 threads = 4 slot groups x 256 bins: the partial slots are
// feature-major here (hist_run, tr), so a workgroup reads its column of every slot
// as contiguous 2 KB rows; the four groups' sums meet in LDS and the first 256
// threads carry on with the scan.  Node y's records go to featrec[2y] (left child)
// and featrec[2y + 1] (right child).  `root`: the root node (slot 0, n = rootn).
__global__ __launch_bounds__(1024) void k_redscan(
    const QrTreeState *__restrict__ ts, const int root, const uint32_t rootn,
    const QrPlan *__restrict__ plans, const QrBlock *__restrict__ blocks, const int nblocks,
    const int G, const u64 *__restrict__ partials, long long *__restrict__ hsum,
    uint32_t *__restrict__ hcnt, const int flocal, const uint32_t *__restrict__ thr_size,
    const int32_t *__restrict__ lf2gf, const QrScalars *__restrict__ scal,
    qr_split_t *__restrict__ featrec, const float *__restrict__ thr,
    float *__restrict__ featthr, const QrScanWg *__restrict__ descs, const u64 minls,
    const double *__restrict__ part_ss, double *__restrict__ jobsum, const QrPrepJob prep,
    const QrResetJob reset) {
  QR_JITTER();
  __shared__ long long cs_s[3][256];
  __shared__ uint32_t cs_c[3][256];
  __shared__ QrPlan sh_plan;
  // (level-wise growth, root launch: the LAST workgroup resets the tree state -- up to 1023
  // node records, a thread each)
  if (reset.maxnodes && blockIdx.x == gridDim.x - 1) {
    obl_reset_body(reset.ts, (int)threadIdx.x, reset.maxnodes, reset.minls);
    return;
  }
  // the ROOT launch of an iteration whose scalars are not finished yet (prep.nwg > 0,
  // qr_prep.h): prep.nwg more workgroups ride behind the features' -- one wave each, what k_prep
  // does in a launch of its own -- and the features' workgroups take the scale from the slots
  if (prep.nwg && (int)blockIdx.x >= flocal) {
    if (threadIdx.x >= 64) return;
    prep_body<8>(prep, blockIdx.x - (uint32_t)flocal, (uint32_t)prep.nwg);
    return;
  }
  const u64 slot_word = prep.nwg ? prep.slots[threadIdx.x & (QR_PREP_SLOTS - 1)] : 0ull;
  const int lf = blockIdx.x;
  const uint32_t t = threadIdx.x & 255, g = threadIdx.x >> 8;
  // a node of the batch: everything comes ready-made from the control kernel (ONE
  // dependent read before the partials can be requested)
#ifdef QR_STEP_TIMING
  const long long rt0 = clock64();
#endif
  // everything that does not depend on the descriptor is requested with it: the feature's
  // global index, its number of thresholds, the slot's threshold value, the scale
  const int gf_top = lf2gf[lf];
  double inv_top = prep.nwg ? 0.0 : scal->inv_scale;
  QrScanWg d;
  // (root with `descs`: the feature's share of the root's plan ready-made, root_shares)
  if (!root || descs) d = descs[(size_t)(root ? 0 : blockIdx.y) * flocal + lf];
  const uint32_t tsize_top = thr_size[gf_top];
  const float my_thr = thr[(size_t)gf_top * QR_MAX_BINS + t];
  if (!root && !d.active) return;
  if (root && !descs) {
    if (threadIdx.x == 0)
      qr_make_plan(rootn, nblocks, blocks,
                   qr_plan_quantum((unsigned long long)rootn * qr_plan_wsum(nblocks, blocks), G - nblocks),
                   &sh_plan);
    __syncthreads();
    int b = 0;
    for (int i = 0; i < nblocks; ++i)
      if (lf >= blocks[i].lf0 && lf < blocks[i].lf0 + blocks[i].nreal) b = i;
    d.slot0 = (uint32_t)(sh_plan.wg_start[b] * sh_plan.kmax);
    d.total = (uint32_t)((sh_plan.wg_start[b + 1] - sh_plan.wg_start[b]) * sh_plan.kmax);
    d.per = sh_plan.per[b];
    d.n = rootn;
    d.kmax = sh_plan.kmax;
    d.small_slot = 0;
    d.big_slot = d.parent_slot = -1;
    d.small_is_left = 1;
    d.col = (uint32_t)(lf - blocks[b].lf0);
  }
  const int small_slot = d.small_slot, big_slot = d.big_slot, parent_slot = d.parent_slot;
  const int small_is_left = d.small_is_left;
  const uint32_t n = d.n, per = d.per;
  const int kmax = d.kmax;
  const int total = (int)d.total;
  // the parent's cumulative cell, for the sibling (requested with the partials)
  long long par_s = 0;
  uint32_t par_c = 0;
  if (!root && g == 0) {
    const size_t pidx = ((size_t)parent_slot * flocal + lf) * 256 + t;
    par_s = hsum[pidx];
    par_c = hcnt[pidx];
  }
  const u64 *src = partials + (size_t)d.slot0 * (256u * 64u) + d.col * 256u + t;
  long long s = 0;
  uint32_t cn = 0;
  // (feature 0's workgroup, last wave) sum and sum of squares of the node's directly
  // built child: fixed-order reduction of the pairs the histogram workgroups of feature
  // block 0 left at their first partial slot (hist_run, sums_out), for the next control
  // step, requested together with the column's cells
  const bool sums_wave = !root && lf == 0 && (threadIdx.x >> 6) == 15;
  double pa = 0.0, pb = 0.0;
  if (sums_wave) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nw0 = d.total / (uint32_t)d.kmax;   // block 0's workgroups of this node
    // (a lane's first four pairs are requested together -- up to 256 workgroups -- instead of a
    // loop of load, add, load, add; measured: 0.4152 against 0.4150 ms per iteration, nothing --
    // the wave's column sums behind it hide the second round trip)
    double2 x[4] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
    if (nw0 > 0) {  // (wave-uniform; an empty child has no slot to read)
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = lane + 64 * k;
        x[k] = *reinterpret_cast<const double2 *>(part_ss + 2 * (size_t)(d.slot0 + (i < nw0 ? i : 0u) * (uint32_t)d.kmax));
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
      if (lane + 64 * k < nw0) {
        pa += x[k].x;
        pb += x[k].y;
      }
    for (uint32_t i = lane + 256; i < nw0; i += 64) {
      pa += part_ss[2 * (size_t)(d.slot0 + i * (uint32_t)d.kmax)];
      pb += part_ss[2 * (size_t)(d.slot0 + i * (uint32_t)d.kmax) + 1];
    }
  }
#ifdef QR_STEP_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  const long long rt1 = clock64();
#endif
  column_sum(src, total, kmax, per, n, (int)g, s, cn);
#ifdef QR_STEP_TIMING
  const long long rt2 = clock64();
#endif
  if (sums_wave) {
    pa = wave_sum(pa);
    pb = wave_sum(pb);
    if ((threadIdx.x & 63) == 0) {
      jobsum[2 * blockIdx.y] = pa;
      jobsum[2 * blockIdx.y + 1] = pb;
    }
  }
  if (g > 0) {
    cs_s[g - 1][t] = s;
    cs_c[g - 1][t] = cn;
  }
  __syncthreads();
  if (g > 0) return;  // whole waves leave; the barriers below count the remaining four
  for (int i = 0; i < 3; ++i) {
    s += cs_s[i][t];
    cn += cs_c[i][t];
  }
  if (prep.nwg) inv_top = ldexp(1.0, -qr_slot_scale_exp(slot_word));
  scan_core(root, small_slot, big_slot, parent_slot, small_is_left, minls, lf, s, cn, 0u, hsum,
            hcnt, flocal, thr_size, lf2gf, scal, featrec + (size_t)2 * (root ? 0 : blockIdx.y) * flocal,
            nullptr, my_thr, featthr + (size_t)2 * (root ? 0 : blockIdx.y) * flocal, !root, par_s,
            par_c, gf_top, tsize_top, inv_top);
#ifdef QR_STEP_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)
    printf("redscan wg0 (root %d, %d slots): desc %lld partials %lld scan %lld cycles\n", root, total, rt1 - rt0,
           rt2 - rt1, clock64() - rt2);
#endif
}

// Level-wise (oblivious) growth: sum of the workgroup partials, prefix of the directly
// built child and sibling by subtraction for every node of the level, in one launch over
// feature-major partials (see k_redscan): grid = (features, nodes of the level), 4 slot
// groups x 256 bins; the gains are summed over the level by k_obl_fill, not here
// (the first 256 threads) cumulative cells of a directly built child from its per-slot
// values (s, cn), and the sibling by subtraction from the parent's (par_s, par_c)
__device__ __forceinline__ void level_prefix_write(long long s, uint32_t cn, const long long par_s,
                                                   const uint32_t par_c, const QrLevelNode &ln,
                                                   const int lf, const uint32_t t, const int flocal,
                                                   long long *__restrict__ hsum, uint32_t *__restrict__ hcnt,
                                                   long long *sh_s, uint32_t *sh_c) {
  s = wave_scan_i64(s);
  cn = wave_scan_u32(cn);
  const int lane = t & 63, wave = t >> 6;
  if (lane == 63) {
    sh_s[wave] = s;
    sh_c[wave] = cn;
  }
  __syncthreads();
  for (int w = 0; w < wave; ++w) {
    s += sh_s[w];
    cn += sh_c[w];
  }
  const size_t hidx = ((size_t)ln.small_slot * flocal + lf) * 256 + t;
  const size_t bidx = ((size_t)ln.big_slot * flocal + lf) * 256 + t;
#ifdef QR_DEBUG_CHECKS
  if (!QR_DBG_OK((u64)ln.small_slot < qr_dbg_caps[4] && (u64)ln.big_slot < qr_dbg_caps[4], 3, ln.small_slot, ln.big_slot)) return;
#endif
  if (hsum) {
    hsum[hidx] = s;
    hsum[bidx] = par_s - s;
  }
  hcnt[hidx] = cn;
  hcnt[bidx] = par_c - cn;
}

// xl != null: a document-sharded rank.  The per-slot sums of the rank's own documents go
// to the level's exchange buffer [node][feature][slot][sum, count] (int64: the all-reduce
// adds them up, k_level_finish_doc takes the prefix), and the rank's own cumulative COUNTS
// -- where its partition of the next level cuts its lists -- to hcnt_loc.
__global__ __launch_bounds__(1024) void k_redscan_level(
    const QrTreeState *__restrict__ ts, const QrBlock *__restrict__ blocks, const int nblocks,
    const u64 *__restrict__ partials, long long *__restrict__ hsum, uint32_t *__restrict__ hcnt,
    const int flocal, long long *__restrict__ xl, uint32_t *__restrict__ hcnt_loc) {
  QR_JITTER();
  __shared__ long long cs_s[3][256];
  __shared__ uint32_t cs_c[3][256];
  __shared__ long long sh_s[4];
  __shared__ uint32_t sh_c[4];
  __shared__ QrPlan plan;
  if (ts->obl_done || (int)blockIdx.y >= ts->l_nodes) return;
  const QrLevelNode &ln = ts->lnode[blockIdx.y];
  if (!ln.active) return;
  const int lf = blockIdx.x;
  const uint32_t t = threadIdx.x & 255, g = threadIdx.x >> 8;
  if (threadIdx.x == 0) qr_make_plan(ln.small_n, nblocks, blocks, ln.q, &plan);
  const size_t pidx = ((size_t)ln.parent_slot * flocal + lf) * 256 + t;
  long long par_s = 0;
  uint32_t par_c = 0;
  if (g == 0) {
    par_s = xl ? 0 : hsum[pidx];
    par_c = xl ? hcnt_loc[pidx] : hcnt[pidx];
  }
  __syncthreads();
  int b = 0;
  for (int i = 0; i < nblocks; ++i)
    if (lf >= blocks[i].lf0 && lf < blocks[i].lf0 + blocks[i].nreal) b = i;
  const uint32_t col = (uint32_t)(lf - blocks[b].lf0);
  const int kmax = plan.kmax;
  const int total = (plan.wg_start[b + 1] - plan.wg_start[b]) * kmax;
  const u64 *src = partials + ((size_t)ln.slot_base + (size_t)plan.wg_start[b] * kmax) * (256u * 64u) +
                   col * 256u + t;
  long long s = 0;
  uint32_t cn = 0;
  column_sum(src, total, kmax, plan.per[b], ln.small_n, (int)g, s, cn);
  if (g > 0) {
    cs_s[g - 1][t] = s;
    cs_c[g - 1][t] = cn;
  }
  __syncthreads();
  if (g > 0) return;  // whole waves leave; the barrier below counts the remaining four
  for (int i = 0; i < 3; ++i) {
    s += cs_s[i][t];
    cn += cs_c[i][t];
  }
  if (xl) {
    const size_t x = (((size_t)blockIdx.y * flocal + lf) * 256 + t) * 2;
    xl[x] = s;
    xl[x + 1] = (long long)cn;
    level_prefix_write(0, cn, 0, par_c, ln, lf, t, flocal, nullptr, hcnt_loc, sh_s, sh_c);
    return;
  }
  level_prefix_write(s, cn, par_s, par_c, ln, lf, t, flocal, hsum, hcnt, sh_s, sh_c);
}

// document-sharded level-wise growth, after the all-reduce of the level's exchange buffer:
// cumulative cells of every directly built child over ALL ranks' documents, siblings by
// subtraction.  Grid = (features, nodes of the level), 256 threads.
__global__ __launch_bounds__(256) void k_level_finish_doc(
    const QrTreeState *__restrict__ ts, const long long *__restrict__ xl,
    long long *__restrict__ hsum, uint32_t *__restrict__ hcnt, const int flocal) {
  QR_JITTER();
  __shared__ long long sh_s[4];
  __shared__ uint32_t sh_c[4];
  if (ts->obl_done || (int)blockIdx.y >= ts->l_nodes) return;
  const QrLevelNode &ln = ts->lnode[blockIdx.y];
  if (!ln.active) return;
  const int lf = blockIdx.x;
  const uint32_t t = threadIdx.x;
  const size_t pidx = ((size_t)ln.parent_slot * flocal + lf) * 256 + t;
  const size_t x = (((size_t)blockIdx.y * flocal + lf) * 256 + t) * 2;
  level_prefix_write(xl[x], (uint32_t)xl[x + 1], hsum[pidx], hcnt[pidx], ln, lf, t, flocal, hsum, hcnt,
                     sh_s, sh_c);
}

// ===========================================================================
// Batched leaf-wise growth on DOCUMENT-SHARDED ranks: k_redscan cut in two around the
// all-reduce.  The batch's exchange buffer `xb` holds, per job j, the per-slot cells of its
// directly built child over this rank's documents -- [j][feature][slot][sum, count], int64 --
// and, behind all jobs' cells, [j][2 * world] f64 bit patterns: the rank's (sum of squares,
// sum) of that child in its own pair, zeros elsewhere, so that the sum all-reduce gathers them
// (as k_reduce does for the one-split protocol).
//   k_bd_reduce (before the all-reduce): column sums of the feature-major partial slots into
//     xb, and the rank's own CUMULATIVE counts of both children into hcnt_loc -- where its
//     partition of a later split cuts its lists.
//   k_bd_scan (after): prefix over the slots of the summed cells, sibling by subtraction,
//     best slot per feature for both children (scan_core: the one-GPU arithmetic on the
//     one-GPU integers), and the children's sums in rank order into jobsum.
// Grid = (features, QR_BATCH) for both.
// ===========================================================================
__global__ __launch_bounds__(1024) void k_bd_reduce(
    const QrScanWg *__restrict__ descs, const u64 *__restrict__ partials, const int flocal,
    long long *__restrict__ xb, uint32_t *__restrict__ hcnt_loc, const double *__restrict__ part_ss,
    const int rank, const int world) {
  QR_JITTER();
  __shared__ long long cs_s[3][256];
  __shared__ uint32_t cs_c[3][256];
  __shared__ long long sh_s[4];
  __shared__ uint32_t sh_c[4];
  const int lf = blockIdx.x;
  const uint32_t t = threadIdx.x & 255, g = threadIdx.x >> 8;
  const QrScanWg d = descs[(size_t)blockIdx.y * flocal + lf];
  if (!d.active) {
    // k_bd_scan does not look at an inactive job's cells, but the host all-reduces the whole
    // buffer: zeros instead of whatever an earlier step left (stale payload multiplied
    // by `world` at every exchange -- harmless wrap-around in int64, signed overflow on a CPU
    // transport)
    if (g == 0) {
      const size_t x = (((size_t)blockIdx.y * flocal + lf) * 256 + t) * 2;
      xb[x] = 0;
      xb[x + 1] = 0;
    }
    if (lf == 0 && threadIdx.x < (uint32_t)(2 * world))
      xb[(size_t)QR_BATCH * flocal * 512 + (size_t)blockIdx.y * 2 * world + threadIdx.x] = 0;
    return;
  }
  uint32_t par_c = 0;
  if (g == 0) par_c = hcnt_loc[((size_t)d.parent_slot * flocal + lf) * 256 + t];
  const u64 *src = partials + (size_t)d.slot0 * (256u * 64u) + d.col * 256u + t;
  long long s = 0;
  uint32_t cn = 0;
  // (feature 0's workgroup, last wave) the rank's sums of the directly built child, as k_redscan
  const bool sums_wave = lf == 0 && (threadIdx.x >> 6) == 15;
  double pa = 0.0, pb = 0.0;
  if (sums_wave) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nw0 = d.total / (uint32_t)d.kmax;
    for (uint32_t i = lane; i < nw0; i += 64) {
      pa += part_ss[2 * (size_t)(d.slot0 + i * (uint32_t)d.kmax)];
      pb += part_ss[2 * (size_t)(d.slot0 + i * (uint32_t)d.kmax) + 1];
    }
  }
  column_sum(src, (int)d.total, d.kmax, d.per, d.n, (int)g, s, cn);
  if (sums_wave) {
    pa = wave_sum(pa);
    pb = wave_sum(pb);
    long long *tail = xb + (size_t)QR_BATCH * flocal * 512 + (size_t)blockIdx.y * 2 * world;
    for (int i = threadIdx.x & 63; i < 2 * world; i += 64) {
      long long v = 0;
      if (i == 2 * rank) v = __double_as_longlong(pa);
      if (i == 2 * rank + 1) v = __double_as_longlong(pb);
      tail[i] = v;
    }
  }
  if (g > 0) {
    cs_s[g - 1][t] = s;
    cs_c[g - 1][t] = cn;
  }
  __syncthreads();
  if (g > 0) return;  // whole waves leave; the barrier below counts the remaining four
  for (int i = 0; i < 3; ++i) {
    s += cs_s[i][t];
    cn += cs_c[i][t];
  }
  const size_t x = (((size_t)blockIdx.y * flocal + lf) * 256 + t) * 2;
  xb[x] = s;
  xb[x + 1] = (long long)cn;
  QrLevelNode ln;
  ln.small_slot = d.small_slot;
  ln.big_slot = d.big_slot;
  level_prefix_write(0, cn, 0, par_c, ln, lf, t, flocal, nullptr, hcnt_loc, sh_s, sh_c);
}

__global__ __launch_bounds__(256) void k_bd_scan(
    const QrScanWg *__restrict__ descs, const long long *__restrict__ xb, long long *__restrict__ hsum,
    uint32_t *__restrict__ hcnt, const int flocal, const uint32_t *__restrict__ thr_size,
    const int32_t *__restrict__ lf2gf, const QrScalars *__restrict__ scal,
    qr_split_t *__restrict__ featrec, const float *__restrict__ thr, float *__restrict__ featthr,
    const u64 minls, double *__restrict__ jobsum, const int world) {
  QR_JITTER();
  const int lf = blockIdx.x;
  const uint32_t t = threadIdx.x;
  const int gf = lf2gf[lf];
  const QrScanWg d = descs[(size_t)blockIdx.y * flocal + lf];
  const uint32_t tsize = thr_size[gf];
  const float my_thr = thr[(size_t)gf * QR_MAX_BINS + t];
  const double inv = scal->inv_scale;
  if (!d.active) return;
  const size_t x = (((size_t)blockIdx.y * flocal + lf) * 256 + t) * 2;
  const long long s = xb[x];
  const uint32_t cn = (uint32_t)xb[x + 1];
  const size_t pidx = ((size_t)d.parent_slot * flocal + lf) * 256 + t;
  const long long par_s = hsum[pidx];
  const uint32_t par_c = hcnt[pidx];
  if (lf == 0 && t == 0) {
    // the ranks' partial sums in rank order: every rank computes the same bits
    const long long *tail = xb + (size_t)QR_BATCH * flocal * 512 + (size_t)blockIdx.y * 2 * world;
    double a = 0.0, b = 0.0;
    for (int r = 0; r < world; ++r) {
      a += __longlong_as_double(tail[2 * r]);
      b += __longlong_as_double(tail[2 * r + 1]);
    }
    jobsum[2 * blockIdx.y] = a;
    jobsum[2 * blockIdx.y + 1] = b;
  }
  scan_core(0, d.small_slot, d.big_slot, d.parent_slot, d.small_is_left, minls, lf, s, cn, 0u, hsum, hcnt,
            flocal, thr_size, lf2gf, scal, featrec + (size_t)2 * blockIdx.y * flocal, nullptr, my_thr,
            featthr + (size_t)2 * blockIdx.y * flocal, true, par_s, par_c, gf, tsize, inv);
}

// ===========================================================================
// k_merge: best over the local features; local features are in ascending
// global order and the comparison is strict, so the lowest feature wins ties
// (rt.cc:297-306).  Also attaches lcount/rcount from the node histogram.
// ===========================================================================
// merge by one wave; every lane returns the same record
// --max-features (rt.cc:222-243): a node's split search sees only a random subset
// of mf_k features.  The reference shuffles the feature ids with a clock-seeded
// engine at every split() call; here the subset of node `node` is the mf_k features
// with the smallest keys hash(seed, node, f), so that a run is reproducible, every
// rank of a multi-GPU run draws the same subset, and no state is carried around.
__device__ __forceinline__ u64 mf_key(u64 seed, uint32_t node, uint32_t f) {
  u64 z = seed + 0x9E3779B97F4A7C15ull * ((u64)node * 0x100000001ull + f + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ bool mf_allowed(u64 seed, uint32_t node, uint32_t f, uint32_t F,
                                           uint32_t mf_k) {
  const u64 mine = mf_key(seed, node, f);
  uint32_t rank = 0;
  for (uint32_t g = 0; g < F; ++g) {
    const u64 k = mf_key(seed, node, g);
    rank += (k < mine || (k == mine && g < f)) ? 1u : 0u;
  }
  return rank < mf_k;
}

__device__ qr_split_t wave_merge(const int root_mode, const int which,
                                 const qr_split_t *featrec, const int flocal,
                                 const uint32_t mf_k, const u64 mf_seed, const uint32_t mf_node,
                                 const uint32_t F, int *lf_out = nullptr,
                                 const float *featthr = nullptr, float *thr_out = nullptr) {
  // (featthr: the threshold VALUE of every feature's record rides along, so that the
  // caller has the winner's without a second, dependent read)
  const int lane = threadIdx.x & 63;
  int best_lf = -1;
  float best_thr = 0.f;
  qr_split_t best;
  best.score = -1.0;
  best.feature = 0xFFFFFFFFu;
  best.thr_id = 0xFFFFFFFFu;
  best.lcount = best.rcount = 0;
  if (root_mode && which == 1) return best;
  // Four records per lane and round, ALL requested before the first is looked at (clamped,
  // unconditional loads): the records are fresh from the scan launch, so every round trip is
  // a cold miss of ~1 us, and a loop that waits for each record in turn put three of them
  // in a row on the per-step chain (F = 136: lanes 0..7 hold three records).
  for (int base = 0; base < flocal; base += 256) {
    qr_split_t r[4];
    float tv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lf = base + lane + 64 * k;
      const int lc = lf < flocal ? lf : flocal - 1;
      r[k] = featrec[(size_t)which * flocal + lc];
      tv[k] = featthr ? featthr[(size_t)which * flocal + lc] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lf = base + lane + 64 * k;
      if (lf >= flocal) continue;
      if (mf_k && r[k].feature != 0xFFFFFFFFu && !mf_allowed(mf_seed, mf_node, r[k].feature, F, mf_k))
        continue;
      if (r[k].score > best.score) {  // ascending lf within the lane
        best = r[k];
        best_lf = lf;
        best_thr = tv[k];
      }
    }
  }
  // max score over the wave, equal scores -> lowest feature index; the lane that
  // holds the winner hands out its slot and the counts k_scan recorded with it
  const double m = wave_max(best.score);
  const bool cand = best.score == m && best.feature != 0xFFFFFFFFu;
  const uint32_t fmin = wave_min_u32(cand ? best.feature : 0xFFFFFFFFu);
  const unsigned long long hit = __ballot(cand && best.feature == fmin);
  if (hit) {
    const int src = __ffsll((long long)hit) - 1;
    best.score = m;
    best.feature = fmin;
    best.thr_id = (uint32_t)__builtin_amdgcn_readlane((int)best.thr_id, src);
    best.lcount = (u64)readlane_i64((long long)best.lcount, src);
    best.rcount = (u64)readlane_i64((long long)best.rcount, src);
    if (lf_out) *lf_out = __builtin_amdgcn_readlane(best_lf, src);
    if (thr_out) *thr_out = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best_thr), src));
  } else {
    if (lf_out) *lf_out = -1;
    if (thr_out) *thr_out = 0.f;
    best.score = -1.0;
    best.feature = 0xFFFFFFFFu;
    best.thr_id = 0xFFFFFFFFu;
    best.lcount = best.rcount = 0;
  }
  return best;
}

// multi-GPU only: publish this rank's best records for the all_gather
__global__ __launch_bounds__(128) void k_merge(
    const QrTreeState *__restrict__ ts, const int root_mode,
    const qr_split_t *__restrict__ featrec, const int flocal,
    const long long *__restrict__ hsum, const uint32_t *__restrict__ hcnt,
    const int32_t *__restrict__ gf2lf, qr_split_t *__restrict__ recs_local,
    const uint32_t mf_k, const u64 mf_seed, const uint32_t F) {
  QR_JITTER();
  if (!root_mode && !ts->desc.active) return;
  const int which = threadIdx.x >> 6;  // wave 0: left child (or root), wave 1: right
  const uint32_t node = root_mode ? 0u : (uint32_t)(which ? ts->desc.right : ts->desc.left);
  const qr_split_t best = wave_merge(root_mode, which, featrec, flocal, mf_k, mf_seed, node, F);
  if ((threadIdx.x & 63) == 0) recs_local[which] = best;
}

// ===========================================================================
// k_decide: the host logic of RegressionTree::fit, on one lane.
// ===========================================================================
// The control state the lane works on: header fields in registers, heap / nodes /
// descriptor through pointers that lead either to LDS copies (small trees: every
// access is an LDS access instead of a dependent L2 round trip) or straight to the
// device-resident QrTreeState.
struct DecideState {
#ifdef QR_DEBUG_CHECKS
  int32_t cap = QR_MAXNODES;  // node records / heap entries `nodes` / `heap` hold (the LDS copies: fewer)
#endif
  int32_t nleaves_req, nnodes, taken, done, step, nsplits, heap_size;
  uint32_t part_epoch;
  QrHeapItem *heap;
  QrNode *nodes;
  QrSplitDesc *desc;
  qr_split_t *split_log;
  qr_split_t *split_log2;  // batched growth: second copy of the tree state (or null)
  // document-sharded only: this rank's cumulative counts and where to put the
  // local view of the split
  const uint32_t *hcnt_loc;
  QrLocalSplit *loc;
  int flocal;
  size_t loc_cells;  // wide bins: cells of a node's ragged count arrays (hcnt_loc[slot * cells + woff[f] + t]); 0: [f][256]
};

__device__ QR_CTRL_FN void heap_push(DecideState &st, double key, int32_t val) {
  // maxheap.h:58-68 (arr[0] is a DBL_MAX sentinel)
#ifdef QR_DEBUG_CHECKS
  if (!QR_DBG_OK(st.heap_size + 1 < st.cap + 2, 4, st.heap_size, st.cap)) return;
#endif
  size_t p = (size_t)(++st.heap_size);
  while (key > st.heap[p >> 1].key) {
    st.heap[p] = st.heap[p >> 1];
    p >>= 1;
  }
  st.heap[p].key = key;
  st.heap[p].val = val;
}

__device__ QR_CTRL_FN void heap_pop(DecideState &st) {
  // maxheap.h:71-84
  const QrHeapItem last = st.heap[st.heap_size--];
  size_t child, p = 1;
  const size_t size = (size_t)st.heap_size;
  while ((p << 1) <= size) {
    child = p << 1;
    if (child < size && st.heap[child + 1].key > st.heap[child].key) ++child;
    if (last.key < st.heap[child].key)
      st.heap[p] = st.heap[child];
    else
      break;
    p = child;
  }
  st.heap[p] = last;
}

// RTNode(sampleids, hist): rtnode.h:97-107
__device__ __forceinline__ void node_stats(QrNode *nd, double sum, double ss, u64 count) {
  // `sum` is the f64 sum of the node's pseudo-responses (not the fixed-point
  // histogram total): a one-document node must come out with deviance == 0
  // exactly, as in the reference, because `deviance > 0` gates the split.
  nd->count = count;
  nd->sum = sum;
  nd->ss = ss;
  nd->value = count ? nd->sum / (double)count : 0.0;
  nd->deviance = ss - nd->sum * nd->sum / (double)count;
}

__device__ __forceinline__ void node_set_best(QrNode *nd, const qr_split_t *recs, int world,
                              int which) {
  // deterministic merge over ranks: max score, ties -> lowest feature
  nd->best_score = -1.0;
  nd->best_f = 0xFFFFFFFFu;
  nd->best_t = 0xFFFFFFFFu;
  nd->best_lc = nd->best_rc = 0;
  for (int r = 0; r < world; ++r) {
    const qr_split_t x = recs[(size_t)r * 2 + which];
    if (x.feature == 0xFFFFFFFFu) continue;
    if (x.score > nd->best_score ||
        (x.score == nd->best_score && x.feature < nd->best_f)) {
      nd->best_score = x.score;
      nd->best_f = x.feature;
      nd->best_t = x.thr_id;
      nd->best_lc = x.lcount;
      nd->best_rc = x.rcount;
    }
  }
}

__device__ QR_CTRL_FN bool node_splittable(const QrNode *nd) {
  // rt.cc:212 (deviance > 0.0f) and :312 (best_score == initvar => unsplittable)
  return nd->deviance > 0.0 && nd->best_f != 0xFFFFFFFFu;
}

// (thr_off: wide-bin contexts keep ragged threshold rows, feature f at thr[thr_off[f]])
__device__ __forceinline__ void make_desc(DecideState &st, int node, const float *thr,
                          const int32_t *gf2lf, const uint32_t *thr_off = nullptr) {
  QrNode *nd = &st.nodes[node];
  QrSplitDesc *d = st.desc;
  const int li = st.nnodes, ri = st.nnodes + 1;
#ifdef QR_DEBUG_CHECKS
  if (!QR_DBG_OK(ri < st.cap && st.nsplits < QR_MAXNODES, 6, li, st.cap)) return;
#endif
  st.nnodes += 2;
  d->active = 1;
  st.part_epoch++;
  d->node = node;
  d->left = li;
  d->right = ri;
  d->begin = nd->begin;
  d->end = nd->end;
  d->src_buf = nd->buf;
  d->dst_buf = nd->buf == 0 ? 1 : 0;
  d->lcount = (uint32_t)nd->best_lc;
  d->rcount = (uint32_t)nd->best_rc;
  d->feature = nd->best_f;
  d->thr_id = nd->best_t;
  d->owner_local = gf2lf[nd->best_f];
  d->small_is_left = d->lcount <= d->rcount;
  d->small_node = d->small_is_left ? li : ri;
  d->big_node = d->small_is_left ? ri : li;
  d->parent_slot = nd->hslot;
  d->small_slot = d->small_node;
  d->big_slot = d->big_node;
  d->small_begin = d->small_is_left ? nd->begin : nd->begin + d->lcount;
  d->small_n = d->small_is_left ? d->lcount : d->rcount;
  nd->feature = (int32_t)nd->best_f;
  nd->thr_id = (int32_t)nd->best_t;
  // (wide contexts: ragged rows indexed by LOCAL feature; the threshold of a feature another
  // rank owns arrives behind the go-left mask, k_thr_patch)
  nd->threshold = thr_off ? (d->owner_local >= 0 ? thr[(size_t)thr_off[d->owner_local] + nd->best_t] : 0.f)
                          : thr[(size_t)nd->best_f * QR_MAX_BINS + nd->best_t];
  nd->left = li;
  nd->right = ri;
  QrNode *L = &st.nodes[li], *R = &st.nodes[ri];
  L->begin = nd->begin;
  L->end = nd->begin + d->lcount;
  if (st.hcnt_loc) {
    // [begin, end) are positions in the rank's own lists; the counts above are global
    const uint32_t ll = thr_off
        ? st.hcnt_loc[(size_t)nd->hslot * st.loc_cells + thr_off[d->owner_local] + nd->best_t]
        : st.hcnt_loc[((size_t)nd->hslot * st.flocal + d->owner_local) * 256 + nd->best_t];
    const uint32_t ln = nd->end - nd->begin;
    L->end = nd->begin + ll;
    st.loc->lcount = ll;
    st.loc->small_begin = d->small_is_left ? nd->begin : nd->begin + ll;
    st.loc->small_n = d->small_is_left ? ll : ln - ll;
  }
  R->begin = L->end;
  R->end = nd->end;
  L->buf = R->buf = d->dst_buf;
  L->hslot = li;
  R->hslot = ri;
  L->feature = R->feature = -1;
  L->thr_id = R->thr_id = -1;
  L->threshold = R->threshold = 0.f;
  L->left = L->right = R->left = R->right = -1;
  L->parent = R->parent = node;
  L->leaf_id = R->leaf_id = -1;
  qr_split_t *lg = &st.split_log[st.nsplits++];
  lg->score = nd->best_score;
  lg->feature = nd->best_f;
  lg->thr_id = nd->best_t;
  lg->lcount = nd->best_lc;
  lg->rcount = nd->best_rc;
}

// The lane's control step on the state `st` points at.  Force-inlined into two call
// sites (LDS-staged / device-resident) so that each gets the memory instructions
// of its address space.
__device__ __forceinline__ void decide_logic(DecideState &st, QrTreeState *ts, const bool root_mode,
                                             const int32_t active, const uint32_t N,
                                             const qr_split_t *recs, const int world,
                                             const QrScalars *scal, const float *thr,
                                             const int32_t *gf2lf, const int docmode, const u64 Nglobal,
                                             const double sum_small, const double ss_small,
                                             const int root_buf, const uint32_t *thr_off = nullptr) {
  if (root_mode) {
    QrNode *root = &st.nodes[0];
    root->begin = 0;
    root->end = N;
    root->buf = root_buf;
    root->hslot = 0;
    root->feature = -1;
    root->thr_id = -1;
    root->threshold = 0.f;
    root->left = root->right = root->parent = -1;
    root->leaf_id = -1;
    node_stats(root, scal->root_sum, scal->root_ss, docmode ? Nglobal : (u64)N);
    node_set_best(root, recs, world, 0);
    st.nnodes = 1;
    st.heap_size = 0;
    st.heap[0].key = 1.7976931348623157e308;  // DBL_MAX sentinel
    st.heap[0].val = -1;
    st.taken = 0;
    st.done = 0;
    st.nsplits = 0;
    st.desc->active = 0;
    if (node_splittable(root))
      make_desc(st, 0, thr, gf2lf, thr_off);
    else
      st.done = 1;
    st.step = 1;
  } else {
    if (active) {
      // children of the split just applied
      const QrSplitDesc d = *st.desc;
      QrNode *P = &st.nodes[d.node];
      QrNode *S = &st.nodes[d.small_node], *B = &st.nodes[d.big_node];
      // directly accumulated child, sibling by subtraction
      // (rtnode_histogram.cc:65-69, 79-86)
      node_stats(S, sum_small, ss_small, d.small_n);
      node_stats(B, P->sum - sum_small, P->ss - ss_small, P->count - d.small_n);
      node_set_best(&st.nodes[d.left], recs, world, 0);
      node_set_best(&st.nodes[d.right], recs, world, 1);
      heap_push(st, st.nodes[d.left].deviance, d.left);    // rt.cc:76-77
      heap_push(st, st.nodes[d.right].deviance, d.right);
      st.desc->active = 0;
    }
    st.step++;
    if (!st.done) {
      bool found = false;
      while (st.heap_size > 0 &&
             (st.nleaves_req == 0 || st.taken + st.heap_size < st.nleaves_req)) {
        const int node = st.heap[1].val;
        heap_pop(st);
        if (node_splittable(&st.nodes[node])) {
          make_desc(st, node, thr, gf2lf, thr_off);
          found = true;
          break;
        }
        ++st.taken;
      }
      if (!found) st.done = 1;
    }
  }
}

#define QR_DECIDE_LDS_NODES 96  /* trees of up to 47 leaves are staged in LDS */

__device__ __forceinline__ void wave_copy8(void *dst, const void *src, size_t bytes) {
  u64 *d = reinterpret_cast<u64 *>(dst);
  const u64 *s_ = reinterpret_cast<const u64 *>(src);
  for (size_t i = threadIdx.x; i < bytes / 8; i += 64) d[i] = s_[i];
}

// Runs on wave 0 of the calling workgroup; every thread must call it (barriers).
__device__ void decide_body(QrTreeState *ts, const uint32_t N, const int flocal,
                            const qr_split_t *recs, const int world, const QrScalars *scal,
                            const double *part_ss, const float *thr, const int32_t *gf2lf,
                            const qr_split_t *featrec, const uint32_t *hcnt_loc, const int docmode,
                            const u64 Nglobal, const long long *tail, const int dworld,
                            const uint32_t mf_k, const u64 mf_seed, const uint32_t F,
                            const int root_buf, const uint32_t *thr_off, const size_t loc_cells) {
  const bool w0 = threadIdx.x < 64;
  // single GPU (and document-sharded, where every rank scans every feature of
  // the all-reduced histogram): the merge over features happens here (no k_merge
  // launch, no exchange); feature-sharded: `recs` is the all-gathered buffer
  __shared__ qr_split_t own[2];
  __shared__ QrNode sh_nodes[QR_DECIDE_LDS_NODES];
  __shared__ QrHeapItem sh_heap[QR_DECIDE_LDS_NODES + 2];
  __shared__ QrSplitDesc sh_desc;
  static_assert(sizeof(QrNode) % 8 == 0 && sizeof(QrHeapItem) % 8 == 0 &&
                    sizeof(QrSplitDesc) % 8 == 0,
                "wave_copy8 moves 8-byte words");
  DecideState st;
  st.nleaves_req = ts->nleaves_req;
  st.nnodes = ts->nnodes;
  st.taken = ts->taken;
  st.done = ts->done;
  st.step = ts->step;
  st.nsplits = ts->nsplits;
  st.heap_size = ts->heap_size;
  st.part_epoch = ts->part_epoch;
  st.split_log = ts->split_log;
  st.split_log2 = nullptr;
  st.hcnt_loc = docmode ? hcnt_loc : nullptr;
  st.loc = &ts->loc;
  st.flocal = flocal;
  st.loc_cells = loc_cells;
  const int32_t active = ts->desc.active;
  const int root_mode = st.step == 0;
  // everything this step can touch: the live nodes + 2 new ones, the heap + 2
  const bool staged = !root_mode && st.nnodes + 2 <= QR_DECIDE_LDS_NODES &&
                      st.heap_size + 3 <= QR_DECIDE_LDS_NODES + 2;
  if (staged && w0) {
    wave_copy8(sh_nodes, ts->nodes, (size_t)st.nnodes * sizeof(QrNode));
    wave_copy8(sh_heap, ts->heap, (size_t)(st.heap_size + 1) * sizeof(QrHeapItem));
    wave_copy8(&sh_desc, &ts->desc, sizeof(QrSplitDesc));
  }
  if (staged) {
    st.nodes = sh_nodes;
#ifdef QR_DEBUG_CHECKS
    st.cap = QR_DECIDE_LDS_NODES;
#endif
    st.heap = sh_heap;
    st.desc = &sh_desc;
  } else {
    st.nodes = ts->nodes;
    st.heap = ts->heap;
    st.desc = &ts->desc;
  }
  if (world == 1) {
    // wave 0 merges the left child's (or the root's) records, wave 1 the right child's
    if (threadIdx.x < 128 && (root_mode || active)) {
      const int which = threadIdx.x >> 6;
      const uint32_t node = root_mode ? 0u : (uint32_t)(which ? ts->desc.right : ts->desc.left);
      const qr_split_t a = wave_merge(root_mode, which, featrec, flocal, mf_k, mf_seed, node, F);
      if ((threadIdx.x & 63) == 0) own[which] = a;
    }
    recs = own;
  }
  // squares_sum_ / sum of the directly built child: fixed-order reduction of the
  // partition workgroups' partials by the whole wave
  double ss_small = 0.0, sum_small = 0.0;
  if (docmode) {
    // per-rank partials gathered by the histogram all-reduce, summed in rank
    // order: every rank computes the same bits
    if (!root_mode && active)
      for (int r = 0; r < dworld; ++r) {
        ss_small += __longlong_as_double(tail[2 * r]);
        sum_small += __longlong_as_double(tail[2 * r + 1]);
      }
  } else if (w0 && !root_mode && active) {
    const uint32_t nwg = (ts->desc.end - ts->desc.begin + QR_PART_SLICE - 1) / QR_PART_SLICE;
    for (uint32_t i = threadIdx.x; i < nwg; i += 64) {
      ss_small += part_ss[2 * i];
      sum_small += part_ss[2 * i + 1];
    }
    ss_small = wave_sum(ss_small);
    sum_small = wave_sum(sum_small);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (staged) {
      st.nodes = sh_nodes;
#ifdef QR_DEBUG_CHECKS
      st.cap = QR_DECIDE_LDS_NODES;
#endif
      st.heap = sh_heap;
      st.desc = &sh_desc;
      decide_logic(st, ts, root_mode, active, N, recs, world, scal, thr, gf2lf, docmode, Nglobal,
                   sum_small, ss_small, root_buf, thr_off);
    } else {
      st.nodes = ts->nodes;
      st.heap = ts->heap;
      st.desc = &ts->desc;
      decide_logic(st, ts, root_mode, active, N, recs, world, scal, thr, gf2lf, docmode, Nglobal,
                   sum_small, ss_small, root_buf, thr_off);
    }
    ts->nnodes = st.nnodes;
    ts->taken = st.taken;
    ts->done = st.done;
    ts->step = st.step;
    ts->nsplits = st.nsplits;
    ts->heap_size = st.heap_size;
    ts->part_epoch = st.part_epoch;
    own[0].feature = (uint32_t)st.nnodes;    // broadcast the new extents for the copy-back
    own[0].thr_id = (uint32_t)st.heap_size;
  }
  if (staged) {
    __syncthreads();
    if (w0) {
      const size_t nn = own[0].feature, hs = own[0].thr_id;
      wave_copy8(ts->nodes, sh_nodes, nn * sizeof(QrNode));
      wave_copy8(ts->heap, sh_heap, (hs + 1) * sizeof(QrHeapItem));
      wave_copy8(&ts->desc, &sh_desc, sizeof(QrSplitDesc));
    }
  }
}

__global__ __launch_bounds__(128) void k_decide(
    QrTreeState *ts, const uint32_t N, const int flocal, const qr_split_t *recs,
    const int world, const QrScalars *__restrict__ scal,
    const double *__restrict__ part_ss, const float *__restrict__ thr,
    const int32_t *__restrict__ gf2lf, const qr_split_t *__restrict__ featrec,
    const uint32_t *__restrict__ hcnt_loc, const int docmode, const u64 Nglobal,
    const long long *__restrict__ tail, const int dworld, const uint32_t mf_k, const u64 mf_seed,
    const uint32_t F, const int root_buf, const uint32_t *__restrict__ thr_off, const size_t loc_cells) {
  QR_JITTER();
  decide_body(ts, N, flocal, recs, world, scal, part_ss, thr, gf2lf, featrec, hcnt_loc, docmode,
              Nglobal, tail, dworld, mf_k, mf_seed, F, root_buf, thr_off, loc_cells);
}

// ===========================================================================
// Pre-sorted lists (k_exact.hip): the split search of a node when the loop POPS it, as the
// reference does (rt.cc:58-90: `split(node)` follows the pop) -- the ten nodes that end up as the
// leaves of a ten-leaf tree are never searched, a fifth of the entries the eager order walks.
// The heap is keyed on the deviance, which a node has from the moment its parent is split (sums
// from the partition), so the order of pops is the reference's whatever is searched when.
//   k_xpop   accounts for the children of the split just applied (statistics, heap pushes) and
//            pops until it finds a node with deviance > 0: the step's scan target (ts->xs_*).
//   [k_xscan + k_xbest on that node's segments -> featrec[0 .. flocal)]
//   k_xapply merges the features' records: a valid split -> make_desc (the step's partition
//            launches apply it); none -> the node is a leaf (`taken`), the next step pops on.
// A step that finds no valid split has used one of the enqueued steps without splitting: the
// last k_xpop of the sequence reports a tree that still has a node to search (QrPinned::early, as
// batched growth does) and the host carries on (qr_k_exact_continue).
// ===========================================================================
__device__ __forceinline__ void xpop_logic(DecideState &st, const bool root_mode, const int32_t active,
                                           const QrSplitDesc &d, const uint32_t N, const QrScalars *scal,
                                           const double sum_small, const double ss_small, int *xs_out,
                                           const int root_buf) {
  int xs = -1;
  if (root_mode) {
    // (N: the root's documents; root_buf 2: every document, the identity lists -- 0: this
    // iteration's sample, compacted into list buffer 0, mart.cc:287-329)
    QrNode *root = &st.nodes[0];
    root->begin = 0;
    root->end = N;
    root->buf = root_buf;
    root->hslot = 0;
    root->feature = -1;
    root->thr_id = -1;
    root->threshold = 0.f;
    root->left = root->right = root->parent = -1;
    root->leaf_id = -1;
    node_stats(root, scal->root_sum, scal->root_ss, (u64)N);
    root->best_score = -1.0;
    root->best_f = root->best_t = 0xFFFFFFFFu;
    root->best_lc = root->best_rc = 0;
    st.nnodes = 1;
    st.heap_size = 0;
    st.heap[0].key = 1.7976931348623157e308;  // DBL_MAX sentinel
    st.heap[0].val = -1;
    st.taken = 0;
    st.done = 0;
    st.nsplits = 0;
    st.step = 1;
    if (root->deviance > 0.0)
      xs = 0;
    else
      st.done = 1;
  } else {
    if (active) {  // children of the split just applied (rtnode_histogram.cc:65-69, 79-86)
      QrNode *P = &st.nodes[d.node];
      QrNode *S = &st.nodes[d.small_node], *B = &st.nodes[d.big_node];
      node_stats(S, sum_small, ss_small, d.small_n);
      node_stats(B, P->sum - sum_small, P->ss - ss_small, P->count - d.small_n);
      for (int k = 0; k < 2; ++k) {
        QrNode *C = &st.nodes[k ? d.right : d.left];
        C->best_score = -1.0;
        C->best_f = C->best_t = 0xFFFFFFFFu;
        C->best_lc = C->best_rc = 0;
      }
      heap_push(st, st.nodes[d.left].deviance, d.left);    // rt.cc:76-77
      heap_push(st, st.nodes[d.right].deviance, d.right);
    }
    st.step++;
    if (!st.done) {
      while (st.heap_size > 0 && (st.nleaves_req == 0 || st.taken + st.heap_size < st.nleaves_req)) {
        const int node = st.heap[1].val;
        heap_pop(st);
        if (st.nodes[node].deviance > 0.0) {  // rt.cc:212; whether it has a split: k_xapply
          xs = node;
          break;
        }
        ++st.taken;
      }
      if (xs < 0) st.done = 1;
    }
  }
  *xs_out = xs;
}

__global__ __launch_bounds__(128) void k_xpop(
    QrTreeState *ts, const int root_mode, const int nleaves_arg, const u64 minls_arg, const uint32_t N,
    const QrScalars *__restrict__ scal, const double *__restrict__ part_ss,
    unsigned long long *__restrict__ gbest, const int flocal, const int final_call,
    int64_t *__restrict__ early, const long long early_seq, const int root_buf) {
  QR_JITTER();
  __shared__ QrNode sh_nodes[QR_DECIDE_LDS_NODES];
  __shared__ QrHeapItem sh_heap[QR_DECIDE_LDS_NODES + 2];
  __shared__ int sh_nn, sh_hs;
  const bool w0 = threadIdx.x < 64;
  // (the best-score words of the scan launch that follows start at "none")
  for (int i = threadIdx.x; i < flocal; i += blockDim.x) gbest[i] = 0ull;
  DecideState st;
  st.nleaves_req = root_mode ? nleaves_arg : ts->nleaves_req;
  st.nnodes = root_mode ? 0 : ts->nnodes;
  st.taken = root_mode ? 0 : ts->taken;
  st.done = root_mode ? 0 : ts->done;
  st.step = root_mode ? 0 : ts->step;
  st.nsplits = root_mode ? 0 : ts->nsplits;
  st.heap_size = root_mode ? 0 : ts->heap_size;
  st.part_epoch = ts->part_epoch;
  st.split_log = ts->split_log;
  st.split_log2 = nullptr;
  st.hcnt_loc = nullptr;
  st.loc = &ts->loc;
  st.flocal = flocal;
  st.desc = &ts->desc;
  const int32_t active = root_mode ? 0 : ts->desc.active;
  const QrSplitDesc d = ts->desc;
  const bool staged = !root_mode && st.nnodes + 2 <= QR_DECIDE_LDS_NODES &&
                      st.heap_size + 3 <= QR_DECIDE_LDS_NODES + 2;
  if (staged && w0) {
    wave_copy8(sh_nodes, ts->nodes, (size_t)st.nnodes * sizeof(QrNode));
    wave_copy8(sh_heap, ts->heap, (size_t)(st.heap_size + 1) * sizeof(QrHeapItem));
  }
  // squares_sum_ / sum of the directly built child: the partition workgroups' partials
  double ss_small = 0.0, sum_small = 0.0;
  if (w0 && active) {
    const uint32_t nwg = (d.end - d.begin + QR_PART_SLICE - 1) / QR_PART_SLICE;
    for (uint32_t i = threadIdx.x; i < nwg; i += 64) {
      ss_small += part_ss[2 * i];
      sum_small += part_ss[2 * i + 1];
    }
    ss_small = wave_sum(ss_small);
    sum_small = wave_sum(sum_small);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int xs = -1;
    if (staged) {
      st.nodes = sh_nodes;
#ifdef QR_DEBUG_CHECKS
      st.cap = QR_DECIDE_LDS_NODES;
#endif
      st.heap = sh_heap;
      xpop_logic(st, root_mode, active, d, N, scal, sum_small, ss_small, &xs, root_buf);
    } else {
      st.nodes = ts->nodes;
      st.heap = ts->heap;
      xpop_logic(st, root_mode, active, d, N, scal, sum_small, ss_small, &xs, root_buf);
    }
    if (root_mode) {  // (what k_tree_reset does for the phase API)
      ts->nleaves_req = nleaves_arg;
      ts->minls = minls_arg;
      ts->nleaves = 0;
      ts->l_nodes = 0;
      ts->real_steps = 0;
    }
    ts->desc.active = 0;
    ts->nnodes = st.nnodes;
    ts->taken = st.taken;
    ts->done = st.done;
    ts->step = st.step;
    ts->nsplits = st.nsplits;
    ts->heap_size = st.heap_size;
    ts->xs_node = xs;
    if (xs >= 0) {
      const QrNode &nd = st.nodes[xs];
      ts->xs_buf = nd.buf;
      ts->xs_begin = nd.begin;
      ts->xs_n = nd.end - nd.begin;
    } else {
      ts->xs_buf = 2;
      ts->xs_begin = ts->xs_n = 0;
    }
    // the last control call of the enqueued sequence still has a node to search: the host
    // carries the tree on; the leaf and score kernels behind this call leave at once
    const int inc = final_call && xs >= 0 ? 1 : 0;
    ts->incomplete = inc;
    if (final_call && early)
      __hip_atomic_store(&early[0], (int64_t)((early_seq << 16) | ((long long)(st.step & 0x7fff) << 1) | inc),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    sh_nn = st.nnodes;
    sh_hs = st.heap_size;
  }
  if (staged) {
    __syncthreads();
    if (w0) {
      wave_copy8(ts->nodes, sh_nodes, (size_t)sh_nn * sizeof(QrNode));
      wave_copy8(ts->heap, sh_heap, (size_t)(sh_hs + 1) * sizeof(QrHeapItem));
    }
  }
}

// the records of the step's scan -> the popped node's best split -> the split being applied
__global__ __launch_bounds__(64) void k_xapply(
    QrTreeState *ts, const int root_mode, const int flocal, const qr_split_t *__restrict__ featrec,
    const float *__restrict__ thr, const int32_t *__restrict__ gf2lf, const uint32_t *__restrict__ thr_off,
    const uint32_t mf_k, const u64 mf_seed, const uint32_t F, const long long *__restrict__ xcs,
    long long *__restrict__ node_tot) {
  QR_JITTER();
  const int node = ts->xs_node;
  if (node < 0) return;  // (workgroup-uniform; k_xpop left desc.active = 0)
  if (threadIdx.x == 0) ts->xs_last = 0;
  const qr_split_t a = wave_merge(1, 0, featrec, flocal, mf_k, mf_seed, (uint32_t)node, F);
  if (threadIdx.x != 0) return;
  __shared__ qr_split_t own[2];
  own[0] = a;
  DecideState st;
  st.nleaves_req = ts->nleaves_req;
  st.nnodes = ts->nnodes;
  st.taken = ts->taken;
  st.done = ts->done;
  st.step = ts->step;
  st.nsplits = ts->nsplits;
  st.heap_size = ts->heap_size;
  st.part_epoch = ts->part_epoch;
  st.split_log = ts->split_log;
  st.split_log2 = nullptr;
  st.hcnt_loc = nullptr;
  st.loc = &ts->loc;
  st.flocal = flocal;
  st.nodes = ts->nodes;
  st.heap = ts->heap;
  st.desc = &ts->desc;
  QrNode *nd = &ts->nodes[node];
  node_set_best(nd, own, 1, 0);
  if (nd->best_f != 0xFFFFFFFFu) {  // (deviance > 0: k_xpop)
    make_desc(st, node, thr, gf2lf, thr_off);
    // the children's fixed-point totals: the winner's cumulative sum IS the left child's
    const long long sl = xcs[ts->desc.owner_local];
    node_tot[ts->desc.left] = sl;
    node_tot[ts->desc.right] = node_tot[node] - sl;
    ts->real_steps = ts->real_steps + 1;
    // (rt.cc:58-90's loop goes on while taken + heap < leaves: with the two children pushed it stops)
    ts->xs_last = st.nleaves_req != 0 && st.taken + st.heap_size + 2 >= st.nleaves_req ? 1 : 0;
  } else if (root_mode) {
    st.done = 1;     // rt.cc:312: the root has no split
  } else {
    ++st.taken;      // a leaf after all; the next step pops on
  }
  ts->nnodes = st.nnodes;
  ts->taken = st.taken;
  ts->done = st.done;
  ts->nsplits = st.nsplits;
  ts->part_epoch = st.part_epoch;
}

// ===========================================================================
// k_decide_batch: RegressionTree::fit with up to QR_BATCH splits applied per step.
//
// The reference's loop (rt.cc:58-90) pops the heap's maximum, splits it, pushes
// the children, and repeats: one split per step, each a chain of dependent
// launches here.  A node's best split is known as soon as its histogram exists and
// does not depend on what else happens to the tree, so a step can ALSO apply the
// split of the most promising other candidate of the heap ("ahead of its turn"):
// its partition writes the other list buffer and its children live under
// provisional node ids, so nothing the reference's order of events can observe
// changes.  When the loop later pops that node, the split is already there: the
// children are moved to the ids the sequential order assigns (histogram slots stay
// where they are) and the loop carries on without a round trip; if the loop never
// pops it (leaf budget exhausted), the node stays the leaf it was.  Heap pushes and
// pops, node numbering and the split log are exactly the sequential ones.
// ===========================================================================
__device__ QR_CTRL_FN void batch_child_init(QrNode *ch, uint32_t b, uint32_t e, int buf,
                                                 int slot, int parent) {
  ch->begin = b;
  ch->end = e;
  ch->buf = buf;
  ch->hslot = slot;
  ch->feature = -1;
  ch->thr_id = -1;
  ch->threshold = 0.f;
  ch->left = ch->right = -1;
  ch->parent = parent;
  ch->leaf_id = -1;
  ch->pre = 0;
  ch->pre_l = ch->pre_r = -1;
}

// the split of `node` becomes part of the tree, with children li / ri
__device__ QR_CTRL_FN void batch_commit(DecideState &st, int node, int li, int ri) {
  QrNode *nd = &st.nodes[node];
  nd->feature = (int32_t)nd->best_f;
  nd->thr_id = (int32_t)nd->best_t;
  nd->threshold = nd->best_thr;
  nd->left = li;
  nd->right = ri;
  qr_split_t lg;
  lg.score = nd->best_score;
  lg.feature = nd->best_f;
  lg.thr_id = nd->best_t;
  lg.lcount = nd->best_lc;
  lg.rcount = nd->best_rc;
  if (st.split_log) st.split_log[st.nsplits] = lg;
  if (st.split_log2) st.split_log2[st.nsplits] = lg;
  ++st.nsplits;
}

// what the control lane carries besides DecideState
struct BatchState {
  int32_t next_prov, next_slot, spec_made, spec_used;
  QrLevelNode *ln;  // the next batch's descriptors
};

// job j of the next batch: apply the best split of `node`
__device__ QR_CTRL_FN void batch_make_job(DecideState &st, BatchState &bs, int j, int node,
                                               bool in_turn) {
  QrNode *nd = &st.nodes[node];
  int li;
  if (in_turn) {
    li = st.nnodes;
    st.nnodes += 2;
  } else {
    li = bs.next_prov;
    bs.next_prov += 2;
  }
  const int ri = li + 1;
#ifdef QR_DEBUG_CHECKS
  if (!QR_DBG_OK(li >= 0 && ri < st.cap && j >= 0 && j < QR_BATCH && (u64)(bs.next_slot + 1) < qr_dbg_caps[4] &&
                 st.nsplits < QR_MAXNODES, 5, ((u64)(uint32_t)li << 32) | (uint32_t)bs.next_slot, ((u64)(uint32_t)st.cap << 32) | (uint32_t)j))
    return;
#endif
  const int sl = bs.next_slot, sr = sl + 1;
  bs.next_slot += 2;
  // (document-sharded rank: [begin, end) are positions in its OWN lists -- its own left count comes
  // from the prefix of its local counts, as make_desc does; which child is built directly, and the
  // node statistics, follow the GLOBAL counts so that every rank takes the same decisions)
  const bool lsmall = nd->best_lc <= nd->best_rc;
  uint32_t lc = (uint32_t)nd->best_lc, rc = (uint32_t)nd->best_rc;
  if (st.hcnt_loc) {
    lc = st.hcnt_loc[((size_t)nd->hslot * st.flocal + nd->best_lf) * 256 + nd->best_t];
    rc = (nd->end - nd->begin) - lc;
  }
  const int dst = nd->buf == 0 ? 1 : 0;
  batch_child_init(&st.nodes[li], nd->begin, nd->begin + lc, dst, sl, node);
  batch_child_init(&st.nodes[ri], nd->begin + lc, nd->end, dst, sr, node);
  QrLevelNode *ln = &bs.ln[j];
  ln->active = 1;
  ln->src_buf = nd->buf;
  ln->dst_buf = dst;
  ln->small_is_left = lsmall;
  ln->begin = nd->begin;
  ln->end = nd->end;
  ln->lcount = lc;
  ln->small_begin = ln->small_is_left ? nd->begin : nd->begin + lc;
  ln->small_n = ln->small_is_left ? lc : rc;
  ln->parent_slot = nd->hslot;
  ln->small_slot = ln->small_is_left ? sl : sr;
  ln->big_slot = ln->small_is_left ? sr : sl;
  ln->q = 0;
  ln->slot_base = 0;
  ln->part_first = 0;
  ln->owner_local = nd->best_lf;
  ln->thr_id = nd->best_t;
  ln->node = node;
  ln->left = li;
  ln->right = ri;
  ln->spec = in_turn ? 0 : 1;
  ln->pad = 0;
  if (in_turn) {
    batch_commit(st, node, li, ri);
  } else {
    nd->pre = 1;
    nd->pre_l = li;
    nd->pre_r = ri;
    bs.spec_made++;
  }
}

// One child of the batch just applied: directly accumulated child or sibling by
// subtraction (rtnode_histogram.cc:65-69, 79-86), and its best split.
__device__ __forceinline__ void batch_child_stats(QrNode *nodes, const QrLevelNode &ln,
                                                  const int which, const qr_split_t *own2,
                                                  const double sum_small, const double ss_small,
                                                  const int32_t lf, const float thrv) {
  const QrNode *P = &nodes[ln.node];
  QrNode *C = &nodes[which ? ln.right : ln.left];
  const bool is_small = (which == 0) == (ln.small_is_left != 0);
  // (the child's count over ALL ranks' documents: ln.small_n is this rank's own on a
  // document-sharded context)
  const u64 gsmall = ln.small_is_left ? P->best_lc : P->best_rc;
  if (is_small)
    node_stats(C, sum_small, ss_small, gsmall);
  else
    node_stats(C, P->sum - sum_small, P->ss - ss_small, P->count - gsmall);
  node_set_best(C, own2, 1, which);
  C->best_lf = lf;
  C->best_thr = thrv;
}

// Splits applied ahead of their turn shorten the chain of dependent launches at the price
// of child histograms that are sometimes never used: that pays while a step is bound by its
// launches, not by the documents it moves.  Measured (scripts/spec_bench.py, ms per
// iteration with / without early splits): 1M documents 0.553 / 0.629, 2M 0.859 / 0.940,
// 4M 1.553 / 1.563, 8M 2.925 / 2.860.
#ifndef QR_SPEC_MAX_DOCS
#define QR_SPEC_MAX_DOCS 6000000u
#endif
// the control lane's step on the state `st` points at (force-inlined into an
// LDS-staged and a device-resident call site, like decide_logic)
__device__ __forceinline__ int batch_logic(DecideState &st, BatchState &bs, const bool root_mode,
                                           const int njobs, const QrLevelNode *prev,
                                           const uint32_t N, const qr_split_t *own,
                                           const QrScalars *scal, const int32_t *own_lf,
                                           const float *own_thr, const int root_buf, const u64 Ncount,
                                           const u64 spec_docs) {
  // (Ncount: the root's documents over all ranks -- N itself except on a document-sharded rank;
  // spec_docs: what a rank's launches work on -- N, or the ranks' average on document shards, the
  // same number on every rank -- decides whether splits ahead of their turn pay)
  int nj = 0;
#ifdef QR_STEP_TIMING
  long long lt[5];
  lt[0] = lt[1] = lt[2] = lt[3] = lt[4] = clock64();
#define QR_LT(i) lt[i] = clock64()
#else
#define QR_LT(i)
#endif
  if (root_mode) {
    QrNode *root = &st.nodes[0];
    batch_child_init(root, 0, N, root_buf, 0, -1);
    node_stats(root, scal->root_sum, scal->root_ss, Ncount);
    node_set_best(root, own, 1, 0);
    root->best_lf = own_lf[0];
    root->best_thr = own_thr[0];
    st.nnodes = 1;
    st.heap_size = 0;
    st.heap[0].key = 1.7976931348623157e308;  // DBL_MAX sentinel
    st.heap[0].val = -1;
    st.taken = 0;
    st.done = 0;
    st.nsplits = 0;
    st.step = 1;
    bs.next_prov = 2 * st.nleaves_req + 1;
    bs.next_slot = 1;
    bs.spec_made = bs.spec_used = 0;
    if (node_splittable(root)) {
      batch_make_job(st, bs, 0, 0, true);
      nj = 1;
    } else {
      st.done = 1;
    }
  } else {
    // (the children of the batch just applied got their statistics and best splits
    // from batch_child_stats, one lane per child)
    if (njobs > 0) {  // job 0 was the split the sequential loop was waiting for
      const int l0 = prev[0].left, r0 = prev[0].right;
      heap_push(st, st.nodes[l0].deviance, l0);  // rt.cc:76-77
      heap_push(st, st.nodes[r0].deviance, r0);
    }
    QR_LT(1);
    st.step++;
    if (!st.done) {
      bool found = false;
      while (st.heap_size > 0 &&
             (st.nleaves_req == 0 || st.taken + st.heap_size < st.nleaves_req)) {
        const int node = st.heap[1].val;
        heap_pop(st);
        QrNode *nd = &st.nodes[node];
        if (!node_splittable(nd)) {
          ++st.taken;
          continue;
        }
        if (nd->pre) {  // applied ahead of its turn: the children take their final ids
          const int li = st.nnodes, ri = li + 1;
#ifdef QR_DEBUG_CHECKS
          if (!QR_DBG_OK(ri < st.cap && nd->pre_l >= 0 && nd->pre_r < st.cap, 7, li, nd->pre_r)) break;
#endif
          st.nnodes += 2;
          st.nodes[li] = st.nodes[nd->pre_l];
          st.nodes[ri] = st.nodes[nd->pre_r];
          nd->pre = 0;
          batch_commit(st, node, li, ri);
          heap_push(st, st.nodes[li].deviance, li);
          heap_push(st, st.nodes[ri].deviance, ri);
          bs.spec_used++;
          continue;
        }
        batch_make_job(st, bs, 0, node, true);
        nj = 1;
        found = true;
        break;
      }
      if (!found) st.done = 1;
    }
  }
  QR_LT(2);
  // candidates applied ahead of their turn: the largest deviances left in the heap,
  // as long as the leaf budget can still reach them
  if (nj == 1 && spec_docs < QR_SPEC_MAX_DOCS) {
    while (nj < QR_BATCH) {
      if (st.nleaves_req != 0 && st.nleaves_req - (st.taken + st.heap_size + 2) < nj) break;
      int pick = -1;
      double key = 0.0;
      for (int i = 1; i <= st.heap_size; ++i) {
        const int v = st.heap[i].val;
        const QrNode *cand = &st.nodes[v];
        if (cand->pre || !node_splittable(cand)) continue;
        if (pick < 0 || st.heap[i].key > key) {
          pick = v;
          key = st.heap[i].key;
        }
      }
      if (pick < 0) break;
      batch_make_job(st, bs, nj, pick, false);
      ++nj;
    }
  }
#ifdef QR_STEP_TIMING
  QR_LT(3);
  if (blockIdx.x == 0)
    printf("batch_logic wg0: pushes %lld pops+job0 %lld candidates+job1 %lld (heap %d)\n", lt[1] - lt[0], lt[2] - lt[1],
           lt[3] - lt[2], st.heap_size);
#endif
  return nj;
}

// The control step of a batch, run by one workgroup of 128 * QR_BATCH threads.
// `root_mode` = first call of a tree; `stage_nodes` > 0: the node records [0,
// stage_nodes) and the heap fit the LDS copies (host: 4 * nleaves + 1 <= 96).
// Everything the step needs from memory is requested up front, before any of it is
// used: it is one link of the per-step chain, and a dependent global round trip costs
// it ~1 us.
//   FUSED = false (k_decide_batch): the state is updated in place (tin == tout).
//   FUSED = true (k_decide_part): EVERY workgroup of the partition launch runs the
//   step redundantly on its own LDS copy (the step is deterministic) and then
//   partitions its slice, which saves the launch boundary between the two; only
//   the `writer` workgroup publishes the new state -- into `tout`, a second copy of
//   the tree state, because the others may still be reading `tin` (the split log,
//   append-only, goes to both copies: `tlog2`).  Needs the staged mode; the epoch of
//   the partition's look-back granules comes from the host (`epoch_arg`).
// Out (LDS, for the caller): sh_next / sh_pw0 / sh_njp = the new batch's nodes, the
// prefix of their partition workgroups, their number; sh_epoch.
// CAP = node records the LDS copy holds: 96 (trees of up to 22 leaves: 4 L + 1 final
// and provisional ids + what a step can add) or 264 (up to 64 leaves) -- two
// instantiations, because the larger footprint costs the small trees ~7 us per iteration
#define QR_BATCH_LDS_SMALL 96
#define QR_BATCH_LDS_LARGE 264
template <bool FUSED, int CAP>
__device__ __forceinline__ void batch_step(
    const QrTreeState *tin, QrTreeState *tout, QrTreeState *tlog2, const bool writer,
    const uint32_t epoch_arg, QrLevelNode *sh_next, uint32_t *sh_pw0, int *sh_njp,
    uint32_t *sh_epoch, const int root_mode, const int nleaves_arg, const u64 minls_arg,
    const int stage_nodes, const uint32_t N, const int flocal,
    const QrScalars *__restrict__ scal, const double *__restrict__ part_ss,
    const qr_split_t *__restrict__ featrec, const float *__restrict__ featthr, const uint32_t F,
    const int root_buf, const int G, const QrBlock *__restrict__ blocks, const int nblocks,
    QrHistWg *__restrict__ hist_wg, const uint32_t hist_grid, QrPartWg *__restrict__ part_wg,
    const uint32_t part_grid, QrPlan *__restrict__ plans, QrScanWg *__restrict__ scan_wg,
    const int final_call = 0, const uint32_t *__restrict__ hcnt_loc = nullptr, const u64 Nglobal = 0,
    const u64 spec_docs_doc = 0) {
  QrTreeState *const ts = tout;  // where the writer publishes
  __shared__ QrPlan sh_plan[QR_BATCH];
  __shared__ qr_split_t own[2 * QR_BATCH];
  __shared__ double sh_sum[QR_BATCH], sh_ss[QR_BATCH];
  __shared__ uint32_t sh_hw0[QR_BATCH + 1], sh_q;
  __shared__ int sh_nj, sh_hi, sh_hs, sh_nn;
  __shared__ QrNode sh_nodes[CAP];
  __shared__ QrHeapItem sh_heap[CAP + 2];
  __shared__ QrLevelNode sh_prev[QR_BATCH];
  __shared__ int32_t own_lf[2 * QR_BATCH];
  __shared__ float own_thr[2 * QR_BATCH];
  __shared__ QrBlock sh_blk[QR_MAXBLK];
  static_assert(sizeof(QrLevelNode) % 4 == 0, "copied as 4-byte words");
  constexpr int NV = (int)((CAP * sizeof(QrNode) / 8 + 128 * QR_BATCH - 1) / (128 * QR_BATCH));
  constexpr int NH = (int)(((CAP + 2) * sizeof(QrHeapItem) / 8 + 128 * QR_BATCH - 1) / (128 * QR_BATCH));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool staged = stage_nodes > 0;
#ifdef QR_STEP_TIMING
  long long bt[6];
  bt[0] = clock64();
#define QR_BT(i) bt[i] = clock64()
#else
#define QR_BT(i)
#endif
  // ---- requests: header, the previous batch's descriptors, node records, heap
  const int njobs_raw = tin->l_nodes;
  // (the first call of a tree starts from its arguments, not from what the last tree left)
  const int32_t h_nleaves_req = root_mode ? nleaves_arg : tin->nleaves_req, h_nnodes = tin->nnodes, h_taken = tin->taken,
                h_done = tin->done, h_step = tin->step, h_nsplits = tin->nsplits,
                h_heap_size = tin->heap_size, h_next_prov = tin->next_prov,
                h_next_slot = tin->next_slot, h_spec_made = tin->spec_made,
                h_spec_used = tin->spec_used, h_real_steps = root_mode ? 0 : tin->real_steps;
  const uint32_t h_part_epoch = tin->part_epoch;
  const u64 h_minls = root_mode ? minls_arg : tin->minls;
#ifdef QR_STEP_TIMING_FINE
  long long ft[5];
  __builtin_amdgcn_s_waitcnt(0);
  ft[0] = clock64();
#endif
  const QrLevelNode myln = tin->lnode[wave < QR_BATCH ? wave : 0];
  // sums of the directly built children of the batch just applied: k_redscan has added
  // up the partition workgroups' partials (one dependent read less on this chain)
  const double js_ss = part_ss[2 * (wave < QR_BATCH ? wave : 0)];
  const double js_sum = part_ss[2 * (wave < QR_BATCH ? wave : 0) + 1];
  u64 v[NV];
  u64 hv[NH];
  const size_t nw = staged && !root_mode ? (size_t)stage_nodes * sizeof(QrNode) / 8 : 0;
  const size_t nh = staged && !root_mode ? (size_t)(stage_nodes + 2) * sizeof(QrHeapItem) / 8 : 0;
  {
    const u64 *src = reinterpret_cast<const u64 *>(tin->nodes);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const size_t i = threadIdx.x + (size_t)k * blockDim.x;
      v[k] = i < nw ? src[i] : 0;
    }
    const u64 *hs = reinterpret_cast<const u64 *>(tin->heap);
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const size_t i = threadIdx.x + (size_t)k * blockDim.x;
      hv[k] = i < nh ? hs[i] : 0;
    }
  }
#ifdef QR_STEP_TIMING_FINE
  __builtin_amdgcn_s_waitcnt(0);
  ft[1] = clock64();
#endif
  // wave 2j + which: the per-feature records of job j's left / right child (stale
  // records of a job that does not exist are merged too and ignored)
  int my_lf = -1;
  float my_thr = 0.f;
  const qr_split_t mine = wave_merge(0, wave, featrec, flocal, 0, 0, 0, F, &my_lf, featthr, &my_thr);
#ifdef QR_STEP_TIMING_FINE
  __builtin_amdgcn_s_waitcnt(0);
  ft[2] = clock64();
#endif
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) sh_blk[b] = blocks[b];
#ifdef QR_STEP_TIMING_FINE
  __builtin_amdgcn_s_waitcnt(0);
  ft[3] = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("batch_step wg0 loads: header %lld lnode+sums+nodes+heap %lld merge %lld blocks %lld\n", ft[0] - bt[0],
           ft[1] - ft[0], ft[2] - ft[1], ft[3] - ft[2]);
#endif
  // ---- uses
  const int njobs = root_mode ? 0 : njobs_raw;  // the batch that has just been applied
  {
    u64 *dst = reinterpret_cast<u64 *>(sh_nodes);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const size_t i = threadIdx.x + (size_t)k * blockDim.x;
      if (i < nw) dst[i] = v[k];
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const size_t i = threadIdx.x + (size_t)k * blockDim.x;
      if (i < nh) reinterpret_cast<u64 *>(sh_heap)[i] = hv[k];
    }
  }
  if (lane == 0) {
    own[wave] = mine;
    own_lf[wave] = my_lf;
    own_thr[wave] = my_thr;
  }
  if (wave < njobs && lane == 0) sh_prev[wave] = myln;
  if (wave < njobs && lane == 0) {
    sh_ss[wave] = js_ss;
    sh_sum[wave] = js_sum;
  }
  __syncthreads();
  QR_BT(1);
  if (threadIdx.x < 2 * njobs) {  // one lane per child of the batch just applied
    const int j = threadIdx.x >> 1, which = threadIdx.x & 1;
    batch_child_stats(staged ? sh_nodes : ts->nodes, sh_prev[j], which, own + 2 * j, sh_sum[j],
                      sh_ss[j], own_lf[2 * j + which], own_thr[2 * j + which]);
  }
  __syncthreads();
  QR_BT(2);
  if (threadIdx.x == 0) {
    DecideState st;
    st.nleaves_req = h_nleaves_req;
    st.nnodes = h_nnodes;
    st.taken = h_taken;
    st.done = h_done;
    st.step = h_step;
    st.nsplits = h_nsplits;
    st.heap_size = h_heap_size;
    st.part_epoch = h_part_epoch;
    st.split_log = writer ? ts->split_log : nullptr;
    st.split_log2 = (writer && tlog2) ? tlog2->split_log : nullptr;
    st.desc = &ts->desc;
    st.hcnt_loc = hcnt_loc;   // (document-sharded ranks: their own cumulative counts, k_bd_reduce)
    st.loc = &ts->loc;
    st.flocal = flocal;
    BatchState bs;
    bs.next_prov = h_next_prov;
    bs.next_slot = h_next_slot;
    bs.spec_made = h_spec_made;
    bs.spec_used = h_spec_used;
    bs.ln = sh_next;
    int nj;
    if (staged) {
      st.nodes = sh_nodes;
#ifdef QR_DEBUG_CHECKS
      st.cap = CAP;
#endif
      st.heap = sh_heap;
      nj = batch_logic(st, bs, root_mode, njobs, sh_prev, N, own, scal, own_lf, own_thr, root_buf, hcnt_loc ? Nglobal : (u64)N,
                       hcnt_loc ? spec_docs_doc : (u64)N);
    } else {
      st.nodes = ts->nodes;
      st.heap = ts->heap;
      nj = batch_logic(st, bs, root_mode, njobs, sh_prev, N, own, scal, own_lf, own_thr, root_buf, hcnt_loc ? Nglobal : (u64)N,
                       hcnt_loc ? spec_docs_doc : (u64)N);
    }
    // one plan quantum for the whole batch (as for a level of an oblivious tree)
    if (nj > 0) {
      unsigned long long tot_small = 0;
      for (int j = 0; j < nj; ++j) tot_small += sh_next[j].small_n;
      if (tot_small == 0) tot_small = 1;
      const int spare = G - nj * nblocks;
      sh_q = qr_plan_quantum(tot_small * qr_plan_wsum(nblocks, sh_blk),
                             spare > G / 4 ? spare : G / 4);
      st.part_epoch++;
    }
    {  // the partition workgroups of the batch's nodes
      uint32_t pw0 = 0;
      for (int j = 0; j < nj; ++j) {
        sh_next[j].part_first = pw0;
        sh_pw0[j] = pw0;
        pw0 += (sh_next[j].end - sh_next[j].begin + QR_PART_SLICE - 1) / QR_PART_SLICE;
      }
      sh_pw0[nj] = pw0;
    }
    sh_nj = nj;
    *sh_njp = nj;
    *sh_epoch = FUSED ? epoch_arg : st.part_epoch;
    sh_hi = bs.next_prov;
    sh_hs = st.heap_size;
    sh_nn = st.nnodes;
    if (writer) {
      if (root_mode) {  // what k_tree_reset does for the one-split-per-step path
        ts->desc.active = 0;
        ts->nleaves = 0;
      }
      // (the state ping-pongs between two copies: the tree's parameters travel with it)
      ts->nleaves_req = h_nleaves_req;
      ts->minls = h_minls;
      ts->l_nodes = nj;
      ts->nnodes = st.nnodes;
      ts->taken = st.taken;
      ts->done = st.done;
      ts->step = st.step;
      ts->nsplits = st.nsplits;
      ts->heap_size = st.heap_size;
      ts->part_epoch = st.part_epoch;
      ts->next_prov = bs.next_prov;
      ts->next_slot = bs.next_slot;
      ts->spec_made = bs.spec_made;
      ts->spec_used = bs.spec_used;
      ts->real_steps = h_real_steps + (nj > 0 ? 1 : 0);
      // the last control call of the enqueued sequence still found a batch to apply: the
      // host guessed too few steps (the leaf / score kernels leave, the host carries on)
      ts->incomplete = final_call && nj > 0 ? 1 : 0;
      ts->l_part_wgs = sh_pw0[nj];
    }
  }
  __syncthreads();
  QR_BT(3);
  const int nj = sh_nj;
#ifdef QR_STEP_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("batch_step wg0: loads %lld child stats %lld logic %lld (nj %d)\n", bt[1] - bt[0], bt[2] - bt[1],
           bt[3] - bt[2], sh_nj);
#endif
  if (!writer) return;  // (workgroup-uniform) the rest publishes the new state
  // the plans of the batch's nodes, one lane each (wide-bin contexts, nblocks == 0: their
  // histogram and scan launches read the jobs themselves, k_wide.hip)
  if (lane == 0 && wave < nj && nblocks > 0)
    qr_make_plan(sh_next[wave].small_n, nblocks, sh_blk, sh_q, &sh_plan[wave]);
  if (staged) {  // (while they compute: the node records and the heap go back)
    u64 *dst = reinterpret_cast<u64 *>(ts->nodes);
    const u64 *src = reinterpret_cast<const u64 *>(sh_nodes);
    for (size_t i = threadIdx.x; i < (size_t)sh_hi * sizeof(QrNode) / 8; i += blockDim.x) dst[i] = src[i];
    u64 *hd = reinterpret_cast<u64 *>(ts->heap);
    const u64 *hs = reinterpret_cast<const u64 *>(sh_heap);
    for (size_t i = threadIdx.x; i < (size_t)(sh_hs + 1) * sizeof(QrHeapItem) / 8; i += blockDim.x)
      hd[i] = hs[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t hw0 = 0, slot0 = 0;
    for (int j = 0; j < nj; ++j) {
      QrLevelNode *ln = &sh_next[j];
      const uint32_t hw = nblocks > 0 ? (uint32_t)sh_plan[j].wg_start[nblocks] : 0u;
      ln->q = sh_q;
      ln->slot_base = slot0;
      sh_hw0[j] = hw0;
      hw0 += hw;
      slot0 += nblocks > 0 ? hw * (uint32_t)sh_plan[j].kmax : 0u;
    }
    sh_hw0[nj] = hw0;
    ts->l_hist_wgs = hw0;
  }
  __syncthreads();
  // the descriptors and plans (k_redscan reads them) and every
  // workgroup's share of the next partition / histogram launches; workgroups beyond
  // the batch's needs get an empty one
  {
    uint32_t *dst = reinterpret_cast<uint32_t *>(ts->lnode);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sh_next);
    for (size_t i = threadIdx.x; i < (size_t)nj * sizeof(QrLevelNode) / 4; i += blockDim.x) dst[i] = src[i];
    uint32_t *pd = reinterpret_cast<uint32_t *>(plans);
    const uint32_t *ps = reinterpret_cast<const uint32_t *>(sh_plan);
    if (nblocks > 0)
      for (size_t i = threadIdx.x; i < (size_t)nj * sizeof(QrPlan) / 4; i += blockDim.x) pd[i] = ps[i];
  }
  // (wide: no per-workgroup histogram / scan descriptors)
  for (uint32_t x = threadIdx.x; nblocks > 0 && x < hist_grid; x += blockDim.x) {
    QrHistWg d;
    d.begin = d.count = d.slot = 0;
    d.block = 0;
    d.buf = d.fw = 0;
    d.off256 = d.pad = 0;
    if (x < sh_hw0[nj]) {
      int j = 0;
      while (j + 1 < nj && x >= sh_hw0[j + 1]) ++j;
      const QrLevelNode &ln = sh_next[j];
      const QrPlan &pl = sh_plan[j];
      const uint32_t xw = x - sh_hw0[j];
      int b = 0;
      while (b + 1 < nblocks && (int)xw >= pl.wg_start[b + 1]) ++b;
      const uint32_t per = pl.per[b];
      const uint32_t r0 = (xw - (uint32_t)pl.wg_start[b]) * per;
      const uint32_t r1 = r0 + per < ln.small_n ? r0 + per : ln.small_n;
      d.begin = ln.small_begin + r0;
      d.count = r1 > r0 ? r1 - r0 : 0u;
      d.slot = ln.slot_base + xw * (uint32_t)pl.kmax;
      d.block = (uint16_t)b;
      d.buf = (uint8_t)ln.dst_buf;
      d.fw = (uint8_t)sh_blk[b].fw;
      d.off256 = (uint32_t)(sh_blk[b].off >> 8);
    }
    hist_wg[x] = d;
  }
  for (uint32_t x = threadIdx.x; nblocks > 0 && x < (uint32_t)(QR_BATCH * flocal); x += blockDim.x) {
    const int j = (int)x / flocal, lf = (int)x - j * flocal;
    QrScanWg d;
    d.active = 0;
    d.slot0 = d.total = d.per = d.n = d.col = d.part_first = d.part_nwg = d.pad = 0;
    d.kmax = 1;
    d.small_slot = d.big_slot = d.parent_slot = d.small_is_left = 0;
    if (j < nj) {
      const QrLevelNode &ln = sh_next[j];
      const QrPlan &pl = sh_plan[j];
      int b = 0;
      for (int i = 0; i < nblocks; ++i)
        if (lf >= sh_blk[i].lf0 && lf < sh_blk[i].lf0 + sh_blk[i].nreal) b = i;
      d.active = 1;
      d.slot0 = ln.slot_base + (uint32_t)(pl.wg_start[b] * pl.kmax);
      d.total = (uint32_t)((pl.wg_start[b + 1] - pl.wg_start[b]) * pl.kmax);
      d.per = pl.per[b];
      d.n = ln.small_n;
      d.kmax = pl.kmax;
      d.small_slot = ln.small_slot;
      d.big_slot = ln.big_slot;
      d.parent_slot = ln.parent_slot;
      d.small_is_left = ln.small_is_left;
      d.col = (uint32_t)(lf - sh_blk[b].lf0);
      d.part_first = ln.part_first;
      d.part_nwg = (ln.end - ln.begin + QR_PART_SLICE - 1) / QR_PART_SLICE;
    }
    scan_wg[x] = d;
  }
  if (!FUSED)
  for (uint32_t w = threadIdx.x; w < part_grid; w += blockDim.x) {
    QrPartWg d;
    d.begin = d.n = d.lcount = d.first = d.w = 0;
    d.owner_local = 0;
    d.thr_id = 0;
    d.src_buf = d.dst_buf = d.small_is_left = d.pad = 0;
    if (w < sh_pw0[nj]) {
      int j = 0;
      while (j + 1 < nj && w >= sh_pw0[j + 1]) ++j;
      const QrLevelNode &ln = sh_next[j];
      d.begin = ln.begin;
      d.n = ln.end - ln.begin;
      d.lcount = ln.lcount;
      d.first = ln.part_first;
      d.w = w - ln.part_first;
      d.owner_local = ln.owner_local;
      d.thr_id = ln.thr_id;
      d.src_buf = (uint8_t)ln.src_buf;
      d.dst_buf = (uint8_t)ln.dst_buf;
      d.small_is_left = (uint8_t)ln.small_is_left;
    }
    part_wg[w] = d;
  }
  // The last control call of a staged tree also numbers the leaves (what k_finish does in
  // a launch of its own, ~4.5 us): RTNode::save_leaves (rtnode.cc:34-46) visits them
  // left first, and a node's left child holds the lower positions of its segment, so a
  // leaf's DFS index is the number of leaves that begin before it -- no walk, one thread
  // per node.  (A tree with an EMPTY leaf -- min-leaf-support 0 -- can tie two leaves on
  // `begin`: one lane walks it instead, as k_finish does.)
  if (!FUSED && final_call && staged && nj == 0) {
    const int nn = sh_nn;
    __shared__ int fin_nl, fin_tie;
    __shared__ int32_t fin_leaf[CAP], fin_stack[CAP];
    if (threadIdx.x == 0) fin_nl = fin_tie = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nn; i += blockDim.x) {
      const bool leaf = sh_nodes[i].feature < 0;
      if (leaf && sh_nodes[i].end == sh_nodes[i].begin) fin_tie = 1;
      if (leaf) atomicAdd(&fin_nl, 1);
    }
    __syncthreads();
    if (!fin_tie) {
      for (int i = threadIdx.x; i < nn; i += blockDim.x) {
        if (sh_nodes[i].feature >= 0) continue;
        const uint32_t b = sh_nodes[i].begin;
        int l = 0;
        for (int j = 0; j < nn; ++j) l += (sh_nodes[j].feature < 0 && sh_nodes[j].begin < b) ? 1 : 0;
        fin_leaf[l] = i;
      }
    } else if (threadIdx.x == 0) {
      int sp = 0, nl = 0;
      fin_stack[sp++] = 0;
      while (sp > 0) {
        const int n = fin_stack[--sp];
        if (sh_nodes[n].feature < 0) {
          fin_leaf[nl++] = n;
        } else {
          fin_stack[sp++] = sh_nodes[n].right;
          fin_stack[sp++] = sh_nodes[n].left;
        }
      }
    }
    __syncthreads();
    const int nl = fin_nl;
    for (int l = threadIdx.x; l < nl; l += blockDim.x) {
      const int n = fin_leaf[l];
      ts->nodes[n].leaf_id = l;
      ts->leaf_nodes[l] = n;
      ts->leaf_begin[l] = sh_nodes[n].begin;
    }
    if (threadIdx.x == 0) {
      ts->nleaves = nl;
      ts->leaf_begin[nl] = sh_nodes[0].end;
    }
  }
}

template <int CAP>
__global__ __launch_bounds__(128 * QR_BATCH) void k_decide_batch(
    QrTreeState *ts, const int root_mode, const int nleaves_arg, const u64 minls_arg,
    const int stage_nodes, const uint32_t N, const int flocal,
    const QrScalars *__restrict__ scal, const double *__restrict__ part_ss,
    const qr_split_t *__restrict__ featrec, const float *__restrict__ featthr, const uint32_t F,
    const int root_buf, const int G, const QrBlock *__restrict__ blocks, const int nblocks,
    QrHistWg *__restrict__ hist_wg, const uint32_t hist_grid, QrPartWg *__restrict__ part_wg,
    const uint32_t part_grid, QrPlan *__restrict__ plans, QrScanWg *__restrict__ scan_wg,
    const QrTreeState *tin, const int final_call, int64_t *__restrict__ early, const long long early_seq,
    const uint32_t *__restrict__ hcnt_loc, const u64 Nglobal, const u64 spec_docs) {
  QR_JITTER();
  __shared__ QrLevelNode sh_next[QR_BATCH];
  __shared__ uint32_t sh_pw0[QR_BATCH + 1], sh_epoch;
  __shared__ int sh_nj;
  // (tin != ts: the last call of a tree grown by k_decide_part, whose state ping-pongs;
  // hcnt_loc != null: a document-sharded rank, N = its own documents, Nglobal = everybody's)
  batch_step<false, CAP>(tin ? tin : ts, ts, nullptr, true, 0u, sh_next, sh_pw0, &sh_nj, &sh_epoch,
                    root_mode, nleaves_arg, minls_arg, stage_nodes, N, flocal, scal, part_ss, featrec,
                    featthr, F, root_buf, G, blocks, nblocks, hist_wg, hist_grid, part_wg, part_grid,
                    plans, scan_wg, final_call, hcnt_loc, Nglobal, spec_docs);
  if (final_call && early) {  // QrPinned::early: the host settles the tree on this
    __syncthreads();
    if (threadIdx.x == 0) {
      // ONE 8-byte word -- sequence number << 16 | steps << 1 | incomplete -- so that no fence
      // (a PCIe round trip of ~3 us at the end of this launch) has to order it behind its data
      const long long inc = __hip_atomic_load(&ts->incomplete, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
      const long long st = __hip_atomic_load(&ts->real_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x7fff;
      __hip_atomic_store(&early[0], (int64_t)((early_seq << 16) | (st << 1) | inc), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ===========================================================================
// Partition (rt.cc:325-334): stable, x <= threshold  <=>  bin <= thr_id.
// ===========================================================================
__device__ __forceinline__ bool go_left(const QrSplitDesc &d, uint32_t p, uint32_t id,
                                        const uint8_t *fm, uint32_t N,
                                        const uint32_t *mask, int use_mask) {
  if (use_mask == 3)  // feature-sharded level-wise growth: go-left bits by DOCUMENT (k_obl_mark)
    return (mask[id >> 5] >> (id & 31)) & 1u;
  if (use_mask == 2)  // wide-bin contexts: u32 bins, feature-major (k_wide.hip)
    return reinterpret_cast<const uint32_t *>(fm)[(size_t)d.owner_local * N + id] <= d.thr_id;
  if (use_mask) return (mask[p >> 5] >> (p & 31)) & 1u;
  // feature-major copy of the bins: one byte per document, 32 neighbouring
  // documents per 32-byte sector (the block-row layout costs a 64-byte sector per
  // document here)
  return fm[(size_t)d.owner_local * N + id] <= d.thr_id;
}

// multi-GPU: the owner of the winning feature publishes the go-left bits of the
// node's positions; everybody else contributes zeros to the bitwise-or/sum.
__global__ __launch_bounds__(256) void k_mask(
    const QrTreeState *__restrict__ ts, const uint8_t *__restrict__ fm,
    const uint32_t Nfm,
    const uint32_t *__restrict__ order0, const uint32_t *__restrict__ order1,
    uint32_t *__restrict__ mask, const uint32_t mask_words, const int wide) {
  QR_JITTER();
  const QrSplitDesc d = ts->desc;
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (wide && w == 0) {
    // wide-bin contexts: the owner also publishes the bits of the threshold VALUE (it alone
    // holds the feature's row); zeros elsewhere, so the sum all-reduce hands it to every rank
    mask[mask_words] = d.active && d.owner_local >= 0 ? __float_as_uint(ts->nodes[d.node].threshold) : 0u;
  }
  if (w >= mask_words) return;
  uint32_t bits = 0;
  if (d.active && d.owner_local >= 0) {
    const uint32_t n = d.end - d.begin;
    const uint32_t *order = d.src_buf == 0 ? order0 : order1;
    // eight positions at a time: their ids together, then the eight values their tests read
    // (id -> value, position after position, is 64 dependent round trips per thread)
    const uint8_t *col8 = fm + (size_t)d.owner_local * Nfm;
    const uint32_t *col32 = reinterpret_cast<const uint32_t *>(fm) + (size_t)d.owner_local * Nfm;
    for (uint32_t k0 = 0; k0 < 32 && w * 32 + k0 < n; k0 += 8) {
      uint32_t id[8], tv[8];
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t p = w * 32 + k0 + k;
        const uint32_t pc = p < n ? p : n - 1;  // (clamped: unconditional loads)
        id[k] = d.src_buf == 2 ? d.begin + pc : order[d.begin + pc];
      }
      if (wide) {
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) tv[k] = col32[id[k]];
      } else {
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) tv[k] = col8[id[k]];
      }
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k)
        if (w * 32 + k0 + k < n && tv[k] <= d.thr_id) bits |= 1u << (k0 + k);
    }
  }
  mask[w] = bits;
}

// after the mask all-reduce on a wide feature-sharded context: the split node's threshold
__global__ void k_thr_patch(QrTreeState *__restrict__ ts, const uint32_t *__restrict__ word) {
  QR_JITTER();
  if (threadIdx.x == 0 && blockIdx.x == 0 && ts->desc.active)
    ts->nodes[ts->desc.node].threshold = __uint_as_float(*word);
}

__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *sh) {
  v = wave_scan_u32(v);
  v = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

#define PART_PER_THREAD (QR_PART_SLICE / 256)

// Single-pass stable partition with a look-back chain over the slice workgroups.
// Workgroup w publishes ONE 8-byte granule {tag = epoch, value = #lefts in its
// slice} with a relaxed agent-scope store (the data is the flag: no fence needed,
// cdna_hip_programming.md guideline 16, form R2) and sums the granules of all
// predecessors, polling each until its tag matches.  Predecessors are always
// dispatched first and never wait on successors, so the chain cannot deadlock.
// The epoch increases with every split of the context's lifetime, so the granule
// array never needs clearing.
struct PartNode {  // what the partition needs to know about the node being split
  uint32_t begin, n, lcount;
  int32_t src_buf, dst_buf, small_is_left;
};

// `w` = this workgroup's slice of the node, `first` = global index of the node's
// first workgroup (granules are indexed globally: `first + w`)
__device__ __forceinline__ void partition_body(
    const PartNode d, const QrSplitDesc &gl, const uint32_t w, const uint32_t first,
    const u64 epoch, const uint8_t *__restrict__ fm, const uint32_t Nfm,
    uint32_t *__restrict__ order0, uint32_t *__restrict__ order1,
    const uint32_t *__restrict__ mask, const int use_mask, u64 *__restrict__ state,
    const double *__restrict__ lambda, double *__restrict__ part_ss) {
  __shared__ uint32_t sh[4];
  __shared__ uint32_t wave_off[4];
  __shared__ double shd[4], shs[4];
  const uint32_t n = d.n;
  const uint32_t base = w * QR_PART_SLICE;
  if (base >= n) return;
  const uint32_t *src = d.src_buf == 0 ? order0 : order1;
  uint32_t *dst = d.dst_buf == 0 ? order0 : order1;
  uint32_t ids[PART_PER_THREAD];
  bool fl[PART_PER_THREAD];
  uint32_t cnt = 0;
  // The thread's eight ids are requested together, then the eight words their tests read:
  // TWO round trips.  (Written as one loop -- id, its test, the next id -- the compiler kept the
  // order: sixteen dependent loads one after the other, each behind an `s_waitcnt vmcnt(0)`,
  // with two waves per SIMD to hide them: most of the partition's 5 - 12 us.)  Loads are
  // unconditional, of positions clamped into the node, so that no branch separates them.
  {
    const uint32_t p0 = base + threadIdx.x * PART_PER_THREAD;
    const uint32_t plast = n - 1;  // (base < n: the node is not empty)
    uint32_t pc[PART_PER_THREAD];
#pragma unroll
    for (uint32_t k = 0; k < PART_PER_THREAD; ++k) pc[k] = p0 + k < plast ? p0 + k : plast;
    if (d.src_buf == 2) {
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) ids[k] = d.begin + pc[k];
    } else {
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) ids[k] = src[d.begin + pc[k]];
    }
    uint32_t tv[PART_PER_THREAD];  // the value the test compares / the mask word
    if (use_mask == 0) {
      const uint8_t *col = fm + (size_t)gl.owner_local * Nfm;
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) tv[k] = col[ids[k]];
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) fl[k] = tv[k] <= gl.thr_id;
    } else if (use_mask == 2) {
      const uint32_t *col = reinterpret_cast<const uint32_t *>(fm) + (size_t)gl.owner_local * Nfm;
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) tv[k] = col[ids[k]];
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) fl[k] = tv[k] <= gl.thr_id;
    } else if (use_mask == 3) {
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) tv[k] = mask[ids[k] >> 5];
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) fl[k] = (tv[k] >> (ids[k] & 31)) & 1u;
    } else {
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) tv[k] = mask[pc[k] >> 5];
#pragma unroll
      for (uint32_t k = 0; k < PART_PER_THREAD; ++k) fl[k] = (tv[k] >> (pc[k] & 31)) & 1u;
    }
#pragma unroll
    for (uint32_t k = 0; k < PART_PER_THREAD; ++k) {
      fl[k] = fl[k] && p0 + k < n;
      if (!(p0 + k < n)) ids[k] = 0;
      cnt += fl[k] ? 1u : 0u;
    }
  }
  // intra-workgroup inclusive scan of cnt
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_scan_u32(cnt);
  if (lane == 63) wave_off[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int x = 0; x < wave; ++x) woff += wave_off[x];
  const uint32_t total = wave_off[0] + wave_off[1] + wave_off[2] + wave_off[3];
  // publish my count, then look back over the node's earlier slices
  if (threadIdx.x == 0)
    __hip_atomic_store(&state[first + w], (epoch << 32) | (u64)total, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  uint32_t pre = 0;
  for (uint32_t v = threadIdx.x; v < w; v += 256) {
    u64 g;
    do {
      g = __hip_atomic_load(&state[first + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((g >> 32) != epoch) __builtin_amdgcn_s_sleep(1);
    } while ((g >> 32) != epoch);
    pre += (uint32_t)g;
  }
  const uint32_t left_before = block_sum_u32(pre, sh);
  uint32_t lpos = left_before + woff + inc - cnt;  // lefts before my first doc
  const uint32_t first_p = base + threadIdx.x * PART_PER_THREAD;
  double sq = 0.0, sm = 0.0;
  // (wide-bin contexts take the directly built child's sums here: the eight pseudo-responses
  // are requested together -- out-of-range slots hold id 0 -- and added in position order)
  double lam[PART_PER_THREAD];
  if (part_ss) {
#pragma unroll
    for (uint32_t k = 0; k < PART_PER_THREAD; ++k) lam[k] = lambda[ids[k]];
  }
#pragma unroll
  for (uint32_t k = 0; k < PART_PER_THREAD; ++k) {
    const uint32_t p = first_p + k;
    if (p < n) {
      uint32_t o;
      if (fl[k]) {
        o = d.begin + lpos;
        ++lpos;
      } else {
        o = d.begin + d.lcount + (p - lpos);
      }
      if (QR_DBG_OK(o >= d.begin && o < d.begin + n && ids[k] < Nfm, 1, o, ((u64)d.begin << 32) | n)) dst[o] = ids[k];
      if (part_ss && fl[k] == (d.small_is_left != 0)) {
        const double l = lam[k];
        sq += l * l;
        sm += l;
      }
    }
  }
  if (!part_ss) return;  // level-wise growth keeps no per-node sums (ot.cc:141-149)
  sq = wave_sum(sq);
  sm = wave_sum(sm);
  __syncthreads();
  if (lane == 0) {
    shd[wave] = sq;
    shs[wave] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part_ss[2 * (size_t)(first + w)] = (shd[0] + shd[1]) + (shd[2] + shd[3]);
    part_ss[2 * (size_t)(first + w) + 1] = (shs[0] + shs[1]) + (shs[2] + shs[3]);
  }
}

__global__ __launch_bounds__(256) void k_partition(
    const QrTreeState *__restrict__ ts, const uint8_t *__restrict__ fm,
    const uint32_t Nfm,
    uint32_t *__restrict__ order0, uint32_t *__restrict__ order1,
    const uint32_t *__restrict__ mask, const int use_mask,
    u64 *__restrict__ state, const double *__restrict__ lambda,
    double *__restrict__ part_ss, const int docmode) {
  QR_JITTER();
  const QrSplitDesc d = ts->desc;
  if (!d.active) return;
  PartNode pn;
  pn.begin = d.begin;
  pn.n = d.end - d.begin;
  pn.lcount = docmode ? ts->loc.lcount : d.lcount;
  pn.src_buf = d.src_buf;
  pn.dst_buf = d.dst_buf;
  pn.small_is_left = d.small_is_left;
  partition_body(pn, d, blockIdx.x, 0, ts->part_epoch, fm, Nfm, order0, order1, mask, use_mask,
                 state, lambda, part_ss);
}

// level-wise growth: every node of the level in one launch; the look-back chain
// of a node runs over its own workgroups only
__global__ __launch_bounds__(256) void k_partition_level(
    const QrTreeState *__restrict__ ts, const uint32_t *__restrict__ map,
    const uint8_t *__restrict__ fm, const uint32_t Nfm, uint32_t *__restrict__ order0,
    uint32_t *__restrict__ order1, u64 *__restrict__ state, const int wide,
    const uint32_t *__restrict__ docmask) {
  QR_JITTER();
  if (ts->obl_done || blockIdx.x >= ts->l_part_wgs) return;
  const QrLevelNode &ln = ts->lnode[map[blockIdx.x]];
  if (!ln.active) return;
  PartNode pn;
  pn.begin = ln.begin;
  pn.n = ln.end - ln.begin;
  pn.lcount = ln.lcount;
  pn.src_buf = ln.src_buf;
  pn.dst_buf = ln.dst_buf;
  pn.small_is_left = ln.small_is_left;
  QrSplitDesc gl;
  gl.owner_local = ts->l_owner_local;
  gl.thr_id = ts->obl_t;
  partition_body(pn, gl, blockIdx.x - ln.part_first, ln.part_first, ts->part_epoch, fm, Nfm,
                 order0, order1, docmask, docmask ? 3 : (wide ? 2 : 0), state, nullptr, nullptr);
}

// batched leaf-wise growth: every node of the batch has its own (feature, slot) and
// needs the sums of its directly built child
__global__ __launch_bounds__(256) void k_partition_batch(
    const QrTreeState *__restrict__ ts, const QrPartWg *__restrict__ wgs,
    const uint8_t *__restrict__ fm, const uint32_t Nfm, uint32_t *__restrict__ order0,
    uint32_t *__restrict__ order1, u64 *__restrict__ state, const double *__restrict__ lambda,
    double *__restrict__ part_ss, const uint32_t epoch_host, const int wide) {
  QR_JITTER();
  // (epoch_host != 0: the granules' tag is counted on the host, as for k_decide_part;
  // wide: `fm` holds u32 bins, k_wide.hip)
  const QrPartWg d = wgs[blockIdx.x];
  if (d.n == 0) return;
  PartNode pn;
  pn.begin = d.begin;
  pn.n = d.n;
  pn.lcount = d.lcount;
  pn.src_buf = d.src_buf;
  pn.dst_buf = d.dst_buf;
  pn.small_is_left = d.small_is_left;
  QrSplitDesc gl;
  gl.owner_local = d.owner_local;
  gl.thr_id = d.thr_id;
  partition_body(pn, gl, d.w, d.first, epoch_host ? (u64)epoch_host : (u64)ts->part_epoch, fm, Nfm, order0,
                 order1, nullptr, wide ? 2 : 0, state, lambda, part_ss);
}

// The control step and the partition in one launch (batch_step<true>): every
// workgroup decides for itself, the last one publishes, the others partition.
static_assert(128 * QR_BATCH == 256, "k_decide_part partitions with 256 threads per workgroup");
template <int CAP>
__global__ __launch_bounds__(128 * QR_BATCH) void k_decide_part(
    const QrTreeState *tin, QrTreeState *tout, QrTreeState *tlog2, const uint32_t epoch,
    const int root_mode, const int nleaves_arg, const u64 minls_arg, const int stage_nodes,
    const uint32_t N, const int flocal, const QrScalars *__restrict__ scal,
    const double *__restrict__ part_ss_in, const qr_split_t *__restrict__ featrec,
    const float *__restrict__ featthr, const uint32_t F, const int root_buf, const int G,
    const QrBlock *__restrict__ blocks, const int nblocks, QrHistWg *__restrict__ hist_wg,
    const uint32_t hist_grid, QrPlan *__restrict__ plans, QrScanWg *__restrict__ scan_wg,
    const uint8_t *__restrict__ fm, const uint32_t Nfm, uint32_t *__restrict__ order0,
    uint32_t *__restrict__ order1, u64 *__restrict__ state, const double *__restrict__ lambda,
    double *__restrict__ part_ss_out, const int wide, const uint32_t *__restrict__ hcnt_loc,
    const u64 Nglobal, const u64 spec_docs) {
  QR_JITTER();
  // (hcnt_loc != null: a document-sharded rank -- N its own documents, Nglobal everybody's)
  __shared__ QrLevelNode sh_next[QR_BATCH];
  __shared__ uint32_t sh_pw0[QR_BATCH + 1], sh_epoch;
  __shared__ int sh_nj;
#ifdef QR_STEP_TIMING
  const long long tq0 = clock64();
#endif
  // the LAST workgroup publishes: the grid is sized for the worst case plus one, so it
  // never has a slice to partition, and nobody's look-back chain waits for it
  batch_step<true, CAP>(tin, tout, tlog2, blockIdx.x == gridDim.x - 1, epoch, sh_next, sh_pw0, &sh_nj, &sh_epoch,
                   root_mode, nleaves_arg, minls_arg, stage_nodes, N, flocal, scal, part_ss_in, featrec,
                   featthr, F, root_buf, G, blocks, nblocks, hist_wg, hist_grid, nullptr, 0u, plans,
                   scan_wg, 0, hcnt_loc, Nglobal, spec_docs);
  __syncthreads();  // (the writer comes back later than the others; sh_* are final for all)
  QR_JITTER();      // (stress builds: between the control step and the workgroup's slice, too)
  const int nj = sh_nj;
#ifdef QR_STEP_TIMING
  const long long tq1 = clock64();
  if (threadIdx.x == 0 && (blockIdx.x == gridDim.x - 1))
    printf("decide_part writer: control+publish %lld cycles, nj %d part wgs %u\n", tq1 - tq0, nj, sh_pw0[nj]);
#endif
  if (blockIdx.x >= sh_pw0[nj]) return;
  int j = 0;
  while (j + 1 < nj && blockIdx.x >= sh_pw0[j + 1]) ++j;
  const QrLevelNode &ln = sh_next[j];
  PartNode pn;
  pn.begin = ln.begin;
  pn.n = ln.end - ln.begin;
  pn.lcount = ln.lcount;
  pn.src_buf = ln.src_buf;
  pn.dst_buf = ln.dst_buf;
  pn.small_is_left = ln.small_is_left;
  QrSplitDesc gl;
  gl.owner_local = ln.owner_local;
  gl.thr_id = ln.thr_id;
  partition_body(pn, gl, blockIdx.x - sh_pw0[j], sh_pw0[j], (u64)sh_epoch, fm, Nfm, order0, order1,
                 nullptr, wide ? 2 : 0, state, lambda, part_ss_out);
#ifdef QR_STEP_TIMING
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x + 1 == sh_pw0[nj]))
    printf("decide_part wg %u of %u: control %lld partition %lld cycles (node n %u)\n", blockIdx.x, sh_pw0[nj],
           tq1 - tq0, clock64() - tq1, pn.n);
#endif
}

// ===========================================================================
// Tree end: leaves, leaf outputs, score update
// ===========================================================================
__global__ __launch_bounds__(256) void k_finish(QrTreeState *__restrict__ ts) {
  QR_JITTER();
  // RTNode::save_leaves (rtnode.cc:34-46): DFS, left first.  The links are staged
  // in LDS by the whole workgroup; one lane walks them there.
  __shared__ int32_t s_feat[QR_MAXNODES], s_left[QR_MAXNODES], s_right[QR_MAXNODES];
  __shared__ int32_t s_leaf[QR_MAXNODES];  // DFS order -> node
  __shared__ int32_t stack[QR_MAXNODES];
  __shared__ int s_nl;
  const int nn = ts->nnodes;
  for (int i = threadIdx.x; i < nn; i += blockDim.x) {
    s_feat[i] = ts->nodes[i].feature;
    s_left[i] = ts->nodes[i].left;
    s_right[i] = ts->nodes[i].right;
  }
  __syncthreads();
  // A complete tree in level order (every oblivious tree: nodes 0 .. first - 1 internal, the
  // rest leaves of one depth): the leaves' left-first DFS order is their index order -- no
  // walk (one lane's walk over 127 nodes took 18 us of an Oblivious-LambdaMART iteration).
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  const int first = (nn - 1) / 2;
  {
    bool bad = (nn & 1) == 0 || ((nn + 1) & nn) != 0;  // nn = 2^k - 1
    for (int i = threadIdx.x; i < nn; i += blockDim.x)
      bad = bad || ((s_feat[i] >= 0) != (i < first)) ||
            (i < first && (s_left[i] != 2 * i + 1 || s_right[i] != 2 * i + 2));
    if (bad) s_bad = 1;
  }
  __syncthreads();
  if (!s_bad) {
    for (int l = threadIdx.x; l < nn - first; l += blockDim.x) s_leaf[l] = first + l;
    if (threadIdx.x == 0) s_nl = nn - first;
  } else if (threadIdx.x == 0) {
    int sp = 0, nl = 0;
    stack[sp++] = 0;
    while (sp > 0) {
      const int n = stack[--sp];
      if (s_feat[n] < 0) {
        s_leaf[nl++] = n;
      } else {
        stack[sp++] = s_right[n];
        stack[sp++] = s_left[n];
      }
    }
    s_nl = nl;
  }
  __syncthreads();
  const int nl = s_nl;
  for (int l = threadIdx.x; l < nl; l += blockDim.x) {
    const int n = s_leaf[l];
    ts->nodes[n].leaf_id = l;
    ts->leaf_nodes[l] = n;
    ts->leaf_begin[l] = ts->nodes[n].begin;  // DFS order == ascending positions
  }
  if (threadIdx.x == 0) {
    ts->nleaves = nl;
    ts->leaf_begin[nl] = ts->nodes[0].end;
  }
}

__device__ __forceinline__ int leaf_of_pos(const uint32_t *lb, int nl, uint32_t p) {
  int lo = 0, hi = nl;  // last l with lb[l] <= p
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (lb[mid] <= p)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// partial sums of (lambda, weight) per (slice, leaf) pair, entry = slice + leaf
__global__ __launch_bounds__(256) void k_leaf_sums(
    const QrTreeState *__restrict__ ts, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, const double *__restrict__ lambda,
    const double *__restrict__ weight, double *__restrict__ leafpart) {
  QR_JITTER();
  __shared__ double sh1[4], sh2[4];
  __shared__ uint32_t lb[QR_MAXNODES + 1];
  __shared__ int8_t lbuf[QR_MAXNODES];  // the list buffer every leaf's segment lives in
  if (ts->incomplete) return;  // (the last batch is not applied yet: its children's lists are not there)
  const int nl = ts->nleaves;
  // the leaves' bounds and buffers once per workgroup, a leaf per thread; then the thread's
  // four ids together and their values together.  (A position at a time -- leaf -> its node ->
  // the node's buffer -> id -> values -- was sixteen dependent round trips per thread: 20 us of
  // an oblivious iteration at depth 6.)  Same values, same additions.
  for (int i = threadIdx.x; i <= nl; i += 256) {
    lb[i] = ts->leaf_begin[i];
    if (i < nl) lbuf[i] = (int8_t)ts->nodes[ts->leaf_nodes[i]].buf;
  }
  __syncthreads();
  const uint32_t N = lb[nl];
  const uint32_t base = blockIdx.x * QR_SLICE;
  if (base >= N) return;
  const uint32_t end = base + QR_SLICE < N ? base + QR_SLICE : N;
  constexpr int PT = QR_SLICE / 256;
  double v1[PT], v2[PT];
  int lf[PT];
  uint32_t id[PT];
#pragma unroll
  for (int k = 0; k < PT; ++k) {  // (unconditional loads of positions clamped into the slice)
    const uint32_t p = base + k * 256 + threadIdx.x;
    const uint32_t pc = p < end ? p : end - 1;
    const int l = leaf_of_pos(lb, nl, pc);
    const int buf = lbuf[l];
    lf[k] = p < end ? l : -1;
    id[k] = buf == 2 ? pc : (buf == 0 ? order0[pc] : order1[pc]);
  }
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    v1[k] = lambda[id[k]];
    v2[k] = weight ? weight[id[k]] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < PT; ++k)
    if (lf[k] < 0) v1[k] = v2[k] = 0.0;
  const int l0 = leaf_of_pos(lb, nl, base);
  for (int l = l0; l < nl && lb[l] < end; ++l) {
    double a = 0.0, b = 0.0;
    for (uint32_t k = 0; k < QR_SLICE / 256; ++k)
      if (lf[k] == l) {
        a += v1[k];
        b += v2[k];
      }
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      sh1[threadIdx.x >> 6] = a;
      sh2[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const size_t e = (size_t)blockIdx.x + l;
      leafpart[2 * e] = (sh1[0] + sh1[1]) + (sh1[2] + sh1[3]);
      leafpart[2 * e + 1] = (sh2[0] + sh2[1]) + (sh2[2] + sh2[3]);
    }
  }
}

// ---- small trees (at most QR_LDOC leaves): the same sums in DOCUMENT order -----------------
// k_leaf_sums reaches lambda[id] / weight[id] through the leaves' document lists: one 64-byte
// sector per 8-byte value once the leaves interleave (16 us per 1M documents, 96 us per 8M).
// Here every document finds its leaf itself -- by walking the tree on the feature-major bins
// (WALK: every feature is on this rank), or from the byte k_leaf_ids_scatter wrote through
// the lists (feature-sharded ranks) -- and lambda / weight stream in coalesced.  The leaf of
// every document is kept in `leafb` for the score update (k_score_update_leaf).  A workgroup
// takes QR_SLICE consecutive documents, four per thread; per leaf: the thread's members in
// document order, the lanes by wave_sum, the four waves pairwise -- a fixed order that does
// not depend on how the leaf was found, so every layout ends with the same bits.
// part: [leaf][slice][2].  present: --subsample's mask (the sums are the sample's).
#define QR_LDOC 16
struct QrNodesOut;
__device__ __forceinline__ void nodes_out_publish(QrNodesOut *out, const long long seq);
__device__ __forceinline__ void nodes_out_write(const QrTreeState *ts, QrNodesOut *out, const long long seq);

template <bool WALK>
__global__ __launch_bounds__(256) void k_leaf_sums_doc(
    const QrTreeState *__restrict__ ts, const uint8_t *__restrict__ fm, const uint32_t N,
    const int32_t *__restrict__ gf2lf, const int wide, uint8_t *__restrict__ leafb,
    const double *__restrict__ lambda, const double *__restrict__ weight,
    const uint8_t *__restrict__ present, double *__restrict__ part) {
  QR_JITTER();
  constexpr int NN = 2 * QR_LDOC;  // nodes of a tree of QR_LDOC leaves (2 L - 1)
  // s_rec[n]: what a step of the walk needs of node n in ONE LDS word -- bit 31 = leaf;
  // a leaf: its DFS index; an internal node: left | right << 8 | index of its test << 16
  __shared__ uint32_t s_rec[NN];
  __shared__ int32_t s_ilf[QR_LDOC], s_ithr[QR_LDOC];  // the internal nodes' tests, compacted
  __shared__ int s_ni;
  __shared__ double sh1[QR_LDOC][16], sh2[QR_LDOC][16];
#ifdef QR_LEAF_TIMING
  long long tq[8]; tq[0] = clock64();
#define LT(i) tq[i] = clock64()
#else
#define LT(i)
#endif
  // (everything the staging needs is requested before any of it is looked at)
  const int incomplete = ts->incomplete;
  const int nl = ts->nleaves;
  const int nn_all = ts->nnodes;
  int f = -1, thr = 0, left = 0, right = 0, leafid = 0;
  if (WALK && threadIdx.x < NN) {
    const QrNode &nd = ts->nodes[threadIdx.x];  // (inside the array whatever nnodes says)
    f = nd.feature;
    thr = nd.thr_id;
    left = nd.left;
    right = nd.right;
    leafid = nd.leaf_id;
  }
  if (incomplete) return;  // (the host carries the tree on and enqueues the leaf kernels again)
  const int nn = nn_all < NN ? nn_all : NN;
  if (WALK && threadIdx.x < 64) {  // one wave: records + compaction of the tests by ballot
    const int i = threadIdx.x;
    const int lfv = i < nn && f >= 0 ? gf2lf[f] : -1;
    const unsigned long long m = __ballot(lfv >= 0);
    const int idx = __popcll(m & ((1ull << i) - 1ull));
    const int ni = __popcll(m) < QR_LDOC ? __popcll(m) : QR_LDOC;
    const int lf_first = m ? __shfl(lfv, __ffsll((long long)m) - 1) : 0;
    if (i < nn)
      s_rec[i] = lfv >= 0 ? ((uint32_t)left | ((uint32_t)right << 8) | ((uint32_t)idx << 16))
                          : (0x80000000u | (uint32_t)leafid);
    if (lfv >= 0 && idx < QR_LDOC) {
      s_ilf[idx] = lfv;
      s_ithr[idx] = thr;
    }
    if (i >= ni && i < QR_LDOC) {  // (padding: repeats a real test, never looked at)
      s_ilf[i] = lf_first;
      s_ithr[i] = 0;
    }
    if (i == 0) s_ni = ni;
  }
  const uint32_t d0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  // the values first: their loads ride under the tests
  double v1[4], v2[4];
  bool in[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t d = d0 + k < N ? d0 + k : N - 1;
    in[k] = d0 + k < N && (!present || present[d] != 0);
    v1[k] = lambda[d];
    v2[k] = weight ? weight[d] : 0.0;
  }
  __syncthreads();
  LT(1);
  int l[4];
  if (WALK) {
    // EVERY test of the tree on the thread's four documents at once -- at most QR_LDOC - 1
    // independent loads (one dword = the four documents' bins of a feature) -- instead of a
    // walk whose every level waits for a load that depends on the level before (a chain tree
    // of 10 leaves is 9 such round trips: 6 of the 10 us a 1M-document walk took).  Bit j of
    // go[k]: document k goes left at the j-th internal node; the walk then runs on LDS.
    uint32_t go[4] = {0, 0, 0, 0};
    const bool quad = (N & 3u) == 0;  // (uniform) the rows start on dword boundaries
    if (s_ni > 0) {
      if (!wide) {
        uint32_t w4[QR_LDOC - 1];
        if (quad) {
          const uint32_t dq = d0 < N ? d0 : N - 4;  // (unconditional loads: clamped)
#pragma unroll
          for (int j = 0; j < QR_LDOC - 1; ++j)
            w4[j] = *reinterpret_cast<const uint32_t *>(fm + (size_t)s_ilf[j] * N + dq);
        } else {
#pragma unroll
          for (int j = 0; j < QR_LDOC - 1; ++j) w4[j] = 0;
#pragma unroll 1
          for (int k = 0; k < 4; ++k) {  // (rare: N not a multiple of 4 -- a document at a time)
            const uint32_t d = d0 + k < N ? d0 + k : N - 1;
#pragma unroll
            for (int j = 0; j < QR_LDOC - 1; ++j) w4[j] |= (uint32_t)fm[(size_t)s_ilf[j] * N + d] << (8 * k);
          }
        }
#pragma unroll
        for (int j = 0; j < QR_LDOC - 1; ++j) {
          const uint32_t t = (uint32_t)s_ithr[j];
#pragma unroll
          for (int k = 0; k < 4; ++k) go[k] |= (((w4[j] >> (8 * k)) & 0xffu) <= t ? 1u : 0u) << j;
        }
      } else {
        const uint32_t *fw = reinterpret_cast<const uint32_t *>(fm);
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {  // (u32 bins: a document at a time, its tests together)
          const uint32_t d = d0 + k < N ? d0 + k : N - 1;
          uint32_t g = 0;
#pragma unroll
          for (int j = 0; j < QR_LDOC - 1; ++j)
            g |= (fw[(size_t)s_ilf[j] * N + d] <= (uint32_t)s_ithr[j] ? 1u : 0u) << j;
          go[k] = g;
        }
      }
    }
    LT(2);
    uint32_t rec[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) rec[k] = s_rec[0];
    for (bool any = !(rec[0] >> 31); any;) {
      any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (!(rec[k] >> 31)) {
          rec[k] = s_rec[(go[k] >> ((rec[k] >> 16) & 0xffu)) & 1u ? rec[k] & 0xffu : (rec[k] >> 8) & 0xffu];
          any |= !(rec[k] >> 31);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] = (int)(rec[k] & 0xffu);
    if (d0 + 3 < N)
      *reinterpret_cast<uint32_t *>(leafb + d0) =
          (uint32_t)l[0] | ((uint32_t)l[1] << 8) | ((uint32_t)l[2] << 16) | ((uint32_t)l[3] << 24);
    else
      for (int k = 0; k < 4 && d0 + k < N; ++k) leafb[d0 + k] = (uint8_t)l[k];
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] = d0 + k < N ? (int)leafb[d0 + k] : -1;
  }
  LT(3);
  // per leaf: the thread's members in document order, the 16 lanes of a DPP row by butterflies
  // (every lane of the row ends with the row's total), the 16 rows of the workgroup by thread
  // `leaf` in row order -- fixed, and cheaper than a full wave_sum per (leaf, value): its eight
  // readlanes and scalar adds were a third of this kernel
  const int lane = threadIdx.x & 63;
  const int row = threadIdx.x >> 4;  // 16 rows of 16 lanes
  for (int t = 0; t < nl; ++t) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (in[k] && l[k] == t) {
        a += v1[k];
        b += v2[k];
      }
    a += dpp_f64<0xB1>(a);   // quad_perm [1,0,3,2]
    b += dpp_f64<0xB1>(b);
    a += dpp_f64<0x4E>(a);   // quad_perm [2,3,0,1]
    b += dpp_f64<0x4E>(b);
    a += dpp_f64<0x141>(a);  // row_half_mirror
    b += dpp_f64<0x141>(b);
    a += dpp_f64<0x140>(a);  // row_mirror
    b += dpp_f64<0x140>(b);
    if ((lane & 15) == 0) {
      sh1[t][row] = a;
      sh2[t][row] = b;
    }
  }
  LT(4);
  __syncthreads();
  if ((int)threadIdx.x < nl) {  // part[leaf][slice][2]: k_leaf_final reads a leaf's slices in a row
    const int t = threadIdx.x;
    const size_t e = (size_t)t * gridDim.x + blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int r = 0; r < 16; r += 4) {  // a wave's four rows as wave_sum adds them, the waves in order
      a += (sh1[t][r] + sh1[t][r + 1]) + (sh1[t][r + 2] + sh1[t][r + 3]);
      b += (sh2[t][r] + sh2[t][r + 1]) + (sh2[t][r + 2] + sh2[t][r + 3]);
    }
    part[2 * e] = a;
    part[2 * e + 1] = b;
  }
#ifdef QR_LEAF_TIMING
  LT(5);
  if (WALK && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 500))
    printf("leaf_sums_doc wg %u: stage %lld tests %lld walk %lld reduce %lld store %lld total %lld\n", blockIdx.x, tq[1]-tq[0], tq[2]-tq[1], tq[3]-tq[2], tq[4]-tq[3], tq[5]-tq[4], tq[5]-tq[0]);
#endif
}

// feature-sharded ranks: the leaf of every listed document, written through the lists
__global__ __launch_bounds__(256) void k_leaf_ids_scatter(
    const QrTreeState *__restrict__ ts, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, uint8_t *__restrict__ leafb) {
  QR_JITTER();
  __shared__ uint32_t lb[QR_LDOC + 1];
  __shared__ int lbuf[QR_LDOC];
  if (ts->incomplete) return;
  const int nl = ts->nleaves;
  if ((int)threadIdx.x <= nl) lb[threadIdx.x] = ts->leaf_begin[threadIdx.x];
  if ((int)threadIdx.x < nl) lbuf[threadIdx.x] = ts->nodes[ts->leaf_nodes[threadIdx.x]].buf;
  __syncthreads();
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= lb[nl]) return;
  const int l = leaf_of_pos(lb, nl, p);
  const int buf = lbuf[l];
  leafb[buf == 2 ? p : (buf == 0 ? order0[p] : order1[p])] = (uint8_t)l;
}

// mart.cc:459-468 from the leaf bytes: scores[d] += shrinkage * leaf_value[leaf(d)], four
// documents per thread (f64 multiply, then add; no contraction)
__global__ __launch_bounds__(256) void k_score_update_leaf(
    const QrTreeState *__restrict__ ts, const uint8_t *__restrict__ leafb, const uint32_t N,
    const double shrinkage, double *__restrict__ scores) {
  QR_JITTER();
  __shared__ double lv[QR_LDOC];
  // (everything is requested at once -- the tree's flag, the leaf values whatever their number,
  // the thread's leaf bytes and scores: one round trip, where flag -> values -> barrier -> data
  // were three)
  const int incomplete = ts->incomplete;
  const double my_lv = threadIdx.x < QR_LDOC ? ts->leaf_value[threadIdx.x] : 0.0;
  const uint32_t d0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool full = d0 + 3 < N;
  uint32_t lw = 0;
  double2 u = make_double2(0.0, 0.0), v = u;
  double2 *sp = reinterpret_cast<double2 *>(scores + (full ? d0 : 0));
  if (full) {
    lw = *reinterpret_cast<const uint32_t *>(leafb + d0);
    u = sp[0];
    v = sp[1];
  }
  if (incomplete) return;  // the host carries the tree on and enqueues this update again
  if (threadIdx.x < QR_LDOC) lv[threadIdx.x] = my_lv;
  __syncthreads();
  if (d0 >= N) return;
  if (full) {
    u.x = u.x + shrinkage * lv[lw & 0xffu];
    u.y = u.y + shrinkage * lv[(lw >> 8) & 0xffu];
    v.x = v.x + shrinkage * lv[(lw >> 16) & 0xffu];
    v.y = v.y + shrinkage * lv[lw >> 24];
    sp[0] = u;
    sp[1] = v;
  } else {
    for (int k = 0; k < 4 && d0 + k < N; ++k) scores[d0 + k] = scores[d0 + k] + shrinkage * lv[leafb[d0 + k]];
  }
}

// compact records of the finished tree (what crosses the C-ABI), written by the
// whole workgroup once the leaf values are in place
// `seq` (pad[2]) is stored LAST, behind a system-scope fence: the host polls it in the
// pinned block instead of waiting for an event (see wait_seq_impl in qr_api.hip).
__device__ __forceinline__ void nodes_out_publish(QrNodesOut *out, const long long seq) {
  __threadfence_system();
  __hip_atomic_store(&out->pad[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void nodes_out_write(const QrTreeState *ts, QrNodesOut *out, const long long seq) {
  const int nn = ts->nnodes;
  if (threadIdx.x == 0) {  // (the header with the records: ONE system-scope fence before the number)
    const int64_t inc = ts->incomplete, steps = ts->real_steps;
    out->nnodes = nn;
    out->pad[0] = inc;
    out->pad[1] = steps;
    out->pad[3] = (int64_t)((uint64_t)seq * 0x9E3779B97F4A7C15ull ^ ((uint64_t)nn << 40) ^ ((uint64_t)inc << 32) ^ (uint64_t)steps);
  }
  QrNodeWire *wire = reinterpret_cast<QrNodeWire *>(out->nodes);
  for (int i = threadIdx.x; i < nn; i += blockDim.x) {
    const QrNode &s = ts->nodes[i];
    QrNodeWire d;
    d.feature = s.feature;
    d.thr_id = s.thr_id;
    d.threshold = s.threshold;
    d.left = s.left;
    d.right = s.right;
    d.tag = 0;
    d.value = s.value;
    d.deviance = s.deviance;
    d.nsamples = s.count;
    d.tag = qr_node_tag(d, (uint64_t)seq);  // (qr_tree_nodes checks it: a self-validating record)
    wire[i] = d;
  }
#ifdef QR_DEBUG_CHECKS
  //  the two places a leaf's output lives -- the record that crosses to the host and the
  // per-leaf table the lazy score update adds from -- must hold the same bits
  for (int l = threadIdx.x; l < ts->nleaves; l += blockDim.x) {
    const unsigned long long a = (unsigned long long)__double_as_longlong(ts->nodes[ts->leaf_nodes[l]].value);
    const unsigned long long b = (unsigned long long)__double_as_longlong(ts->leaf_value[l]);
    (void)QR_DBG_OK(a == b, 9, a, b);
  }
#endif
  __threadfence_system();
  __syncthreads();
  //  the number goes out behind a fence of the thread that stores it, as a release:
  // every thread's records were fenced to system scope before the barrier, and the storing thread
  // orders its own view of them before the word the host polls
  if (threadIdx.x == 0) nodes_out_publish(out, seq);
}

// rt.cc:165-207
__global__ __launch_bounds__(1024) void k_leaf_final(QrTreeState *__restrict__ ts,
                                                     const double *__restrict__ leafpart,
                                                     const int newton, const int docmode,
                                                     long long *__restrict__ xleaf,
                                                     const int rank, const int world,
                                                     const int stride,
                                                     QrNodesOut *__restrict__ nodes_out,
                                                     const long long seq, const uint32_t dense_slices) {
  QR_JITTER();
  // the leaves' bounds and nodes first, by all threads: a tree of 64 leaves is four rounds of
  // the loop below per wave, and a round that fetches its bounds itself waits for them before it
  // can ask for its partials (15 us for an oblivious tree of depth 6).  A thread per entry of the
  // two arrays, whatever the number of leaves turns out to be (entries beyond it are never
  // looked at): the requests leave with the header's, not behind it.
  static_assert(QR_MAXNODES == 1024, "one thread per leaf entry");
  const int incomplete = ts->incomplete;
  const int nl = ts->nleaves;
  const uint32_t my_lb = ts->leaf_begin[threadIdx.x];
  const int32_t my_ln = ts->leaf_nodes[threadIdx.x];
  const uint32_t last_lb = threadIdx.x == 0 ? ts->leaf_begin[QR_MAXNODES] : 0u;
  if (incomplete) {  // tell the host, which carries the tree on (qr_k_tree_continue)
    if (threadIdx.x == 0) {
      nodes_out->pad[0] = 1;
      nodes_out->pad[1] = ts->real_steps;
      nodes_out_publish(nodes_out, seq);
    }
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ uint32_t sh_lb[QR_MAXNODES + 1];
  __shared__ int32_t sh_ln[QR_MAXNODES];
  sh_lb[threadIdx.x] = my_lb;
  sh_ln[threadIdx.x] = my_ln;
  if (threadIdx.x == 0) sh_lb[QR_MAXNODES] = last_lb;
  if (docmode)  // own slot <- local sums, zeros elsewhere (all-reduce == all-gather)
    for (int i = threadIdx.x; i < world * stride; i += 1024) xleaf[i] = 0;
  __syncthreads();
  for (int l = wave; l < nl; l += 16) {  // one wave per leaf, fixed reduction tree
    const uint32_t b = sh_lb[l], e = sh_lb[l + 1];
    double s1 = 0.0, s2 = 0.0;
    if (dense_slices) {  // k_leaf_sums_doc's [leaf][slice][2]: every slice may hold the leaf
      for (uint32_t s = lane; s < dense_slices; s += 8 * 64) {
        double2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t sk = s + 64u * k;
          v[k] = sk < dense_slices
                     ? *reinterpret_cast<const double2 *>(leafpart + 2 * ((size_t)l * dense_slices + sk))
                     : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (s + 64u * k < dense_slices) {
            s1 += v[k].x;
            s2 += v[k].y;
          }
      }
    } else if (e > b) {
      const uint32_t sl0 = b / QR_SLICE, sl1 = (e - 1) / QR_SLICE;
      // (eight rounds of loads in flight: 8M documents are 30 rounds per lane)
      for (uint32_t s = sl0 + lane; s <= sl1; s += 8 * 64) {
        double2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t sk = s + 64u * k;
          v[k] = sk <= sl1 ? *reinterpret_cast<const double2 *>(leafpart + 2 * ((size_t)sk + l)) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (s + 64u * k <= sl1) {
            s1 += v[k].x;
            s2 += v[k].y;
          }
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
      if (docmode) {
        xleaf[(size_t)rank * stride + 2 * l] = __double_as_longlong(s1);
        xleaf[(size_t)rank * stride + 2 * l + 1] = __double_as_longlong(s2);
        continue;
      }
      double v;
      if (newton)
        v = s2 >= 2.2204460492503131e-16 ? s1 / s2 : 0.0;  // DBL_EPSILON
      else
        v = s1 / (double)(e - b);
      ts->leaf_value[l] = v;
      ts->nodes[sh_ln[l]].value = v;
    }
  }
  if (docmode) return;  // k_leaf_global writes the records after the exchange
  __syncthreads();
  nodes_out_write(ts, nodes_out, seq);
}

// document-sharded: leaf outputs from the gathered per-rank sums, added in rank
// order (every rank ends with the same bits); rt.cc:165-207
__global__ __launch_bounds__(1024) void k_leaf_global(QrTreeState *__restrict__ ts,
                                                      const long long *__restrict__ xleaf,
                                                      const int newton, const int world,
                                                      const int stride,
                                                      QrNodesOut *__restrict__ nodes_out,
                                                      const long long seq) {
  QR_JITTER();
  // (batched growth whose enqueued steps did not suffice: the host carries the tree on and
  // enqueues the leaf kernels again, as on one GPU)
  if (ts->incomplete) return;
  const int nl = ts->nleaves;
  for (int l = threadIdx.x; l < nl; l += 1024) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < world; ++r) {
      s1 += __longlong_as_double(xleaf[(size_t)r * stride + 2 * l]);
      s2 += __longlong_as_double(xleaf[(size_t)r * stride + 2 * l + 1]);
    }
    const QrNode *nd = &ts->nodes[ts->leaf_nodes[l]];
    double v;
    if (newton)
      v = s2 >= 2.2204460492503131e-16 ? s1 / s2 : 0.0;  // DBL_EPSILON
    else
      v = s1 / (double)nd->count;
    ts->leaf_value[l] = v;
    ts->nodes[ts->leaf_nodes[l]].value = v;
  }
  __syncthreads();
  nodes_out_write(ts, nodes_out, seq);
}

// mart.cc:459-468 through the leaf membership instead of a tree walk:
// scores[i] += shrinkage * leaf(i)   (f64 multiply, then add; no contraction)
__global__ __launch_bounds__(256) void k_score_update(
    const QrTreeState *__restrict__ ts, const uint32_t *__restrict__ order0,
    const uint32_t *__restrict__ order1, const double shrinkage,
    double *__restrict__ scores) {
  QR_JITTER();
  __shared__ uint32_t lb[QR_MAXNODES + 1];
  __shared__ double lv[QR_MAXNODES];
  __shared__ int lbuf[QR_MAXNODES];
  const int nl = ts->nleaves;
  for (int i = threadIdx.x; i <= nl; i += 256) lb[i] = ts->leaf_begin[i];
  for (int i = threadIdx.x; i < nl; i += 256) {
    lv[i] = ts->leaf_value[i];
    lbuf[i] = ts->nodes[ts->leaf_nodes[i]].buf;
  }
  __syncthreads();
  const uint32_t N = lb[nl];
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const int l = leaf_of_pos(lb, nl, p);
  const int buf = lbuf[l];
  const uint32_t id = buf == 2 ? p : (buf == 0 ? order0[p] : order1[p]);
  const double add = shrinkage * lv[l];
  scores[id] = scores[id] + add;
}

// The same update in document order: walk the tree on the feature-major bins
// (x <= threshold  <=>  bin <= slot).  Scores are read and written coalesced, a
// level costs one byte per document out of a row that neighbouring documents share;
// the membership version above touches scores[id] through the leaves' document
// lists, i.e. one 32-byte sector per 8-byte score.  Needs every feature on this
// rank; also the update of a --subsample iteration, whose leaves hold the sample only.
__global__ __launch_bounds__(256) void k_score_update_walk(
    const QrTreeState *__restrict__ ts, const uint8_t *__restrict__ fm, const uint32_t N,
    const int32_t *__restrict__ gf2lf, const double shrinkage, double *__restrict__ scores,
    const int wide) {
  QR_JITTER();
  __shared__ int32_t s_lf[QR_MAXNODES], s_thr[QR_MAXNODES], s_left[QR_MAXNODES], s_right[QR_MAXNODES];
  __shared__ double s_val[QR_MAXNODES];
  if (ts->incomplete) return;  // the host carries the tree on and enqueues this update again
  const int nn = ts->nnodes;
  for (int i = threadIdx.x; i < nn; i += 256) {
    const int f = ts->nodes[i].feature;
    s_lf[i] = f >= 0 ? gf2lf[f] : -1;
    s_thr[i] = ts->nodes[i].thr_id;
    s_left[i] = ts->nodes[i].left;
    s_right[i] = ts->nodes[i].right;
    s_val[i] = ts->nodes[i].value;
  }
  // A complete tree in level order whose levels share one test each -- every oblivious tree
  // (ot.cc: one (feature, threshold) per level) -- needs no walk: the levels' bytes are
  // requested together (a dword = the thread's four documents) and the node index follows
  // from the tests by arithmetic.  (The walk below is a chain of dependent loads, one per
  // level: 12 us of a depth-6 iteration at 1M documents.)
  __shared__ int s_notobl;
  if (threadIdx.x == 0) s_notobl = 0;
  __syncthreads();
  const int first = (nn - 1) / 2;  // (internal nodes of a complete tree: 0 .. first - 1)
  {
    bool bad = wide || (N & 3u) != 0 || (nn & 1) == 0 || ((nn + 1) & nn) != 0 || nn > 1023;
    for (int i = threadIdx.x; i < nn && !bad; i += 256) {
      const int lv = 31 - __clz(i + 1);          // level of node i
      const int head = (1 << lv) - 1;            // first node of that level
      bad = ((s_lf[i] >= 0) != (i < first)) ||
            (i < first && (s_left[i] != 2 * i + 1 || s_right[i] != 2 * i + 2 || s_lf[i] != s_lf[head] ||
                           s_thr[i] != s_thr[head]));
    }
    if (bad) s_notobl = 1;
  }
  __syncthreads();
  // Four consecutive documents per thread: their walks are independent chains of dependent
  // one-byte loads, so four are in flight per lane instead of one (the launch is bound by the
  // latency of those chains: 90 us for 8M documents with one), and the scores move as two
  // 16-byte accesses.
  const uint32_t d0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (d0 >= N) return;
  int n[4] = {0, 0, 0, 0};
  const uint32_t *fw = reinterpret_cast<const uint32_t *>(fm);
  if (!s_notobl) {
    const int depth = 31 - __clz(nn + 1) - 1;  // nn = 2^(depth + 1) - 1, depth <= 9
    uint32_t w4[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) {
      const int head = l < depth ? (1 << l) - 1 : 0;
      w4[l] = l < depth ? *reinterpret_cast<const uint32_t *>(fm + (size_t)s_lf[head] * N + d0) : 0u;
    }
#pragma unroll
    for (int l = 0; l < 9; ++l) {
      if (l >= depth) break;
      const uint32_t t = (uint32_t)s_thr[(1 << l) - 1];
#pragma unroll
      for (int k = 0; k < 4; ++k) n[k] = 2 * n[k] + (((w4[l] >> (8 * k)) & 0xffu) <= t ? 1 : 2);
    }
  } else
  for (bool any = s_lf[0] >= 0; any;) {
    uint32_t b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lf = s_lf[n[k]];
      const uint32_t d = d0 + k < N ? d0 + k : N - 1;
      b[k] = lf < 0 ? 0u : (wide ? fw[(size_t)lf * N + d] : (uint32_t)fm[(size_t)lf * N + d]);
    }
    any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (s_lf[n[k]] >= 0) {
        n[k] = b[k] <= (uint32_t)s_thr[n[k]] ? s_left[n[k]] : s_right[n[k]];
        any |= s_lf[n[k]] >= 0;
      }
  }
  if (d0 + 3 < N && (reinterpret_cast<size_t>(scores) & 15) == 0) {
    double2 *sp = reinterpret_cast<double2 *>(scores + d0);
    double2 u = sp[0], v = sp[1];
    u.x = u.x + shrinkage * s_val[n[0]];
    u.y = u.y + shrinkage * s_val[n[1]];
    v.x = v.x + shrinkage * s_val[n[2]];
    v.y = v.y + shrinkage * s_val[n[3]];
    sp[0] = u;
    sp[1] = v;
  } else {
    for (int k = 0; k < 4 && d0 + k < N; ++k) scores[d0 + k] = scores[d0 + k] + shrinkage * s_val[n[k]];
  }
}

// mart.cc:447-457: validation scores by walking the tree on raw f32 rows
__global__ __launch_bounds__(256) void k_valid_update(
    const QrTreeState *__restrict__ ts, const float *__restrict__ raw,
    const uint32_t vN, const uint32_t F, const double shrinkage,
    double *__restrict__ vscores) {
  QR_JITTER();
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= vN || ts->incomplete) return;
  const float *x = raw + (size_t)i * F;
  int n = 0;
  while (ts->nodes[n].feature >= 0)
    n = x[ts->nodes[n].feature] <= ts->nodes[n].threshold ? ts->nodes[n].left
                                                           : ts->nodes[n].right;
  const double add = shrinkage * ts->nodes[n].value;
  vscores[i] = vscores[i] + add;
}


// ===========================================================================
// Oblivious (level-wise, symmetric) trees: ObliviousRT::fit, ot.cc:32-201
// ===========================================================================
// fill() + argmax of one level for one feature (ot.cc:177-201, 67-92): the gain
// of slot t is summed over the nodes of the level in node order; a slot where any
// node violates minls is invalid for the whole level; only sums > 0 compete and
// the first maximum wins.
// (returns false when the tree was finished at an earlier level: nothing to do)
__device__ __forceinline__ bool obl_fill_body(
    const QrTreeState *__restrict__ ts, const int level,
    const long long *__restrict__ hsum, const uint32_t *__restrict__ hcnt,
    const int flocal, const uint32_t *__restrict__ thr_size,
    const int32_t *__restrict__ lf2gf, const QrScalars *__restrict__ scal,
    qr_split_t *__restrict__ featrec) {
  __shared__ Best sh_b[4];
  if (ts->obl_done) return false;
  const int lf = blockIdx.x;
  const uint32_t t = threadIdx.x;
  const int gf = lf2gf[lf];
  const uint32_t tsize = thr_size[gf];
  const u64 minls = ts->minls;
  const double inv_scale = scal->inv_scale;
  const int lbegin = (1 << level) - 1, lend = (1 << (level + 1)) - 1;
  // the level's histogram slots first, a node per thread: read inside the loop below, a node's
  // slot is a dependent load in front of its cells (32 nodes: 28 us of round trips, one after
  // the other); then the cells of eight nodes are requested together.  Same additions in node
  // order.
  __shared__ int32_t sh_slot[QR_MAXLEVEL];
  const int nlev = lend - lbegin;
  for (int i = (int)t; i < nlev; i += (int)blockDim.x) sh_slot[i] = ts->nodes[lbegin + i].hslot;
  __syncthreads();
  double sum = 0.0;
  bool invalid = t >= tsize;
  for (int i0 = 0; i0 < nlev; i0 += 8) {
    long long cs[8], S[8];
    uint32_t lc32[8], C32[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k < nlev ? i0 + k : nlev - 1;  // (clamped: unconditional loads)
      const size_t base = ((size_t)sh_slot[i] * flocal + lf) * 256;
      cs[k] = hsum[base + t];
      S[k] = hsum[base + 255];
      lc32[k] = hcnt[base + t];
      C32[k] = hcnt[base + 255];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (i0 + k >= nlev) break;
      const u64 lc = lc32[k], C = C32[k];
      const u64 rc = C - lc;
      if (lc >= minls && rc >= minls) {
        const double s = (double)S[k] * inv_scale;
        const double lsum = (double)cs[k] * inv_scale;
        const double rsum = s - lsum;
        sum += lsum * lsum / (double)lc + rsum * rsum / (double)rc;
      } else
        invalid = true;
    }
  }
  Best v;
  v.score = -1.0;
  v.t = 0xFFFFFFFFu;
  if (!invalid && sum > 0.0) {  // NaN fails the comparison, as in ot.cc:77-78
    v.score = sum;
    v.t = t;
  }
  v = block_best(v, sh_b);
  if (t == 0) {
    qr_split_t *o = &featrec[lf];
    o->score = v.score;
    o->feature = v.t == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)gf;
    o->thr_id = v.t;
    o->lcount = o->rcount = 0;
  }
  return true;
}
__global__ __launch_bounds__(256) void k_obl_fill(
    const QrTreeState *__restrict__ ts, const int level,
    const long long *__restrict__ hsum, const uint32_t *__restrict__ hcnt,
    const int flocal, const uint32_t *__restrict__ thr_size,
    const int32_t *__restrict__ lf2gf, const QrScalars *__restrict__ scal,
    qr_split_t *__restrict__ featrec) {
  QR_JITTER();
  (void)obl_fill_body(ts, level, hsum, hcnt, flocal, thr_size, lf2gf, scal, featrec);
}

// choose the level's (feature, slot): first maximum over features (ot.cc:84-95).
// Run by the first wave of k_obl_plan; the choice goes out through `pick` (LDS) for
// the rest of the workgroup: {done, feature, slot}.
// first maximum over a list of records: highest score, equal scores -> lowest feature
// (the list is the rank's per-feature records, or -- feature-sharded -- one record per rank)
__device__ __forceinline__ qr_split_t obl_pick(const qr_split_t *__restrict__ recs, const int n,
                                               const int stride) {
  const int lane = threadIdx.x & 63;
  qr_split_t best;
  best.score = -1.0;
  best.feature = 0xFFFFFFFFu;
  best.thr_id = 0xFFFFFFFFu;
  best.lcount = best.rcount = 0;
  for (int i = lane; i < n; i += 64) {
    const qr_split_t r = recs[(size_t)i * stride];
    if (r.feature != 0xFFFFFFFFu && (r.score > best.score || (r.score == best.score && r.feature < best.feature)))
      best = r;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double os = __shfl_xor(best.score, off, 64);
    const uint32_t of = __shfl_xor(best.feature, off, 64);
    const uint32_t ot = __shfl_xor(best.thr_id, off, 64);
    if (of != 0xFFFFFFFFu && (os > best.score || (os == best.score && of < best.feature))) {
      best.score = os;
      best.feature = of;
      best.thr_id = ot;
    }
  }
  return best;
}

__device__ __forceinline__ void obl_level_body(QrTreeState *__restrict__ ts, const int level,
                                               const uint32_t N,
                                               const qr_split_t *__restrict__ featrec,
                                               const int flocal,
                                               const QrScalars *__restrict__ scal,
                                               uint32_t *pick, const qr_split_t *__restrict__ recs_all,
                                               const int world, const u64 Nglobal, const int root_buf) {
  const int lane = threadIdx.x;
  if (level == 0 && lane == 0) {
    QrNode *root = &ts->nodes[0];
    root->begin = 0;
    root->end = N;
    root->buf = root_buf;  // 2: every document, 0: this iteration's sample (its list)
    root->hslot = 0;
    root->feature = -1;
    root->thr_id = -1;
    root->threshold = 0.f;
    root->left = root->right = root->parent = -1;
    root->leaf_id = -1;
    node_stats(root, scal->root_sum, scal->root_ss, Nglobal ? Nglobal : (u64)N);
    ts->nnodes = 1;
    ts->obl_done = 0;
    ts->nsplits = 0;
    ts->desc.active = 0;
  }
  // (lane 0 wrote obl_done = 0 above at level 0; the others may still see the old value)
  const int was_done = level == 0 ? 0 : ts->obl_done;
  if (was_done) {
    if (lane == 0) pick[0] = 1;
    return;
  }
  // (feature-sharded: the ranks' bests were all-gathered, slot 0 of every rank's pair)
  const qr_split_t best = recs_all ? obl_pick(recs_all, world, 2) : obl_pick(featrec, flocal, 1);
  if (lane != 0) return;
  ts->obl_level = level;
  if (best.feature == 0xFFFFFFFFu) {  // ot.cc:96: node is unsplittable
    ts->obl_done = 1;
    pick[0] = 2;  // (finished at this level: obl_plan_body numbers the leaves)
    return;
  }
  pick[0] = 0;
  pick[1] = best.feature;
  pick[2] = best.thr_id;
  ts->obl_f = best.feature;
  ts->obl_t = best.thr_id;
  ts->obl_score = best.score;
  qr_split_t *lg = &ts->split_log[ts->nsplits++];
  lg->score = best.score;
  lg->feature = best.feature;
  lg->thr_id = best.thr_id;
  lg->lcount = lg->rcount = 0;
}

// Descriptors of ALL nodes of the level for the level's split (one thread per
// node), the children's records, and the work plan of the level's launches: how
// many histogram workgroups / partial slots / partition workgroups each node gets
// and the workgroup -> node maps.
__device__ __forceinline__ void obl_plan_body(
    QrTreeState *__restrict__ ts, const int level, const int last_level, const int G,
    const int flocal, const uint32_t *__restrict__ hcnt, const float *__restrict__ thr,
    const int32_t *__restrict__ gf2lf, const QrBlock *__restrict__ blocks, const int nblocks,
    uint32_t *__restrict__ hist_map, uint32_t *__restrict__ part_map, const uint32_t N,
    const qr_split_t *__restrict__ featrec, const QrScalars *__restrict__ scal,
    const uint32_t *__restrict__ woff, const size_t wcells, const qr_split_t *__restrict__ recs_all,
    const int world, const uint32_t *__restrict__ lcounts, const uint32_t *__restrict__ hcnt_loc,
    const u64 Nglobal, const int root_buf) {
  // (woff != null: wide-bin context -- ragged rows, no histogram plan: k_wide.hip's
  // launches are sized on the host.  recs_all / lcounts != null: feature-sharded -- the level's
  // split is the best of the ranks' records, and the nodes' left counts came with the mask)
  __shared__ uint32_t sh_a[QR_MAXLEVEL], sh_b[QR_MAXLEVEL], sh_c[QR_MAXLEVEL];
  __shared__ uint32_t tot_small, tot_hw, tot_pw;
  __shared__ uint32_t pick[3];
  // (hcnt_loc != null: document-sharded -- hcnt holds the counts over ALL ranks' documents,
  // which decide the split, the smaller side and the nodes' sizes; where the rank's own lists
  // are cut is a matter of its own counts)
  if (threadIdx.x < 64) obl_level_body(ts, level, N, featrec, flocal, scal, pick, recs_all, world, Nglobal, root_buf);
  __syncthreads();
  const int nodes = 1 << level;
  const int j = threadIdx.x;
  if (pick[0]) {
    // pick[0] == 2: the tree ends HERE (ot.cc:96) -- its leaves are this level's nodes, and
    // RTNode::save_leaves' left-first order (rtnode.cc:34-46) is their index order: numbered
    // now, so that the single-GPU path needs no k_finish launch
    if (pick[0] == 2 && j < nodes) {
      const int node = nodes - 1 + j;
      ts->nodes[node].leaf_id = j;
      ts->leaf_nodes[j] = node;
      ts->leaf_begin[j] = ts->nodes[node].begin;
      if (j == 0) {
        ts->nleaves = nodes;
        ts->leaf_begin[nodes] = ts->nodes[0].end;
      }
    }
    return;
  }
  const uint32_t f = pick[1], t = pick[2];
  const int lf = gf2lf[f];
  QrLevelNode ln;
  ln.active = 0;
  ln.small_n = 0;
  uint32_t nseg = 0;
  if (j < nodes) {
    const int node = nodes - 1 + j;
    QrNode *nd = &ts->nodes[node];
    // (a rank that does not own the feature: the count of ANY of its features' last slot is
    // the node's size)
    const int lfc = lf >= 0 ? lf : 0;
    const size_t base = woff ? (size_t)nd->hslot * wcells + woff[lfc] : ((size_t)nd->hslot * flocal + lfc) * 256;
    const uint32_t lastt = woff ? woff[lfc + 1] - woff[lfc] - 1 : 255u;
    const uint32_t lcount_all = lcounts ? lcounts[j] : hcnt[base + t];
    const uint32_t rcount_all = hcnt[base + lastt] - lcount_all;
    const uint32_t lcount = hcnt_loc ? hcnt_loc[base + t] : lcount_all;  // of the rank's own list
    const int li = 2 * node + 1, ri = 2 * node + 2;
    nseg = nd->end - nd->begin;
    ln.active = 1;
    ln.begin = nd->begin;
    ln.end = nd->end;
    ln.src_buf = nd->buf;
    ln.dst_buf = nd->buf == 0 ? 1 : 0;
    ln.lcount = lcount;
    ln.small_is_left = lcount_all <= rcount_all;
    ln.parent_slot = nd->hslot;
    ln.small_slot = ln.small_is_left ? li : ri;
    ln.big_slot = ln.small_is_left ? ri : li;
    ln.small_begin = ln.small_is_left ? nd->begin : nd->begin + lcount;
    ln.small_n = ln.small_is_left ? lcount : nseg - lcount;
    nd->feature = (int32_t)f;
    nd->thr_id = (int32_t)t;
    nd->threshold = thr[(woff ? (size_t)woff[f] : (size_t)f * QR_MAX_BINS) + t];
    nd->left = li;
    nd->right = ri;
    QrNode *L = &ts->nodes[li], *R = &ts->nodes[ri];
    L->begin = nd->begin;
    L->end = nd->begin + lcount;
    R->begin = L->end;
    R->end = nd->end;
    L->buf = R->buf = ln.dst_buf;
    L->hslot = li;
    R->hslot = ri;
    L->feature = R->feature = -1;
    L->thr_id = R->thr_id = -1;
    L->threshold = R->threshold = 0.f;
    L->left = L->right = R->left = R->right = -1;
    L->parent = R->parent = node;
    L->leaf_id = R->leaf_id = -1;
    L->count = lcount_all;
    R->count = rcount_all;
    L->sum = R->sum = L->ss = R->ss = L->deviance = R->deviance = 0.0;
    L->value = R->value = 0.0;  // overwritten by update_output (ot.cc:141-149)
    if (last_level) {  // the leaves, in left-first order = index order (see above)
      L->leaf_id = 2 * j;
      R->leaf_id = 2 * j + 1;
      ts->leaf_nodes[2 * j] = li;
      ts->leaf_nodes[2 * j + 1] = ri;
      ts->leaf_begin[2 * j] = L->begin;
      ts->leaf_begin[2 * j + 1] = R->begin;
      if (j == 0) ts->nleaves = 2 * nodes;
      if (j == nodes - 1) ts->leaf_begin[2 * nodes] = nd->end;
    }
  }
  sh_a[j] = ln.small_n;
  __syncthreads();
  if (j == 0) {
    uint32_t s = 0;
    for (int i = 0; i < nodes; ++i) s += sh_a[i];
    tot_small = s ? s : 1u;
  }
  __syncthreads();
  // One plan quantum for the whole level: every histogram workgroup of every node
  // and block then carries the same load.  Total workgroups <= geff + nodes *
  // nblocks, i.e. within the G compute units (a single round) while the level is
  // narrow enough.  ot.cc:127: no histograms for the leaves of the last level.
  uint32_t hw = 0, slots = 0, pw = 0;
  if (j < nodes) {
    pw = (nseg + QR_PART_SLICE - 1) / QR_PART_SLICE;
    ln.q = 0;
    if (!last_level && !woff) {
      const int spare = G - nodes * nblocks;
      ln.q = qr_plan_quantum((unsigned long long)tot_small * qr_plan_wsum(nblocks, blocks),
                             spare > G / 4 ? spare : G / 4);
      QrPlan pl;
      qr_make_plan(ln.small_n, nblocks, blocks, ln.q, &pl);
      hw = (uint32_t)pl.wg_start[nblocks];
      slots = hw * (uint32_t)pl.kmax;
    }
  }
  sh_a[j] = hw;
  sh_b[j] = slots;
  sh_c[j] = pw;
  __syncthreads();
  if (j == 0) {  // exclusive prefixes
    uint32_t a = 0, b = 0, c = 0;
    for (int i = 0; i < nodes; ++i) {
      const uint32_t x = sh_a[i], y = sh_b[i], z = sh_c[i];
      sh_a[i] = a;
      sh_b[i] = b;
      sh_c[i] = c;
      a += x;
      b += y;
      c += z;
    }
    tot_hw = a;
    tot_pw = c;
    ts->l_hist_wgs = a;
    ts->l_part_wgs = c;
    ts->l_nodes = nodes;
    ts->l_owner_local = lf;
    if (ts->nnodes < 4 * nodes - 1) ts->nnodes = 4 * nodes - 1;  // children of the level exist
    ts->part_epoch++;
  }
  __syncthreads();
  if (j < nodes) {
    ln.slot_base = sh_b[j];
    ln.part_first = sh_c[j];
    ln.pad = 0;
    ts->lnode[j] = ln;
  }
  // the workgroup -> node maps of the level's launches, by ALL threads (a thread per node
  // writing its node's entries one after the other was 18 us at the root level: one lane, 750
  // stores): entry x belongs to the last node whose first entry is not beyond x
  auto owner = [&](const uint32_t *first, const uint32_t x) {
    int lo = 0, hi = nodes;  // first[lo] <= x (first[0] == 0); answer in [lo, hi)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (first[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
  };
  for (uint32_t x = threadIdx.x; x < tot_hw; x += blockDim.x) {
    const int n = owner(sh_a, x);
    hist_map[x] = ((uint32_t)n << 16) | (x - sh_a[n]);
  }
  for (uint32_t w = threadIdx.x; w < tot_pw; w += blockDim.x) part_map[w] = (uint32_t)owner(sh_c, w);
  // (Measured and not kept: ready-made per-workgroup shares for the level's partition and
  // histogram launches, as batched leaf-wise growth has them (QrHistWg / QrPartWg) -- the
  // workgroups save a dependent read each, but this single workgroup, which sits on the chain,
  // pays for writing them: 0.577 against 0.574 ms per iteration at depth 6.)
}
__global__ __launch_bounds__(256) void k_obl_plan(
    QrTreeState *__restrict__ ts, const int level, const int last_level, const int G,
    const int flocal, const uint32_t *__restrict__ hcnt, const float *__restrict__ thr,
    const int32_t *__restrict__ gf2lf, const QrBlock *__restrict__ blocks, const int nblocks,
    uint32_t *__restrict__ hist_map, uint32_t *__restrict__ part_map, const uint32_t N,
    const qr_split_t *__restrict__ featrec, const QrScalars *__restrict__ scal,
    const uint32_t *__restrict__ woff, const size_t wcells, const qr_split_t *__restrict__ recs_all,
    const int world, const uint32_t *__restrict__ lcounts, const uint32_t *__restrict__ hcnt_loc,
    const u64 Nglobal, const int root_buf) {
  QR_JITTER();
  obl_plan_body(ts, level, last_level, G, flocal, hcnt, thr, gf2lf, blocks, nblocks, hist_map, part_map, N,
                featrec, scal, woff, wcells, recs_all, world, lcounts, hcnt_loc, Nglobal, root_buf);
}

// (Measured and not kept: k_obl_fill and k_obl_plan in ONE launch, the plan run by the workgroup
// that arrives last at a ticket -- release by every arriving workgroup, acquire by the last.
// 100.2 us for the six levels against 45.0 + 49.7 in two launches, 0.570 against 0.563 ms per
// iteration at depth 6: the hand-over costs more than the launch it saves, as it did between
// scan and control step of the leaf-wise path)

// Feature-sharded level-wise growth (SURVEY.md section 8e applied to ot.cc:32-201).  Every
// rank sums the level's gains over ITS features (k_obl_fill) and publishes its best
// (k_obl_propose); after the all-gather every rank picks the same (feature, slot)
// (obl_pick); the owner of that feature publishes, for the all-reduce that doubles as a
// broadcast, the go-left bit of every DOCUMENT (all nodes of the level take the same
// split) and the left count of every node of the level (k_obl_mark); then every rank
// partitions all nodes with the mask and builds the children's histograms of its own
// features (k_obl_plan, k_partition_level, k_hist_level, k_redscan_level).
__global__ __launch_bounds__(64) void k_obl_propose(const QrTreeState *__restrict__ ts, const int level,
                                                    const qr_split_t *__restrict__ featrec,
                                                    const int flocal, qr_split_t *__restrict__ recs_local) {
  QR_JITTER();
  qr_split_t best;
  if (level > 0 && ts->obl_done) {
    best.score = -1.0;
    best.feature = 0xFFFFFFFFu;
    best.thr_id = 0xFFFFFFFFu;
    best.lcount = best.rcount = 0;
  } else
    best = obl_pick(featrec, flocal, 1);
  if (threadIdx.x == 0) {
    best.lcount = best.rcount = 0;
    recs_local[0] = best;
    recs_local[1] = best;
  }
}

__global__ __launch_bounds__(256) void k_obl_mark(
    const QrTreeState *__restrict__ ts, const int level, const qr_split_t *__restrict__ recs_all,
    const int world, const int32_t *__restrict__ gf2lf, const uint32_t N,
    const uint32_t *__restrict__ hcnt, const int flocal, const uint8_t *__restrict__ fm,
    uint32_t *__restrict__ mask, const uint32_t mask_words) {
  QR_JITTER();
  __shared__ uint32_t sh_f, sh_t;
  if (threadIdx.x < 64) {
    const qr_split_t best = obl_pick(recs_all, world, 2);
    if (threadIdx.x == 0) {
      sh_f = best.feature;
      sh_t = best.thr_id;
    }
  }
  __syncthreads();
  const uint32_t f = sh_f, t = sh_t;
  const int lf = f == 0xFFFFFFFFu ? -1 : gf2lf[f];
  const uint32_t w = blockIdx.x * 256 + threadIdx.x;
  if (w < mask_words) {
    uint32_t bits = 0;
    if (lf >= 0) {
      // the word's 32 bytes are requested together (positions clamped into the column: no
      // branch between the loads; one at a time they were 32 round trips per thread)
      const uint8_t *col = fm + (size_t)lf * N;
      uint32_t bv[32];
#pragma unroll
      for (uint32_t k = 0; k < 32; ++k) {
        const uint32_t d = w * 32 + k;
        bv[k] = col[d < N ? d : N - 1];
      }
#pragma unroll
      for (uint32_t k = 0; k < 32; ++k)
        if (w * 32 + k < N && bv[k] <= t) bits |= 1u << k;
    }
    mask[w] = bits;
  }
  // the nodes' left counts ride behind the bits (QR_MAXLEVEL words)
  if (blockIdx.x == 0) {
    const int nodes = 1 << level;
    for (int j = threadIdx.x; j < QR_MAXLEVEL; j += 256) {
      uint32_t v = 0;
      if (lf >= 0 && j < nodes) v = hcnt[((size_t)(nodes - 1 + j) * flocal + lf) * 256 + t];  // hslot == node index
      mask[mask_words + j] = v;
    }
  }
}

__global__ void k_obl_reset(QrTreeState *ts, int maxnodes, u64 minls) {
  QR_JITTER();
  obl_reset_body(ts, (int)(blockIdx.x * blockDim.x + threadIdx.x), maxnodes, minls);
}

// ===========================================================================
// host launchers
// ===========================================================================
static size_t hist_lds(const qr_ctx *c) {
  int fwmax = 16;
  for (const auto &b : c->blocks) fwmax = b.fw > fwmax ? b.fw : fwmax;
  return (size_t)256 * fwmax * 8;
}

static int launch_scan(qr_ctx *c, int root_mode);

// the root launch's per-workgroup shares (what hist_body derives per workgroup), made once per
// (documents of the root, grid, list buffer) and kept on the device
static int root_shares(qr_ctx *c, uint32_t rootn, int G, int root_buf) {
  // (the key names everything the shares are made from -- the documents of the root, the
  // grid, the list buffer AND the generation of c->blocks -- and validity is a flag of its own, so a
  // root of 0 documents or a rebuild of the blocks by any path cannot be served stale shares)
  if (c->root_wg_valid && c->d_root_wg && c->root_wg_n == rootn && c->root_wg_g == G &&
      c->root_wg_buf == root_buf && c->root_wg_gen == c->blocks_gen)
    return QR_OK;
  c->root_wg_valid = false;
  std::vector<QrHistWg> h((size_t)G);
  QrPlan plan;
  const uint32_t q = qr_plan_quantum((unsigned long long)rootn * qr_plan_wsum(c->nblocks, c->blocks.data()),
                                     G - c->nblocks);
  qr_make_plan(rootn, c->nblocks, c->blocks.data(), q, &plan);
  for (int wg = 0; wg < G; ++wg) {
    QrHistWg d{};
    int b = -1;
    for (int i = 0; i < c->nblocks; ++i)
      if (wg >= plan.wg_start[i] && wg < plan.wg_start[i + 1]) b = i;
    if (b >= 0) {
      const uint32_t j = (uint32_t)(wg - plan.wg_start[b]), per = plan.per[b];
      const uint32_t r0 = j * per, r1 = r0 + per < rootn ? r0 + per : rootn;
      d.begin = r0;
      d.count = r1 > r0 ? r1 - r0 : 0u;
      d.slot = (uint32_t)wg * (uint32_t)plan.kmax;
      d.block = (uint16_t)b;
      d.buf = (uint8_t)root_buf;
      d.fw = (uint8_t)c->blocks[b].fw;
      d.off256 = (uint32_t)(c->blocks[b].off >> 8);
    }
    h[(size_t)wg] = d;
  }
  // ... and the root scan launch's per-feature shares of the same plan (what k_redscan's root
  // branch derives: thread 0 planning, a barrier, the block search)
  std::vector<QrScanWg> hs((size_t)c->flocal);
  for (int lf = 0; lf < c->flocal; ++lf) {
    QrScanWg d{};
    int b = 0;
    for (int i = 0; i < c->nblocks; ++i)
      if (lf >= c->blocks[i].lf0 && lf < c->blocks[i].lf0 + c->blocks[i].nreal) b = i;
    d.active = 1;
    d.slot0 = (uint32_t)(plan.wg_start[b] * plan.kmax);
    d.total = (uint32_t)((plan.wg_start[b + 1] - plan.wg_start[b]) * plan.kmax);
    d.per = plan.per[b];
    d.n = rootn;
    d.kmax = plan.kmax;
    d.small_slot = 0;
    d.big_slot = d.parent_slot = -1;
    d.small_is_left = 1;
    d.col = (uint32_t)(lf - c->blocks[b].lf0);
    hs[(size_t)lf] = d;
  }
  // one wait (a launch that reads the old shares may be in flight), then both tables
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  if (!c->d_root_scan || c->root_scan_n != c->flocal) {
    if (c->d_root_scan) (void)hipFree(c->d_root_scan);
    c->d_root_scan = nullptr;
    c->root_scan_n = 0;
    QR_CHECK(c, hipMalloc((void **)&c->d_root_scan, (size_t)c->flocal * sizeof(QrScanWg)));
    c->root_scan_n = c->flocal;
  }
  if (!c->d_root_wg || c->root_wg_g != G) {
    if (c->d_root_wg) (void)hipFree(c->d_root_wg);
    c->d_root_wg = nullptr;
    c->root_wg_g = 0;
    QR_CHECK(c, hipMalloc((void **)&c->d_root_wg, (size_t)G * sizeof(QrHistWg)));
    c->root_wg_g = G;
  }
  // (blocking copies: `hs` and `h` live on this frame, and an error return between an asynchronous
  // copy and its wait would leave the copy reading a dead frame -- VERDICT r5; the stream is idle here)
  QR_CHECK(c, hipMemcpy(c->d_root_scan, hs.data(), (size_t)c->flocal * sizeof(QrScanWg), hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_root_wg, h.data(), (size_t)G * sizeof(QrHistWg), hipMemcpyHostToDevice));
  c->root_wg_n = rootn;
  c->root_wg_g = G;
  c->root_wg_buf = root_buf;
  c->root_wg_gen = c->blocks_gen;
  c->root_wg_valid = true;
  return QR_OK;
}

static int launch_hist_scan(qr_ctx *c, int root_mode, bool fused = false) {
  // (the request for a reset workgroup is consumed HERE, whatever path this call
  // leaves by -- an early error return must not leave it set for the next tree's root scan)
  const int reset_nodes = c->obl_reset_nodes;
  c->obl_reset_nodes = 0;
  // The scalars of the lambda pass before this tree may still be unfinished (qr_lambda_compute
  // defers them, qr_prep.h).  The fused root launches let the workgroups that finish them ride
  // in the scan launch and take the scale from the iteration's slot set meanwhile; every other
  // path has them finished first, in a launch of their own.
  const bool ride = c->prep_deferred && root_mode && fused && !c->wide;
  if (c->prep_deferred && !ride) {
    const int frc = qr_k_prep_flush(c);
    if (frc) return frc;
  }
  QrPrepJob prep{};
  const unsigned long long *root_slots = nullptr;
  if (ride) {
    c->prep_deferred = false;
    qr_k_prep_job(c, c->prep_nss, c->prep_with_metric, c->prep_publish, &prep);
    root_slots = prep.slots = qr_prep_slots(c, c->prep_parity);
  } else {
    prep.nwg = 0;
  }
  if (c->wide) {  // more than 255 thresholds: k_wide.hip
    const int wrc = qr_k_whist_scan(c, root_mode);
    if (wrc) return wrc;
    if (c->world > 1 && !c->dmode) {  // feature-sharded: the rank's best records for the all-gather
      hipLaunchKernelGGL(k_merge, dim3(1), dim3(128), 0, c->stream, c->d_tree, root_mode,
                         c->d_featrec, c->flocal, c->d_hsum, c->d_hcnt, c->d_gf2lf,
                         c->d_recs_local, c->mf_k, c->mf_seed + c->tree_counter, (uint32_t)c->F);
      QR_CHECK(c, hipGetLastError());
    }
    return QR_OK;
  }
  const size_t lds = hist_lds(c);
  size_t &attr_lds = c->attr_hist_lds;
  if (lds > attr_lds) {
    QR_CHECK(c, hipFuncSetAttribute((const void *)k_hist,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    QR_CHECK(c, hipFuncSetAttribute((const void *)k_hist_root,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    attr_lds = lds;
  }
  const int G = c->ncu;
  const QrScanWg *shares_scan = nullptr;  // (root: the scan launch's per-feature shares, root_shares)
  const bool prof = c->prof_on && root_mode && (c->prof_tick++ % c->prof_stride) == 0;
  const uint32_t rootn = (uint32_t)(c->sub_k ? c->sub_n : c->N);  // documents of the root node
  if (root_mode) {
    const int root_buf = c->sub_k ? 0 : 2;  // the sample's list / every document
    const int src = root_shares(c, rootn, G, root_buf);
    if (src) return src;
    const QrHistWg *shares = c->d_root_wg;
    shares_scan = c->d_root_scan;
    if (prof) {
      // bench.py's roofline: the two events are attached to the launch itself (they
      // take the kernel's own begin / end timestamps, like the profiler's trace)
      hipEvent_t e0 = nullptr, e1 = nullptr;
      QR_CHECK(c, hipEventCreate(&e0));
      QR_CHECK(c, hipEventCreate(&e1));
      hipExtLaunchKernelGGL(k_hist_root, dim3(G), dim3(1024), lds, c->stream, e0, e1, 0, rootn,
                            c->d_blocks, c->nblocks, c->d_bins, c->d_order[0], c->d_lambda,
                            c->d_scalars, (u64 *)c->d_partials, root_buf, fused ? 1 : 0, root_slots, shares);
      QR_CHECK(c, hipGetLastError());
      c->prof_events.push_back({e0, e1});
    } else {
      hipLaunchKernelGGL(k_hist_root, dim3(G), dim3(1024), lds, c->stream, rootn, c->d_blocks,
                         c->nblocks, c->d_bins, c->d_order[0], c->d_lambda, c->d_scalars,
                         (u64 *)c->d_partials, root_buf, fused ? 1 : 0, root_slots, shares);
      QR_CHECK(c, hipGetLastError());
    }
  } else {
    hipLaunchKernelGGL(k_hist, dim3(G), dim3(1024), lds, c->stream, c->d_tree,
                       root_mode, rootn, c->d_blocks, c->nblocks, c->d_bins,
                       c->d_order[0], c->d_order[1], c->d_lambda, c->d_scalars,
                       (u64 *)c->d_partials, c->dmode, c->sub_k ? 0 : 2);
    QR_CHECK(c, hipGetLastError());
  }
  if (fused) {  // batched growth: feature-major partials, reduce + scan in one launch
    // (level-wise growth: one more workgroup, the last, resets the tree state -- c->obl_reset_nodes)
    QrResetJob reset{c->d_tree, (u64)c->cur_minls, (int32_t)reset_nodes, 0};
    hipLaunchKernelGGL(k_redscan, dim3(c->flocal + prep.nwg + (reset.maxnodes ? 1 : 0), 1), dim3(1024), 0, c->stream, c->d_tree, 1, rootn,
                       c->d_lplan, c->d_blocks, c->nblocks, G, (const u64 *)c->d_partials, c->d_hsum,
                       c->d_hcnt, c->flocal, c->d_thr_size, c->d_lf2gf, c->d_scalars, c->d_featrec,
                       c->d_thr, c->d_featthr, shares_scan, (u64)c->cur_minls,
                       (const double *)nullptr, (double *)nullptr, prep, reset);
    QR_CHECK(c, hipGetLastError());
    return QR_OK;
  }
  size_t cells = 0;
  for (const auto &b : c->blocks) cells += (size_t)256 * b.fw;
  hipLaunchKernelGGL(k_reduce, dim3((unsigned)(cells / 64)), dim3(512), 0, c->stream,
                     c->d_tree, root_mode, rootn, c->d_blocks, c->nblocks, G,
                     (const u64 *)c->d_partials, c->d_red_sum, c->d_red_cnt, c->dmode,
                     c->d_part_ss, c->dmode ? c->d_xh + 2 * c->xh_cells : (long long *)nullptr,
                     c->rank, c->world, c->dmode ? c->d_red_cnt_loc : (uint32_t *)nullptr);
  QR_CHECK(c, hipGetLastError());
  if (c->dmode) return QR_OK;  // the scan follows the all-reduce (launch_scan)
  return launch_scan(c, root_mode);
}

static int launch_scan(qr_ctx *c, int root_mode) {
  hipLaunchKernelGGL(k_scan, dim3(c->flocal), dim3(256), 0, c->stream, c->d_tree,
                     root_mode, (uint32_t)c->N, c->d_blocks, c->nblocks, c->d_red_sum,
                     c->d_red_cnt, c->d_hsum, c->d_hcnt, c->flocal, c->d_thr_size,
                     c->d_lf2gf, c->d_scalars, c->d_featrec, c->dmode ? 2u : 1u,
                     c->dmode ? c->d_red_cnt_loc : (const uint32_t *)nullptr,
                     c->dmode ? c->d_hcnt_loc : (uint32_t *)nullptr, c->d_thr, c->d_featthr);
  QR_CHECK(c, hipGetLastError());
  if (c->world > 1 && !c->dmode) {
    hipLaunchKernelGGL(k_merge, dim3(1), dim3(128), 0, c->stream, c->d_tree, root_mode,
                       c->d_featrec, c->flocal, c->d_hsum, c->d_hcnt, c->d_gf2lf,
                       c->d_recs_local, c->mf_k, c->mf_seed + c->tree_counter, (uint32_t)c->F);
    QR_CHECK(c, hipGetLastError());
  }
  return QR_OK;
}

__global__ void k_tree_reset(QrTreeState *ts, int nleaves, u64 minls) {
  QR_JITTER();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ts->nleaves_req = nleaves;
  ts->nnodes = 0;
  ts->taken = 0;
  ts->done = 0;
  ts->step = 0;
  ts->nsplits = 0;
  ts->minls = minls;
  ts->heap_size = 0;
  ts->desc.active = 0;
  ts->nleaves = 0;
  ts->l_nodes = 0;
  ts->incomplete = 0;
  ts->real_steps = 0;
}

int qr_k_tree_begin(qr_ctx *c, size_t nleaves, uint64_t minls) {
  QR_DBG_CAPS(c);
  c->finish_in_decide = false;
  c->tree_step = 0;
  c->tree_counter += 0x9E3779B97F4A7C15ull;  // a fresh feature-subset stream per tree
  hipLaunchKernelGGL(k_tree_reset, dim3(1), dim3(64), 0, c->stream, c->d_tree,
                     (int)nleaves, (u64)minls);
  QR_CHECK(c, hipGetLastError());
  return launch_hist_scan(c, 1);
}

int qr_k_tree_decide(qr_ctx *c) {
  const bool fshard = c->world > 1 && !c->dmode;
  const qr_split_t *recs = fshard ? c->d_recs_all : c->d_recs_local;
  if (c->dmode) {
    // the histogram of this step has just been all-reduced; the first decide of a
    // tree follows the root histogram
    int rc = c->wide ? qr_k_wscan_doc(c, c->tree_step == 0) : launch_scan(c, c->tree_step == 0);
    if (rc) return rc;
  }
  // (Letting k_scan's last workgroup take the decision saves this launch but was
  // measured slower: the agent-scope release/acquire it needs writes back and
  // invalidates the XCD L2s, 22.8 us for the fused kernel against 6 + 9 us.)
  hipLaunchKernelGGL(k_decide, dim3(1), dim3(128), 0, c->stream, c->d_tree,
                     (uint32_t)(c->sub_k ? c->sub_n : c->N), c->flocal, recs, fshard ? c->world : 1,
                     c->d_scalars, c->d_part_ss, c->wide ? c->d_wthr : c->d_thr, c->d_gf2lf, c->d_featrec,
                     c->d_hcnt_loc, c->dmode, (u64)(c->sub_k ? c->sub_k : c->Nglobal),
                     c->dmode ? c->d_xh + 2 * c->xh_cells : (const long long *)nullptr,
                     c->world, c->mf_k, c->mf_seed + c->tree_counter, (uint32_t)c->F,
                     c->sub_k ? 0 : 2, c->wide ? c->d_woff : (const uint32_t *)nullptr,
                     c->wide ? c->wcells : (size_t)0);
  QR_CHECK(c, hipGetLastError());
  ++c->tree_step;
  if (fshard) {
    const unsigned grid = (unsigned)((c->mask_words + 255) / 256);
    hipLaunchKernelGGL(k_mask, dim3(grid), dim3(256), 0, c->stream, c->d_tree,
                       c->wide ? reinterpret_cast<const uint8_t *>(c->d_wbins) : c->d_bins_fm, (uint32_t)c->N,
                       c->d_order[0], c->d_order[1], c->d_mask, (uint32_t)c->mask_words, c->wide ? 1 : 0);
    QR_CHECK(c, hipGetLastError());
  }
  return QR_OK;
}

int qr_k_xpop(qr_ctx *c, int root_mode, size_t nleaves, uint64_t minls, int final_call) {
  hipLaunchKernelGGL(k_xpop, dim3(1), dim3(128), 0, c->stream, c->d_tree, root_mode, (int)nleaves, (u64)minls,
                     (uint32_t)(c->sub_k ? c->sub_n : c->N), c->d_scalars, c->d_part_ss,
                     (unsigned long long *)c->d_xgbest, c->flocal, final_call,
                     final_call ? c->d_pin->early : (int64_t *)nullptr, (long long)(final_call ? ++c->early_seq : 0),
                     c->sub_k ? 0 : 2);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_xapply(qr_ctx *c, int root_mode) {
  hipLaunchKernelGGL(k_xapply, dim3(1), dim3(64), 0, c->stream, c->d_tree, root_mode, c->flocal, c->d_featrec,
                     c->d_wthr, c->d_gf2lf, c->d_woff, c->mf_k, c->mf_seed + c->tree_counter, (uint32_t)c->F,
                     (const long long *)c->d_xcs, c->d_xnode_tot);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// the partition of the document-order lists for the split k_xapply made (as qr_k_tree_apply,
// without the histogram launches behind it)
int qr_k_xpartition(qr_ctx *c) {
  const unsigned pgrid = (unsigned)((c->N + QR_PART_SLICE - 1) / QR_PART_SLICE);
  hipLaunchKernelGGL(k_partition, dim3(pgrid), dim3(256), 0, c->stream, c->d_tree,
                     reinterpret_cast<const uint8_t *>(c->d_wbins), (uint32_t)c->N, c->d_order[0], c->d_order[1],
                     c->d_mask, 2, (u64 *)c->d_part_state, c->d_lambda, c->d_part_ss, 0);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_tree_apply(qr_ctx *c) {
  const unsigned pgrid = (unsigned)((c->N + QR_PART_SLICE - 1) / QR_PART_SLICE);
  const bool fshard = c->world > 1 && !c->dmode;
  if (c->wide && fshard) {  // the threshold value that came with the reduced mask
    hipLaunchKernelGGL(k_thr_patch, dim3(1), dim3(64), 0, c->stream, c->d_tree, c->d_mask + c->mask_words);
    QR_CHECK(c, hipGetLastError());
  }
  // (feature-sharded: the reduced go-left bits, whatever the bins are; single-GPU wide: u32 bins)
  const int use_mask = fshard ? 1 : (c->wide ? 2 : 0);
  // (document-sharded: the rank's own left count comes from its local prefix counts)
  hipLaunchKernelGGL(k_partition, dim3(pgrid), dim3(256), 0, c->stream, c->d_tree,
                     c->wide ? reinterpret_cast<const uint8_t *>(c->d_wbins) : c->d_bins_fm,
                     (uint32_t)c->N, c->d_order[0], c->d_order[1],
                     c->d_mask, use_mask, (u64 *)c->d_part_state, c->d_lambda, c->d_part_ss,
                     c->dmode);
  QR_CHECK(c, hipGetLastError());
  return launch_hist_scan(c, 0);
}


// Leaf-wise tree with up to QR_BATCH splits per step (k_decide_batch): the whole fit is
// enqueued at once; steps the tree does not need leave at once.
// the launches of one growth step behind its control call: the batch's child histograms,
// then reduce + scan
static int launch_batch_hist_scan(qr_ctx *c, const unsigned hg, const uint32_t rootn, const uint64_t minls,
                                  const double *pss, const QrTreeState *ts_step = nullptr) {
  // (wide-bin contexts: k_wide.hip's kernels on the step's jobs, read from the copy of the tree
  // state the step's control call wrote; the child sums come from the partition, `pss`)
  if (c->wide) return qr_k_whist_scan_batch(c, ts_step ? ts_step : c->d_tree, pss);
  const size_t lds = hist_lds(c);
  if (c->prof_on && c->prof_child) {  // bench.py's roofline_child_hist: events on the launch itself
    hipEvent_t e0 = nullptr, e1 = nullptr;
    QR_CHECK(c, hipEventCreate(&e0));
    QR_CHECK(c, hipEventCreate(&e1));
    hipExtLaunchKernelGGL(k_hist_batch, dim3(hg), dim3(1024), lds, c->stream, e0, e1, 0, c->d_lhist_wg,
                          c->d_blocks, c->d_bins, c->d_order[0], c->d_order[1], c->d_lambda,
                          c->d_scalars, (u64 *)c->d_lpartials, c->d_lhistsum);
    c->prof_events_child.push_back({e0, e1});
  } else
    hipLaunchKernelGGL(k_hist_batch, dim3(hg), dim3(1024), lds, c->stream, c->d_lhist_wg,
                       c->d_blocks, c->d_bins, c->d_order[0], c->d_order[1], c->d_lambda,
                       c->d_scalars, (u64 *)c->d_lpartials, c->d_lhistsum);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_redscan, dim3(c->flocal, QR_BATCH), dim3(1024), 0, c->stream, c->d_tree, 0,
                     rootn, c->d_lplan, c->d_blocks, c->nblocks, c->ncu, (const u64 *)c->d_lpartials,
                     c->d_hsum, c->d_hcnt, c->flocal, c->d_thr_size, c->d_lf2gf, c->d_scalars,
                     c->d_featrec, c->d_thr, c->d_featthr, c->d_lscan_wg, (u64)minls, c->d_lhistsum, c->d_jobsum,
                     QrPrepJob{}, QrResetJob{});
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

struct BatchGeom {
  unsigned pg, hg;
  uint32_t rootn;
  bool small, fused;
  int stage_nodes;
};
static BatchGeom batch_geom(const qr_ctx *c, size_t nleaves) {
  BatchGeom g;
  g.pg = (unsigned)std::min<size_t>(c->lpart_cap, c->N / QR_PART_SLICE + QR_BATCH + 1);
  g.hg = (unsigned)std::min<size_t>(c->lhist_cap, (size_t)std::max(c->ncu, c->ncu / 4 + QR_BATCH * c->nblocks));
  g.rootn = (uint32_t)(c->sub_k ? c->sub_n : c->N);
  // final ids [0, 2 nleaves + 1) + provisional ones [.., 4 nleaves + 1)
  g.small = 4 * nleaves + 1 + 2 * QR_BATCH <= QR_BATCH_LDS_SMALL;
  g.stage_nodes = 4 * nleaves + 1 + 2 * QR_BATCH <= QR_BATCH_LDS_LARGE ? (int)(4 * nleaves + 1) : 0;
  // Staged trees: the control step runs inside the partition launch (k_decide_part) and
  // the tree state ping-pongs between two copies, arranged so that the last call
  // (control step only: it accounts for the last batch) writes c->d_tree.
  g.fused = g.stage_nodes > 0 && c->d_tree2 != nullptr && c->N <= c->fuse_max_docs;
  return g;
}

// document shards: what decides whether splits ahead of their turn pay is the size of a RANK's
// launches; every rank must decide alike, so it is the ranks' average share of the root's documents
static inline u64 doc_spec_docs(const qr_ctx *c) {
  const u64 n = (u64)(c->sub_k ? c->sub_k : c->Nglobal);
  return c->world > 0 ? n / (u64)c->world : n;
}

// a control call on its own (k_decide_batch): the first of a tree (root), one between two
// steps of a tree whose state does not fit the LDS copies, or the last of the enqueued
// sequence (`final_call`: accounts for the last batch and tells whether more is to come)
static int launch_decide_batch(qr_ctx *c, const BatchGeom &g, size_t nleaves, uint64_t minls, int root,
                               QrTreeState *tout, const QrTreeState *tin, const double *pss_in,
                               int final_call) {
  hipLaunchKernelGGL(g.small ? k_decide_batch<QR_BATCH_LDS_SMALL> : k_decide_batch<QR_BATCH_LDS_LARGE>,
                     dim3(1), dim3(128 * QR_BATCH), 0, c->stream, tout, root, (int)nleaves, (u64)minls,
                     g.stage_nodes, g.rootn, c->flocal, c->d_scalars, c->d_jobsum, c->d_featrec, c->d_featthr,
                     (uint32_t)c->F, c->sub_k ? 0 : 2, c->ncu, c->d_blocks, c->nblocks, c->d_lhist_wg, g.hg,
                     c->d_lpart_wg, g.pg, c->d_lplan, c->d_lscan_wg, tin, final_call,
                     final_call ? c->d_pin->early : (int64_t *)nullptr, (long long)(final_call ? ++c->early_seq : 0),
                     c->dmode ? c->d_hcnt_loc : (const uint32_t *)nullptr,
                     (u64)(c->sub_k ? c->sub_k : c->Nglobal), doc_spec_docs(c));
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// Leaf-wise tree with up to QR_BATCH splits per step (k_decide_batch): the whole fit is
// enqueued at once.  A tree of L leaves needs at most L - 1 steps and usually far fewer
// (two splits per step; 5-6 of 9 on the bench workload), but how many is only known on
// the device, and a surplus step still costs three launches (~14 us).  So the host
// enqueues a GUESS -- the previous tree's steps + 1 (c->steps_hint) -- and the last control
// call of the sequence says whether it was enough (QrTreeState::incomplete).  If not
// (rare: consecutive trees have similar shapes), the leaf and score kernels behind it
// leave at once and qr_k_tree_continue carries the tree on when the host fetches the
// records.  The first tree of a context enqueues the worst case.
int qr_k_tree_fit_batch(qr_ctx *c, size_t nleaves, uint64_t minls) {
  QR_DBG_CAPS(c);
  c->tree_step = 0;
  c->tree_counter += 0x9E3779B97F4A7C15ull;
  c->cur_minls = minls;
  // (no reset launch: the first k_decide_batch call starts from its arguments)
  c->batch_root = true;
  int rc = launch_hist_scan(c, 1, true);  // root histogram -> slot 0, records -> featrec[0]
  c->batch_root = false;
  if (rc) return rc;
  const BatchGeom g = batch_geom(c, nleaves);
  c->finish_in_decide = g.stage_nodes > 0;
  size_t steps = nleaves - 1;
  if (c->steps_force >= 0)
    steps = std::min<size_t>(steps, (size_t)std::max<long>(c->steps_force, 1));
  else if (c->steps_hint)
    steps = std::min(steps, c->steps_hint);
  if (steps < 1) steps = 1;
  c->tree_step = (int)steps;
  QrTreeState *const T[2] = {c->d_tree, c->d_tree2};
  double *const PSS[2] = {c->d_lpart_ss, c->d_lpart_ss2};
  // (the partition's byte per document: the u8 feature-major copy, or the wide path's u32 bins)
  const uint8_t *const fm = c->wide ? reinterpret_cast<const uint8_t *>(c->d_wbins) : c->d_bins_fm;
  // calls 0 .. steps: call s < steps decides and partitions step s; call `steps` only decides
  for (size_t s = 0; s <= steps; ++s) {
    QrTreeState *tout = g.fused ? T[(steps - s) & 1] : c->d_tree;
    QrTreeState *tin = g.fused ? T[(steps + 1 - s) & 1] : c->d_tree;  // what call s - 1 wrote
    const double *pss_in = g.fused ? PSS[(s + 1) & 1] : c->d_lpart_ss;
    if (!g.fused || s == steps) {
      if ((rc = launch_decide_batch(c, g, nleaves, minls, s == 0 ? 1 : 0, tout,
                                    g.fused ? (const QrTreeState *)tin : (const QrTreeState *)nullptr, pss_in,
                                    s == steps ? 1 : 0)))
        return rc;
      if (s == steps) break;  // the last call only accounts for the last batch
      hipLaunchKernelGGL(k_partition_batch, dim3(g.pg), dim3(256), 0, c->stream, c->d_tree,
                         c->d_lpart_wg, fm, (uint32_t)c->N, c->d_order[0], c->d_order[1],
                         (u64 *)c->d_bpart_state, c->d_lambda, c->wide ? c->d_lpart_ss : (double *)nullptr,
                         ++c->bepoch, c->wide ? 1 : 0);
      QR_CHECK(c, hipGetLastError());
    } else {
      hipLaunchKernelGGL(g.small ? k_decide_part<QR_BATCH_LDS_SMALL> : k_decide_part<QR_BATCH_LDS_LARGE>,
                         dim3(g.pg), dim3(128 * QR_BATCH), 0, c->stream,
                         (const QrTreeState *)tin, tout, T[(steps + 1 - s) & 1], ++c->bepoch,
                         s == 0 ? 1 : 0, (int)nleaves, (u64)minls, g.stage_nodes, g.rootn, c->flocal,
                         c->d_scalars, c->d_jobsum, c->d_featrec, c->d_featthr, (uint32_t)c->F,
                         c->sub_k ? 0 : 2, c->ncu, c->d_blocks, c->nblocks, c->d_lhist_wg, g.hg,
                         c->d_lplan, c->d_lscan_wg, fm, (uint32_t)c->N, c->d_order[0],
                         c->d_order[1], (u64 *)c->d_bpart_state, c->d_lambda,
                         c->wide ? PSS[s & 1] : (double *)nullptr, c->wide ? 1 : 0,
                         (const uint32_t *)nullptr, (u64)0, (u64)0);
      QR_CHECK(c, hipGetLastError());
    }
    if ((rc = launch_batch_hist_scan(c, g.hg, g.rootn, minls, g.fused ? PSS[s & 1] : c->d_lpart_ss, tout))) return rc;
  }
  return QR_OK;
}

// The guess was too low: c->d_tree holds a consistent state with the next batch ready
// (the last control call prepared it like any other).  Apply it and carry on, one control
// call per step on the device-resident state, for the worst case that is left; the last
// call is final again (and cannot be incomplete: the leaf budget is exhausted by then).
int qr_k_tree_continue(qr_ctx *c, size_t nleaves, uint64_t minls, size_t steps_done, size_t max_steps) {
  QR_DBG_CAPS(c);
  const BatchGeom g = batch_geom(c, nleaves);
  // (max_steps: the caller may take the rest in pieces and look again after each -- the
  // worst case left is mostly launches that find nothing to do)
  size_t left = nleaves - 1 > steps_done ? nleaves - 1 - steps_done : 1;
  if (max_steps && left > max_steps) left = max_steps;
  int rc;
  for (size_t r = 0; r < left; ++r) {
    hipLaunchKernelGGL(k_partition_batch, dim3(g.pg), dim3(256), 0, c->stream, c->d_tree,
                       c->d_lpart_wg, c->wide ? reinterpret_cast<const uint8_t *>(c->d_wbins) : c->d_bins_fm,
                       (uint32_t)c->N, c->d_order[0], c->d_order[1], (u64 *)c->d_bpart_state, c->d_lambda,
                       c->wide ? c->d_lpart_ss : (double *)nullptr, ++c->bepoch, c->wide ? 1 : 0);
    QR_CHECK(c, hipGetLastError());
    if ((rc = launch_batch_hist_scan(c, g.hg, g.rootn, minls, c->d_lpart_ss, c->d_tree))) return rc;
    if ((rc = launch_decide_batch(c, g, nleaves, minls, 0, c->d_tree, (const QrTreeState *)nullptr,
                                  c->d_lpart_ss, r + 1 == left ? 1 : 0)))
      return rc;
  }
  return QR_OK;
}

// ---- document-sharded ranks: the same growth, phase by phase, with the host's all-reduces
// in between (dist.py DocShardedTrainer.fit_tree / host/mart_multi.cc):
//   root_hist -> [all-reduce d_xh] -> root_decide -> { apply -> [all-reduce d_xb] -> decide } x steps
// Every rank holds the same all-reduced integers and the same rank-ordered f64 sums, so the
// control steps agree bit for bit without exchanging anything else.
int qr_k_dbatch_root_hist(qr_ctx *c, size_t nleaves, uint64_t minls) {
  QR_DBG_CAPS(c);
  c->tree_step = 0;
  c->tree_counter += 0x9E3779B97F4A7C15ull;
  c->cur_minls = minls;
  const BatchGeom g = batch_geom(c, nleaves);
  c->finish_in_decide = g.stage_nodes > 0;
  return launch_hist_scan(c, 1);  // k_hist_root + k_reduce into the exchange buffer
}

// (Staged trees: the control step of a growth step rides in the NEXT step's partition launch
// (k_decide_part), as on one GPU; the tree state ping-pongs between c->d_tree and c->d_tree2
// (c->dtree_cur = the copy the last control step wrote) and the last control call of a sequence --
// a k_decide_batch of its own -- writes c->d_tree, where everything behind the growth looks.)
int qr_k_dbatch_root_decide(qr_ctx *c, size_t nleaves, uint64_t minls) {
  int rc = launch_scan(c, 1);  // slot 0 (sums, counts, the rank's own counts) + featrec[0]
  if (rc) return rc;
  const BatchGeom g = batch_geom(c, nleaves);
  c->dtree_cur = c->d_tree;
  c->dbatch_first = g.fused;   // (fused: the root's control step is the first partition launch's)
  c->dbatch_prepared = !g.fused;
  if (g.fused) return QR_OK;
  return launch_decide_batch(c, g, nleaves, minls, 1, c->d_tree, (const QrTreeState *)nullptr, c->d_jobsum, 0);
}

int qr_k_dbatch_apply(qr_ctx *c, size_t nleaves) {
  const BatchGeom g = batch_geom(c, nleaves);
  // (dbatch_prepared: a control call of its own -- the root's on unstaged trees, the last one of a
  // sequence that turned out too short -- has already made this step's batch: only apply it)
  if (g.fused && !c->dbatch_prepared) {
    QrTreeState *tin = c->dtree_cur, *tout = tin == c->d_tree ? c->d_tree2 : c->d_tree;
    hipLaunchKernelGGL(g.small ? k_decide_part<QR_BATCH_LDS_SMALL> : k_decide_part<QR_BATCH_LDS_LARGE>,
                       dim3(g.pg), dim3(128 * QR_BATCH), 0, c->stream, (const QrTreeState *)tin, tout, tin,
                       ++c->bepoch, c->dbatch_first ? 1 : 0, (int)nleaves, (u64)c->cur_minls, g.stage_nodes, g.rootn,
                       c->flocal, c->d_scalars, c->d_jobsum, c->d_featrec, c->d_featthr, (uint32_t)c->F,
                       c->sub_k ? 0 : 2, c->ncu, c->d_blocks, c->nblocks, c->d_lhist_wg, g.hg, c->d_lplan,
                       c->d_lscan_wg, c->d_bins_fm, (uint32_t)c->N, c->d_order[0], c->d_order[1],
                       (u64 *)c->d_bpart_state, c->d_lambda, (double *)nullptr, 0,
                       (const uint32_t *)c->d_hcnt_loc, (u64)(c->sub_k ? c->sub_k : c->Nglobal), doc_spec_docs(c));
    QR_CHECK(c, hipGetLastError());
    c->dtree_cur = tout;
    c->dbatch_first = false;
  } else {
    hipLaunchKernelGGL(k_partition_batch, dim3(g.pg), dim3(256), 0, c->stream, c->d_tree, c->d_lpart_wg,
                       c->d_bins_fm, (uint32_t)c->N, c->d_order[0], c->d_order[1], (u64 *)c->d_bpart_state,
                       c->d_lambda, (double *)nullptr, ++c->bepoch, 0);
    QR_CHECK(c, hipGetLastError());
    c->dbatch_prepared = false;
  }
  hipLaunchKernelGGL(k_hist_batch, dim3(g.hg), dim3(1024), hist_lds(c), c->stream, c->d_lhist_wg, c->d_blocks,
                     c->d_bins, c->d_order[0], c->d_order[1], c->d_lambda, c->d_scalars, (u64 *)c->d_lpartials,
                     c->d_lhistsum);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_bd_reduce, dim3(c->flocal, QR_BATCH), dim3(1024), 0, c->stream, c->d_lscan_wg,
                     (const u64 *)c->d_lpartials, c->flocal, c->d_xb, c->d_hcnt_loc, c->d_lhistsum, c->rank,
                     c->world);
  QR_CHECK(c, hipGetLastError());
  ++c->tree_step;
  return QR_OK;
}

int qr_k_dbatch_decide(qr_ctx *c, size_t nleaves, uint64_t minls, int final_call) {
  hipLaunchKernelGGL(k_bd_scan, dim3(c->flocal, QR_BATCH), dim3(256), 0, c->stream, c->d_lscan_wg,
                     (const long long *)c->d_xb, c->d_hsum, c->d_hcnt, c->flocal, c->d_thr_size, c->d_lf2gf,
                     c->d_scalars, c->d_featrec, c->d_thr, c->d_featthr, (u64)minls, c->d_jobsum, c->world);
  QR_CHECK(c, hipGetLastError());
  const BatchGeom g = batch_geom(c, nleaves);
  if (g.fused && !final_call) return QR_OK;  // (the control step: the next step's partition launch)
  const QrTreeState *tin = g.fused && c->dtree_cur != c->d_tree ? c->dtree_cur : (const QrTreeState *)nullptr;
  const int rc = launch_decide_batch(c, g, nleaves, minls, 0, c->d_tree, tin, c->d_jobsum, final_call);
  c->dtree_cur = c->d_tree;
  c->dbatch_prepared = true;  // (if the tree goes on, its next batch is ready in c->d_tree)
  return rc;
}

int qr_k_oblivious_fit(qr_ctx *c, size_t depth, uint64_t minls) {
  QR_DBG_CAPS(c);
  // (k_obl_plan numbers the leaves of the level the tree ends at: no k_finish launch)
  c->finish_in_decide = !c->obl_own_launches;
  const int maxnodes = (1 << (depth + 1)) - 1;
  if (c->wide || c->obl_own_launches) {
    hipLaunchKernelGGL(k_obl_reset, dim3((maxnodes + 255) / 256), dim3(256), 0, c->stream,
                       c->d_tree, maxnodes, (u64)minls);
    QR_CHECK(c, hipGetLastError());
  } else {
    c->obl_reset_nodes = maxnodes;  // (u8 bins: the last workgroup of the root scan launch does it)
  }
  // root histogram -> slot 0 (u8 bins: reduce + scan in one launch over feature-major partials,
  // as the batched leaf-wise path does)
  c->cur_minls = minls;
  int rc = launch_hist_scan(c, 1, !c->wide);
  if (rc) return rc;
  const size_t lds = c->wide ? 0 : hist_lds(c);
  const unsigned pgrid = (unsigned)c->lpart_cap, hgrid = (unsigned)c->lhist_cap;
  // every level: choose the split, plan the level, then ONE partition, ONE
  // histogram, ONE reduce and ONE scan launch for all of its nodes (the launches
  // are sized for the worst case; surplus workgroups leave at once)
  for (int level = 0; level < (int)depth; ++level) {
    const int nodes = 1 << level;
    const int last = level == (int)depth - 1;
    if (c->wide) {
      if ((rc = qr_k_wobl_fill(c, level))) return rc;
    } else {
      hipLaunchKernelGGL(k_obl_fill, dim3(c->flocal), dim3(256), 0, c->stream, c->d_tree, level,
                         c->d_hsum, c->d_hcnt, c->flocal, c->d_thr_size, c->d_lf2gf,
                         c->d_scalars, c->d_featrec);
      QR_CHECK(c, hipGetLastError());
    }
    hipLaunchKernelGGL(k_obl_plan, dim3(1), dim3(256), 0, c->stream, c->d_tree, level, last,
                       c->ncu, c->flocal, c->d_hcnt, c->wide ? c->d_wthr : c->d_thr, c->d_gf2lf,
                       c->d_blocks, c->nblocks, c->d_lhist_map, c->d_lpart_map,
                       (uint32_t)(c->sub_k ? c->sub_n : c->N),
                       c->d_featrec, c->d_scalars, c->wide ? c->d_woff : (const uint32_t *)nullptr,
                       c->wcells, (const qr_split_t *)nullptr, 1, (const uint32_t *)nullptr,
                       (const uint32_t *)nullptr, (u64)c->sub_k, c->sub_k ? 0 : 2);
    QR_CHECK(c, hipGetLastError());
    const unsigned pg = std::min<unsigned>(pgrid, (unsigned)(c->N / QR_PART_SLICE + nodes + 1));
    hipLaunchKernelGGL(k_partition_level, dim3(pg), dim3(256), 0, c->stream, c->d_tree,
                       c->d_lpart_map,
                       c->wide ? reinterpret_cast<const uint8_t *>(c->d_wbins) : c->d_bins_fm,
                       (uint32_t)c->N, c->d_order[0], c->d_order[1], (u64 *)c->d_lpart_state,
                       c->wide ? 1 : 0, (const uint32_t *)nullptr);
    QR_CHECK(c, hipGetLastError());
    if (last) break;  // ot.cc:127: no histograms for the leaves
    if (c->wide) {
      if ((rc = qr_k_wobl_hist(c, nodes))) return rc;
      continue;
    }
    // k_obl_plan keeps the total within ncu workgroups while nodes * nblocks <= 3/4 ncu
    const unsigned hg = std::min<unsigned>(
        hgrid, (unsigned)std::max(c->ncu, c->ncu / 4 + nodes * c->nblocks));
    hipLaunchKernelGGL(k_hist_level, dim3(hg), dim3(1024), lds, c->stream, c->d_tree,
                       c->d_lhist_map, c->d_blocks, c->nblocks, c->d_bins, c->d_order[0],
                       c->d_order[1], c->d_lambda, c->d_scalars, (u64 *)c->d_lpartials);
    QR_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(k_redscan_level, dim3(c->flocal, (unsigned)nodes), dim3(1024), 0, c->stream,
                       c->d_tree, c->d_blocks, c->nblocks, (const u64 *)c->d_lpartials, c->d_hsum,
                       c->d_hcnt, c->flocal, (long long *)nullptr, (uint32_t *)nullptr);
    QR_CHECK(c, hipGetLastError());
  }
  return QR_OK;
}

// ---- feature-sharded level-wise growth, phase by phase (the host puts the all-gather of
// the records and the all-reduce of mask + counts in between) ---------------------------
int qr_k_obl_begin(qr_ctx *c, size_t depth, uint64_t minls) {
  QR_DBG_CAPS(c);
  c->finish_in_decide = false;
  const int maxnodes = (1 << (depth + 1)) - 1;
  hipLaunchKernelGGL(k_obl_reset, dim3((maxnodes + 255) / 256), dim3(256), 0, c->stream, c->d_tree,
                     maxnodes, (u64)minls);
  QR_CHECK(c, hipGetLastError());
  return launch_hist_scan(c, 1);  // root histogram of the rank's features -> slot 0
}

int qr_k_obl_propose(qr_ctx *c, int level) {
  if (c->dmode) {
    // the exchange that precedes this call has summed the ranks' cells: the root's through
    // the leaf-wise path's buffer (k_scan takes the prefix), a level's children through the
    // level buffer
    if (level == 0) {
      const int rc = launch_scan(c, 1);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(k_level_finish_doc, dim3(c->flocal, 1u << (level - 1)), dim3(256), 0, c->stream,
                         c->d_tree, c->d_xlevel, c->d_hsum, c->d_hcnt, c->flocal);
      QR_CHECK(c, hipGetLastError());
    }
  }
  hipLaunchKernelGGL(k_obl_fill, dim3(c->flocal), dim3(256), 0, c->stream, c->d_tree, level, c->d_hsum,
                     c->d_hcnt, c->flocal, c->d_thr_size, c->d_lf2gf, c->d_scalars, c->d_featrec);
  QR_CHECK(c, hipGetLastError());
  if (c->dmode) return QR_OK;  // every rank holds every feature of the summed histograms: no records to exchange
  hipLaunchKernelGGL(k_obl_propose, dim3(1), dim3(64), 0, c->stream, c->d_tree, level, c->d_featrec,
                     c->flocal, c->d_recs_local);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_obl_mark(qr_ctx *c, int level) {
  const unsigned grid = (unsigned)((c->mask_words + 255) / 256);
  hipLaunchKernelGGL(k_obl_mark, dim3(grid), dim3(256), 0, c->stream, c->d_tree, level, c->d_recs_all,
                     c->world, c->d_gf2lf, (uint32_t)c->N, c->d_hcnt, c->flocal, c->d_bins_fm, c->d_mask,
                     (uint32_t)c->mask_words);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_obl_apply(qr_ctx *c, int level, int last) {
  const int nodes = 1 << level;
  hipLaunchKernelGGL(k_obl_plan, dim3(1), dim3(256), 0, c->stream, c->d_tree, level, last, c->ncu,
                     c->flocal, c->d_hcnt, c->d_thr, c->d_gf2lf, c->d_blocks, c->nblocks, c->d_lhist_map,
                     c->d_lpart_map, (uint32_t)(c->sub_k ? c->sub_n : c->N), c->d_featrec, c->d_scalars, (const uint32_t *)nullptr,
                     (size_t)0, c->dmode ? (const qr_split_t *)nullptr : c->d_recs_all, c->dmode ? 1 : c->world,
                     c->dmode ? (const uint32_t *)nullptr : c->d_mask + c->mask_words,
                     c->dmode ? c->d_hcnt_loc : (const uint32_t *)nullptr,
                     c->sub_k ? (u64)c->sub_k : (c->dmode ? (u64)c->Nglobal : (u64)0), c->sub_k ? 0 : 2);
  QR_CHECK(c, hipGetLastError());
  const unsigned pg = std::min<unsigned>((unsigned)c->lpart_cap, (unsigned)(c->N / QR_PART_SLICE + nodes + 1));
  hipLaunchKernelGGL(k_partition_level, dim3(pg), dim3(256), 0, c->stream, c->d_tree, c->d_lpart_map,
                     c->d_bins_fm, (uint32_t)c->N, c->d_order[0], c->d_order[1], (u64 *)c->d_lpart_state, 0,
                     c->dmode ? (const uint32_t *)nullptr : c->d_mask);
  QR_CHECK(c, hipGetLastError());
  if (last) return QR_OK;  // ot.cc:127: no histograms for the leaves
  const unsigned hg = std::min<unsigned>((unsigned)c->lhist_cap,
                                         (unsigned)std::max(c->ncu, c->ncu / 4 + nodes * c->nblocks));
  hipLaunchKernelGGL(k_hist_level, dim3(hg), dim3(1024), hist_lds(c), c->stream, c->d_tree, c->d_lhist_map,
                     c->d_blocks, c->nblocks, c->d_bins, c->d_order[0], c->d_order[1], c->d_lambda,
                     c->d_scalars, (u64 *)c->d_lpartials);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_redscan_level, dim3(c->flocal, (unsigned)nodes), dim3(1024), 0, c->stream, c->d_tree,
                     c->d_blocks, c->nblocks, (const u64 *)c->d_lpartials, c->d_hsum, c->d_hcnt, c->flocal,
                     c->dmode ? c->d_xlevel : (long long *)nullptr,
                     c->dmode ? c->d_hcnt_loc : (uint32_t *)nullptr);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_tree_finish(qr_ctx *c, int newton) {
  const unsigned sgrid = (unsigned)((c->N + QR_SLICE - 1) / QR_SLICE);
  if (!c->finish_in_decide) {  // (batched, staged trees: the last control call numbered the leaves)
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, c->stream, c->d_tree);
    QR_CHECK(c, hipGetLastError());
  }
  // trees of up to QR_LDOC leaves: sums in document order (k_leaf_sums_doc), the leaf bytes
  // kept for the score update
  const bool doc_path = c->leaf_cap >= 1 && c->leaf_cap <= QR_LDOC && c->d_leafb && !c->leaf_by_position;
  const bool walk = c->flocal == (int)c->F;
  const double *wgt = newton ? c->d_weight : (const double *)nullptr;
  const uint8_t *present = c->sub_k ? c->d_present : (const uint8_t *)nullptr;
  c->leafb_valid = false;
  if (doc_path && walk) {
    hipLaunchKernelGGL(k_leaf_sums_doc<true>, dim3(sgrid), dim3(256), 0, c->stream, c->d_tree,
                       c->wide ? reinterpret_cast<const uint8_t *>(c->d_wbins) : c->d_bins_fm, (uint32_t)c->N,
                       c->d_gf2lf, c->wide ? 1 : 0, c->d_leafb, c->d_lambda, wgt, present, c->d_leafpart);
    c->leafb_valid = true;  // every document has walked
  } else if (doc_path) {
    hipLaunchKernelGGL(k_leaf_ids_scatter, dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream,
                       c->d_tree, c->d_order[0], c->d_order[1], c->d_leafb);
    QR_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(k_leaf_sums_doc<false>, dim3(sgrid), dim3(256), 0, c->stream, c->d_tree,
                       (const uint8_t *)nullptr, (uint32_t)c->N, (const int32_t *)nullptr, 0, c->d_leafb,
                       c->d_lambda, wgt, present, c->d_leafpart);
    c->leafb_valid = !c->sub_k;  // (a sample's lists do not hold every document)
  } else {
    hipLaunchKernelGGL(k_leaf_sums, dim3(sgrid), dim3(256), 0, c->stream, c->d_tree,
                       c->d_order[0], c->d_order[1], c->d_lambda, wgt, c->d_leafpart);
  }
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_leaf_final, dim3(1), dim3(1024), 0, c->stream, c->d_tree,
                     c->d_leafpart, newton, c->dmode, c->d_xleaf, c->rank, c->world,
                     (int)(2 * c->cur_nleaves), &c->d_pin->tree, (long long)(c->dmode ? c->nodes_seq : ++c->nodes_seq),
                     doc_path ? (uint32_t)sgrid : 0u);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

int qr_k_tree_leaves_global(qr_ctx *c, int newton) {
  hipLaunchKernelGGL(k_leaf_global, dim3(1), dim3(1024), 0, c->stream, c->d_tree,
                     c->d_xleaf, newton, c->world, (int)(2 * c->cur_nleaves), &c->d_pin->tree,
                     (long long)++c->nodes_seq);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// qr_debug_check: drains the device, reports a failed launch and -- in a -DQR_DEBUG_CHECKS build --
// the first bounds violation a growth kernel recorded (and skipped)
int qr_k_debug_check(qr_ctx *c) {
  QR_CHECK(c, hipDeviceSynchronize());
  QR_CHECK(c, hipGetLastError());
#ifdef QR_DEBUG_CHECKS
  unsigned long long w[4] = {0, 0, 0, 0};
  QR_CHECK(c, hipMemcpyFromSymbol(w, HIP_SYMBOL(qr_dbg_word), sizeof(w)));
  if (w[0]) {
    const unsigned long long zero[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(qr_dbg_word), zero, sizeof(zero));
    char msg[256];
    snprintf(msg, sizeof(msg), "QR_DEBUG_CHECKS: bounds violation, check %llu in workgroup %llu, values 0x%llx 0x%llx (%llu in all)",
             w[0] >> 32, w[0] & 0xffffffffull, w[2], w[3], w[1]);
    c->err = msg;
    return QR_ERR_STATE;
  }
  QR_DBG_CAPS(c);
#endif
  return QR_OK;
}

// a score update left to the next lambda pass (qr_k_lambda), launched after all: somebody
// else wants the scores first
int qr_k_scores_flush(qr_ctx *c) {
  if (!c->lazy_scores) return QR_OK;
  c->lazy_scores = false;
  hipLaunchKernelGGL(k_score_update_leaf, dim3((unsigned)((c->N + 1023) / 1024)), dim3(256), 0, c->stream,
                     c->d_tree, c->d_leafb, (uint32_t)c->N, c->lazy_shrinkage, c->d_scores);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// repeat: the same update enqueued again behind a tree that was carried on (its first launch
// found the tree incomplete and left at once; a pending lazy update was never launched at all)
int qr_k_scores_update(qr_ctx *c, double shrinkage, bool repeat) {
  const unsigned grid = (unsigned)((c->N + 255) / 256);
  // One GPU, every document's leaf known, no sample: the update waits for the lambda pass of
  // the next iteration, which reads every score anyway (mart.cc:464-467 -> lambdamart.cc:70):
  // one launch and 16 N bytes less per iteration.  Whoever else looks at the scores first
  // (tree_settle) has it launched after all.  QR_LAZY_SCORES=0: always at once.
  if (c->lazy_scores && !repeat) {  // (a second update behind one that nobody consumed: that one first)
    const int frc = qr_k_scores_flush(c);
    if (frc) return frc;
  }
  if (c->leafb_valid && !c->no_lazy_scores && c->world == 1 && !c->dmode && !c->sub_k) {
    c->lazy_scores = true;
    c->lazy_shrinkage = shrinkage;
  } else
  if (c->leafb_valid) {
    // the leaf of every document is known (k_leaf_sums_doc): no walk, no lists
    hipLaunchKernelGGL(k_score_update_leaf, dim3((unsigned)((c->N + 1023) / 1024)), dim3(256), 0, c->stream,
                       c->d_tree, c->d_leafb, (uint32_t)c->N, shrinkage, c->d_scores);
  } else if (c->flocal == (int)c->F) {
    // every feature is here (single GPU, document-sharded): document-order walk;
    // also what a --subsample iteration needs (its leaves hold the sample only,
    // every training document is updated, mart.cc:345)
    hipLaunchKernelGGL(k_score_update_walk, dim3((unsigned)((c->N + 1023) / 1024)), dim3(256), 0, c->stream, c->d_tree,
                       c->wide ? reinterpret_cast<const uint8_t *>(c->d_wbins) : c->d_bins_fm,
                       (uint32_t)c->N, c->d_gf2lf, shrinkage, c->d_scores, c->wide ? 1 : 0);
  } else if (c->sub_k) {
    // a feature-sharded rank under --subsample: the leaves hold the sample only and its bins
    // cover its own features only, but the raw rows are replicated: walk them (mart.cc:345)
    hipLaunchKernelGGL(k_valid_update, dim3(grid), dim3(256), 0, c->stream, c->d_tree, c->d_raw,
                       (uint32_t)c->N, (uint32_t)c->F, shrinkage, c->d_scores);
  } else {
    hipLaunchKernelGGL(k_score_update, dim3(grid), dim3(256), 0, c->stream, c->d_tree,
                       c->d_order[0], c->d_order[1], shrinkage, c->d_scores);
  }
  QR_CHECK(c, hipGetLastError());
  if (c->vN) {
    const unsigned vgrid = (unsigned)((c->vN + 255) / 256);
    hipLaunchKernelGGL(k_valid_update, dim3(vgrid), dim3(256), 0, c->stream,
                       c->d_tree, c->d_vraw, (uint32_t)c->vN, (uint32_t)c->F,
                       shrinkage, c->d_vscores);
    QR_CHECK(c, hipGetLastError());
  }
  return QR_OK;
}
