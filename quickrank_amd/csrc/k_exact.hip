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
// A segment is cut into TILES (4096 entries for the scan, 8192 for the partition), one workgroup
// each, in ONE launch: a tile publishes its own aggregate (of gradients / of left-going entries) as
// a self-validating 8-byte word {launch epoch : 16, value : 48} the moment it knows it, walks back
// over its predecessors' words until it meets an inclusive prefix, and publishes its own (decoupled
// look-back; tiles are dispatched feature-fastest, so the tiles a walk meets are resident or done)
// -- every CU works on every node, and a document's gradient is gathered once.
// k_xflag (go-left bytes), k_xpart (F x tiles), k_xscan (F x tiles), k_xbest (F: first maximum over
// a feature's tiles).
// qr_tree_fit searches a node's split when the loop POPS it (qr_k_exact_fit; k_tree.hip k_xpop /
// k_xapply), which is the reference's own order (rt.cc:58-90): the leaves a tree ends with are
// never searched, and the children's gradient totals the gain needs come from the parent's search
// (the winner's cumulative sum is the left child's).  Per tree the lists move pi N F entries
// through the scans and pi N F through the partitions (pi = documents partitioned / N, ~3.7 on the
// stand-in).  The phase API (qr_tree_begin / decide / apply; QR_X_EAGER=1) searches both children
// behind every split: (1 + pi) N F entries and a pass for the totals (k_xtotal).
// Measured on the MSLR-shaped stand-in (712,928 x 136, 67M slots): 22.4 ms per iteration on slot-
// indexed histograms (round 3), 3.1-3.4 eager, 2.6 with the search at the pop.
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "qr_internal.h"
#include "qr_wave.h"
#include "qr_dev.h"

#define QR_X_E 8u                      /* entries per thread and chunk */
#define QR_X_CHUNK (1024u * QR_X_E)    /* entries per chunk of a 1024-thread workgroup (k_xpart) */
#ifndef QR_XS_T
#define QR_XS_T 512u                   /* k_xscan: threads per workgroup = entries per tile / 8 (measured: 256 / 512 / 1024 threads) */
#endif
#define QR_XS_W (QR_XS_T / 64u)
#define QR_XS_CHUNK (QR_XS_T * QR_X_E)

__device__ __forceinline__ uint32_t x_id(const u64 e) { return (uint32_t)e; }
// the lists are read once per launch: streamed past the caches (non-temporal), so that what IS
// re-read -- the gradients, the go-left bytes -- keeps its lines (-DQR_X_NO_NT: plain loads, A/B)
__device__ __forceinline__ u64 x_load(const u64 *p) {
#ifdef QR_X_NO_NT
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}
__device__ __forceinline__ void x_store(u64 *p, const u64 v) {
#ifdef QR_X_NO_NT
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}
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

// ---------------------------------------------------------------------------
// k_xflag: the go-left byte of every document of the node being split, from the split feature's
// own segment (sorted by slot: `slot <= t*`  <=>  `x <= threshold`, rt.cc:327-334).  A byte per
// document: the 136 partition workgroups then gather from N bytes instead of the feature's
// 4 N-byte column.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_xflag(const QrTreeState *__restrict__ ts, const u64 *__restrict__ xroot,
                                               const u64 *__restrict__ x0, const u64 *__restrict__ x1, const size_t N,
                                               uint8_t *__restrict__ goleft, const int lazy) {
  // (lazy: the split search at the pop -- the last split's children are never walked, ts->xs_last)
  const QrSplitDesc d = ts->desc;
  if (!d.active || d.owner_local < 0 || (lazy && ts->xs_last)) return;
  const u64 *src = x_lists(d.src_buf, xroot, x0, x1) + (size_t)d.owner_local * N + d.begin;
  const uint32_t n = d.end - d.begin;
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const u64 e = src[p];
    goleft[x_id(e)] = x_slot(e) <= d.thr_id ? 1 : 0;
  }
}

// ... on a FEATURE-sharded context (round 6): only the owner of the winning feature holds its
// segment; every rank holds the go-left bits the owner published and the ranks all-reduced
// (k_mask: bit p = position p of the node's segment of the document-order list, the one
// k_partition has just read), so every rank -- the owner too -- takes its bytes from there.
__global__ __launch_bounds__(256) void k_xflag_mask(const QrTreeState *__restrict__ ts,
                                                    const uint32_t *__restrict__ order0,
                                                    const uint32_t *__restrict__ order1,
                                                    const uint32_t *__restrict__ mask, uint8_t *__restrict__ goleft) {
  const QrSplitDesc d = ts->desc;
  if (!d.active) return;
  const uint32_t n = d.end - d.begin;
  const uint32_t *order = d.src_buf == 0 ? order0 : order1;
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const uint32_t id = d.src_buf == 2 ? d.begin + p : order[d.begin + p];
    goleft[id] = (uint8_t)((mask[p >> 5] >> (p & 31)) & 1u);
  }
}

// ---------------------------------------------------------------------------
// k_xpart: the split being applied (ts->desc), feature blockIdx.x's segment of the parent,
// stable partition into the children's segments of the destination lists.  Entries two chunks
// ahead and the go-left bytes one chunk ahead are in flight while a chunk is placed.
// ---------------------------------------------------------------------------
// Decoupled look-back (one poller per tile).  A tile owns three 8-byte words, each self-validating
// {launch epoch : 16, payload : 48} and written once per launch with a relaxed agent-scope store:
//   [0] its own AGGREGATE, as soon as it knows it;
//   [1], [2] its INCLUSIVE PREFIX (low 32 bits / the rest, signed), once its predecessors' are in.
// The poller walks back from the tile before it: a prefix ends the walk, an aggregate is added and
// the walk goes on, a word of an older launch is waited for.  Tiles are dispatched feature-fastest
// (blockIdx.x = feature), so the tiles a walk meets are a dispatch row or two older: resident or
// done -- the walk cannot wait for a workgroup that is not running yet.
#define QR_X_PUBW 3u
__device__ __forceinline__ bool x_word(const u64 *w, const u64 epoch, long long *payload) {
  const u64 v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *payload = ((long long)(v << 16)) >> 16;  // sign-extended 48 bits
  return (v >> 48) == epoch;
}
__device__ __forceinline__ void x_put(u64 *w, const u64 epoch, const long long payload) {
  __hip_atomic_store(w, (epoch << 48) | ((u64)payload & 0xFFFFFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void x_publish_aggregate(u64 *row, const uint32_t tile, const u64 epoch, const long long v) {
  x_put(row + (size_t)tile * QR_X_PUBW, epoch, v);
}
__device__ __forceinline__ void x_publish_prefix(u64 *row, const uint32_t tile, const u64 epoch, const long long v) {
  x_put(row + (size_t)tile * QR_X_PUBW + 1, epoch, (long long)(uint32_t)v);
  x_put(row + (size_t)tile * QR_X_PUBW + 2, epoch, v >> 32);
}
// sum of the aggregates of tiles [0, tile): ONE lane calls this
__device__ __forceinline__ long long x_look_back(const u64 *row, const uint32_t tile, const u64 epoch) {
  long long sum = 0;
  for (uint32_t i = tile; i-- > 0;) {
    const u64 *w = row + (size_t)i * QR_X_PUBW;
    for (;;) {
      long long lo, hi, ag;
      if (x_word(w + 1, epoch, &lo) && x_word(w + 2, epoch, &hi)) return sum + ((hi << 32) | (lo & 0xFFFFFFFFll));
      if (x_word(w, epoch, &ag)) {
        sum += ag;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  return sum;
}

__global__ __launch_bounds__(1024) void k_xpart(const QrTreeState *__restrict__ ts, const u64 *__restrict__ xroot,
                                                u64 *__restrict__ x0, u64 *__restrict__ x1, const size_t N,
                                                const uint8_t *__restrict__ goleft, u64 *__restrict__ pub,
                                                const uint32_t tiles, const u64 epoch, const int lazy,
                                                const int remote_owner = 0) {
  // (remote_owner: a feature-sharded rank partitions its own features' segments whoever owns the
  // split feature -- the go-left bytes came with the reduced mask, k_xflag_mask)
  // Tile blockIdx.y of feature blockIdx.x.  Entry k of thread t sits at position c0 + k * 1024 + t:
  // a wave's load is 512 contiguous bytes.  The rank of an entry among the tile's left-going ones:
  // lanes before it in its wave (ballot), waves before it in its slab k and the slabs before (one
  // scan of the 8 x 16 wave counts by the first wave); among the node's: + the tiles before.
  __shared__ uint32_t sh_c[QR_X_E * 16], sh_p[QR_X_E * 16 + 1];
  __shared__ uint32_t sh_before;
  const QrSplitDesc d = ts->desc;
  if (!d.active || (d.owner_local < 0 && !remote_owner) || (lazy && ts->xs_last)) return;
  const uint32_t n = d.end - d.begin;
  const uint32_t tile = blockIdx.y;
  const uint32_t c0 = tile * QR_X_CHUNK;
  if (c0 >= n) return;
  const size_t fo = (size_t)blockIdx.x * N;
  const u64 *src = x_lists(d.src_buf, xroot, x0, x1) + fo + d.begin;
  u64 *dst = (d.dst_buf == 0 ? x0 : x1) + fo;
  u64 *mypub = pub + (size_t)blockIdx.x * tiles * QR_X_PUBW;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  u64 e[QR_X_E];
  uint32_t fl[QR_X_E];
#pragma unroll
  for (uint32_t k = 0; k < QR_X_E; ++k) {
    const uint32_t p = c0 + k * 1024u + threadIdx.x;
    e[k] = x_load(src + (p < n ? p : n - 1));  // (clamped: unconditional loads)
  }
#pragma unroll
  for (uint32_t k = 0; k < QR_X_E; ++k) fl[k] = goleft[x_id(e[k])];
  uint32_t lp[QR_X_E];  // left-going lanes before this one in its wave and slab
  bool left[QR_X_E];
#pragma unroll
  for (uint32_t k = 0; k < QR_X_E; ++k) {
    left[k] = c0 + k * 1024u + threadIdx.x < n && fl[k] != 0;
    const unsigned long long m = __ballot(left[k]);
    lp[k] = (uint32_t)__popcll(m & lt);
    if (lane == 0) sh_c[k * 16 + wave] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (wave == 0) {  // exclusive prefix of the 128 counts in (slab, wave) order = position order
    const uint32_t a0 = sh_c[2 * lane], a1 = sh_c[2 * lane + 1];
    const uint32_t inc = wave_scan_u32(a0 + a1);
    sh_p[2 * lane] = inc - a0 - a1;
    sh_p[2 * lane + 1] = inc - a1;
    const uint32_t ltot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    if (lane == 0) {  // the tile's aggregate, the walk over its predecessors, its inclusive prefix
      x_publish_aggregate(mypub, tile, epoch, (long long)ltot);
      const long long bef = x_look_back(mypub, tile, epoch);
      x_publish_prefix(mypub, tile, epoch, bef + (long long)ltot);
      sh_before = (uint32_t)bef;
    }
  }
  __syncthreads();
  const uint32_t before = sh_before;
  const uint32_t lcur = d.begin + before, rcur = d.begin + d.lcount + (c0 - before);
#pragma unroll
  for (uint32_t k = 0; k < QR_X_E; ++k) {
    const uint32_t pc = k * 1024u + threadIdx.x;  // entries of the tile before this one
    if (c0 + pc < n) {
      const uint32_t lb = sh_p[k * 16 + wave] + lp[k];  // ... of which go left
      if (left[k])
        x_store(dst + lcur + lb, e[k]);
      else
        x_store(dst + rcur + (pc - lb), e[k]);
    }
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
//
// With every distinct value a threshold nearly every entry ends a run, and the gain of rt.cc:278
// is two IEEE f64 divisions (~80 vector instructions): evaluated for all of them it was most of
// the kernel's time.  So a candidate is first PRICED -- the same expression with the hardware
// reciprocal, good to ~1e-7 -- and only one that comes within 1e-5 of the best score the
// workgroup has seen so far gets the exact evaluation that decides (a candidate that is not the
// maximum to 1e-5 cannot be the first maximum; ties are within it).  Entries two chunks ahead and
// gradients one chunk ahead are in flight while a chunk is evaluated; one barrier pair per chunk.
// ---------------------------------------------------------------------------
struct XTileBest {
  double score;
  uint32_t t, lc;
  long long cs;   // the cumulative fixed-point sum at the candidate: the left child's total
};

__global__ __launch_bounds__(QR_XS_T) void k_xscan(
    const QrTreeState *__restrict__ ts, const int mode, const uint32_t rootn, const u64 *__restrict__ xroot,
    const u64 *__restrict__ x0, const u64 *__restrict__ x1, const size_t N, const double *__restrict__ lambda,
    const QrScalars *__restrict__ scal, const long long *__restrict__ tot, const uint32_t *__restrict__ woff,
    const int flocal, u64 *__restrict__ pub, const uint32_t tiles, const u64 epoch,
    unsigned long long *__restrict__ gbest, XTileBest *__restrict__ tbest, const u64 minls_root) {
  // Tile blockIdx.y of feature blockIdx.x of node blockIdx.z.  Entry k of thread t sits at position
  // c0 + k * QR_XS_T + t (coalesced loads, as k_xpart): QR_X_E slabs of QR_XS_T consecutive
  // positions, a wave scans its 64 positions of every slab on DPP, the first wave scans the
  // QR_X_E x QR_XS_W wave totals in position order, publishes the tile's total and collects the
  // totals of the tiles before it.
  __shared__ long long sh_ws[QR_X_E * QR_XS_W], sh_pre[QR_X_E * QR_XS_W];
  constexpr uint32_t NWT = QR_X_E * QR_XS_W;   // wave totals of a tile: 64 (512 threads) or 128 (1024)
  __shared__ long long sh_carry;
  __shared__ uint32_t sh_wf[QR_X_E * QR_XS_W + 1];  // the waves' first slots per slab, then the next tile's
  __shared__ double sh_price[QR_XS_W];
  __shared__ Best sh_b[QR_XS_W];
  __shared__ uint32_t sh_lc[QR_XS_W];
  __shared__ long long sh_cs[QR_XS_W];
  static_assert(NWT <= 64 || NWT == 128, "the first wave scans one or two wave totals per lane");
  const int lf = blockIdx.x, which = blockIdx.z;
  const uint32_t tile = blockIdx.y;
  uint32_t begin = 0, n = rootn;
  const u64 *lst = xroot;
  int tot_at = which;
  if (mode == 1) {
    const QrSplitDesc d = ts->desc;
    if (!d.active) return;
    begin = which == 0 ? d.begin : d.begin + d.lcount;
    n = which == 0 ? d.lcount : d.end - d.begin - d.lcount;
    lst = d.dst_buf == 0 ? x0 : x1;
  } else if (mode == 2) {  // the node the loop has just popped (k_xpop); `tot` is indexed by node
    tot_at = ts->xs_node;
    if (tot_at < 0) return;
    begin = ts->xs_begin;
    n = ts->xs_n;
    lst = x_lists(ts->xs_buf, xroot, x0, x1);
  }
  const uint32_t c0 = tile * QR_XS_CHUNK;
  if (c0 >= n) return;
  const u64 *src = lst + (size_t)lf * N + begin;
  const size_t row = (size_t)which * flocal + lf;
  u64 *mypub = pub + row * tiles * QR_X_PUBW;
  const u64 minls = (mode != 1 && minls_root != ~0ull) ? minls_root : ts->minls;
  const double scale = scal->scale, inv_scale = scal->inv_scale;
  const long long S = tot[tot_at];
  const double s_d = (double)S * inv_scale;
  const uint32_t tsize = woff[lf + 1] - woff[lf];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  u64 e[QR_X_E];
#pragma unroll
  for (uint32_t k = 0; k < QR_X_E; ++k) {
    const uint32_t p = c0 + k * QR_XS_T + threadIdx.x;
    e[k] = x_load(src + (p < n ? p : n - 1));
  }
  const bool more = c0 + QR_XS_CHUNK < n;
  const u64 enext = src[more ? c0 + QR_XS_CHUNK : n - 1];  // (the run-end test of the tile's last entry)
  double lam[QR_X_E];
#pragma unroll
  for (uint32_t k = 0; k < QR_X_E; ++k) lam[k] = lambda[x_id(e[k])];
  const unsigned long long g0 = __hip_atomic_load(gbest + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  long long inc[QR_X_E];  // inclusive prefix over the wave's 64 positions of slab k
#pragma unroll
  for (uint32_t k = 0; k < QR_X_E; ++k) {
    const long long q = c0 + k * QR_XS_T + threadIdx.x < n ? quantize(lam[k] * scale) : 0;
    inc[k] = wave_scan_i64(q);
    if (lane == 63) sh_ws[k * QR_XS_W + wave] = inc[k];
    if (lane == 0) sh_wf[k * QR_XS_W + wave] = x_slot(e[k]);
  }
  if (threadIdx.x == 0) sh_wf[QR_X_E * QR_XS_W] = more ? x_slot(enext) : 0xFFFFFFFFu;
  __syncthreads();
  if (wave == 0) {  // the wave totals in (slab, wave) order = position order: one or two per lane
    long long iv;
    if (NWT <= 64) {
      const long long v = lane < NWT ? sh_ws[lane < NWT ? lane : 0] : 0;
      iv = wave_scan_i64(v);
      if (lane < NWT) sh_pre[lane] = iv - v;
    } else {
      const long long a0 = sh_ws[2 * lane], a1 = sh_ws[2 * lane + 1];
      iv = wave_scan_i64(a0 + a1);
      sh_pre[2 * lane] = iv - a0 - a1;
      sh_pre[2 * lane + 1] = iv - a1;
    }
    const long long total = readlane_i64(iv, 63);
    if (lane == 0) {  // the tile's aggregate, the walk over its predecessors, its inclusive prefix
      x_publish_aggregate(mypub, tile, epoch, total);
      const long long bef = x_look_back(mypub, tile, epoch);
      x_publish_prefix(mypub, tile, epoch, bef + total);
      sh_carry = bef;
    }
  }
  __syncthreads();
  const long long carry = sh_carry;
  // price every candidate (reciprocals instead of divisions: good to ~1e-14), then evaluate exactly
  // only those within 1e-11 of the best price of the tile or of the best exact score an earlier
  // tile has published.  (The margin matters: while many gradients are zero, runs of neighbouring
  // candidates differ by 1 / lc ~ 1e-6 relative -- with a margin of 1e-5 three percent of all
  // candidates took the exact path, and with them nearly every wave: 1.8 ms for the root.)
  double price[QR_X_E];
  long long cs[QR_X_E];
  double pmax = -1.0;
#pragma unroll
  for (uint32_t k = 0; k < QR_X_E; ++k) {
    const uint32_t p = c0 + k * QR_XS_T + threadIdx.x;
    const uint32_t t = x_slot(e[k]);
    // the slot behind this entry: the next lane's, the next wave's first, the next slab's, the next tile's
    uint32_t tn = (uint32_t)__shfl_down((int)t, 1);
    if (lane == 63) tn = sh_wf[k * QR_XS_W + wave + 1];
    const uint32_t lc = p + 1, rc = n - lc;
    price[k] = -1.0;
    cs[k] = carry + sh_pre[k * QR_XS_W + wave] + inc[k];
    if (p < n && (lc == n || tn != t) && t < tsize && lc >= minls && rc >= minls) {
      const double lsum = (double)cs[k] * inv_scale, rsum = s_d - lsum;
      // (hardware reciprocal + two Newton steps: a few ulp; the exact evaluation keeps IEEE divisions)
      const double dl = (double)lc, dr = (double)rc;
      double il = __builtin_amdgcn_rcp(dl), ir = __builtin_amdgcn_rcp(dr);
      il = fma(fma(-dl, il, 1.0), il, il);
      ir = fma(fma(-dr, ir, 1.0), ir, ir);
      il = fma(fma(-dl, il, 1.0), il, il);
      ir = fma(fma(-dr, ir, 1.0), ir, ir);
      const double pr = lsum * lsum * il + rsum * rsum * ir;
      price[k] = pr >= 0.0 ? pr : -1.0;  // (a NaN -- 0 / 0 -- is no candidate, as in slot_gain)
      pmax = price[k] > pmax ? price[k] : pmax;
    }
  }
  pmax = wave_max(pmax);
  if (lane == 0) sh_price[wave] = pmax;
  __syncthreads();
#pragma unroll
  for (uint32_t w = 0; w < QR_XS_W; ++w) pmax = sh_price[w] > pmax ? sh_price[w] : pmax;
  const double gb = __longlong_as_double((long long)g0);
  const double prune = (gb > pmax ? gb : pmax) * (1.0 - 1e-11);
  Best best;
  best.score = -1.0;
  best.t = 0xFFFFFFFFu;
  uint32_t best_lc = 0;
  long long best_cs = 0;
#pragma unroll
  for (uint32_t k = 0; k < QR_X_E; ++k) {
    if (price[k] >= 0.0 && price[k] >= prune) {
      const uint32_t lc = c0 + k * QR_XS_T + threadIdx.x + 1;
      const Best v = slot_gain(cs[k], lc, S, n, x_slot(e[k]), tsize, minls, inv_scale);
      if (v.score > best.score) {  // ascending positions per thread: strict > keeps the first
        best = v;
        best_lc = lc;
        best_cs = cs[k];
      }
    }
  }
  // first maximum over the tile: highest score, then lowest slot
  const double m = wave_max(best.score);
  const uint32_t tmin = wave_min_u32(best.score == m && best.t != 0xFFFFFFFFu ? best.t : 0xFFFFFFFFu);
  const unsigned long long holder = __ballot(best.score == m && best.t == tmin && tmin != 0xFFFFFFFFu);
  const uint32_t wlc = holder ? (uint32_t)__builtin_amdgcn_readlane((int)best_lc, __ffsll((long long)holder) - 1) : 0u;
  const long long wcs = holder ? readlane_i64(best_cs, __ffsll((long long)holder) - 1) : 0ll;
  if (lane == 0) {
    Best w;
    w.score = tmin != 0xFFFFFFFFu ? m : -1.0;
    w.t = tmin;
    sh_b[wave] = w;
    sh_lc[wave] = wlc;
    sh_cs[wave] = wcs;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Best r = sh_b[0];
    uint32_t lc = sh_lc[0];
    long long wc = sh_cs[0];
    for (int i = 1; i < (int)QR_XS_W; ++i) {
      const Best o = sh_b[i];
      if (o.score > r.score || (o.score == r.score && o.t < r.t)) {
        r = o;
        lc = sh_lc[i];
        wc = sh_cs[i];
      }
    }
    XTileBest tb;
    tb.score = r.t == 0xFFFFFFFFu ? -1.0 : r.score;
    tb.t = r.t;
    tb.lc = lc;
    tb.cs = wc;
    tbest[row * tiles + tile] = tb;
    if (tb.score > 0.0) atomicMax(gbest + row, (unsigned long long)__double_as_longlong(tb.score));
  }
}

// first maximum over a feature's tiles (ascending tiles = ascending slots) -> featrec / featthr
__global__ __launch_bounds__(64) void k_xbest(const QrTreeState *__restrict__ ts, const int mode, const uint32_t rootn,
                                              const XTileBest *__restrict__ tbest, const uint32_t tiles,
                                              const uint32_t *__restrict__ woff, const int flocal,
                                              const int32_t *__restrict__ lf2gf, const float *__restrict__ thr,
                                              qr_split_t *__restrict__ featrec, float *__restrict__ featthr,
                                              long long *__restrict__ xcs) {
  const int lf = blockIdx.x, which = blockIdx.y;
  uint32_t n = rootn;
  if (mode == 1) {
    const QrSplitDesc d = ts->desc;
    if (!d.active) return;
    n = which == 0 ? d.lcount : d.end - d.begin - d.lcount;
  } else if (mode == 2) {
    if (ts->xs_node < 0) return;
    n = ts->xs_n;
  }
  const size_t row = (size_t)which * flocal + lf;
  const uint32_t used = (n + QR_XS_CHUNK - 1) / QR_XS_CHUNK;
  Best b;
  b.score = -1.0;
  b.t = 0xFFFFFFFFu;
  uint32_t lc = 0;
  long long wcs = 0;
  for (uint32_t i = threadIdx.x; i < used; i += 64) {  // ascending tiles per lane: strict > keeps the first
    const XTileBest v = tbest[row * tiles + i];
    if (v.t != 0xFFFFFFFFu && v.score > b.score) {
      b.score = v.score;
      b.t = v.t;
      lc = v.lc;
      wcs = v.cs;
    }
  }
  const double m = wave_max(b.score);
  const uint32_t tmin = wave_min_u32(b.score == m && b.t != 0xFFFFFFFFu ? b.t : 0xFFFFFFFFu);
  const unsigned long long holder = __ballot(b.score == m && b.t == tmin && tmin != 0xFFFFFFFFu);
  if (threadIdx.x == 0) {
    const bool none = !holder;
    const uint32_t wlc = none ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)lc, __ffsll((long long)holder) - 1);
    if (xcs) xcs[lf] = none ? 0ll : readlane_i64(wcs, __ffsll((long long)holder) - 1);
    qr_split_t *o = &featrec[row];
    o->score = none ? -1.0 : m;
    o->feature = none ? 0xFFFFFFFFu : (uint32_t)lf2gf[lf];
    o->thr_id = none ? 0xFFFFFFFFu : tmin;
    o->lcount = none ? 0 : wlc;
    o->rcount = none ? 0 : n - wlc;
    featthr[row] = none ? 0.f : thr[woff[lf] + tmin];
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
  if (c->d_xgoleft) (void)hipFree(c->d_xgoleft);
  if (c->d_xpub) (void)hipFree(c->d_xpub);
  if (c->d_xtbest) (void)hipFree(c->d_xtbest);
  if (c->d_xnode_tot) (void)hipFree(c->d_xnode_tot);
  c->d_xnode_tot = c->d_xcs = nullptr;
  c->d_xgbest = nullptr;
  c->d_xroot = c->d_xlist[0] = c->d_xlist[1] = nullptr;
  c->d_xpub = nullptr;
  c->d_xtot = nullptr;
  c->d_xgoleft = nullptr;
  c->d_xtbest = nullptr;
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
  // tiles a segment can have, and the words they publish: [2 nodes][F][scan tiles] then [F][partition tiles]
  c->xtiles_s = (uint32_t)((N + QR_XS_CHUNK - 1) / QR_XS_CHUNK);
  c->xtiles_p = (uint32_t)((N + QR_X_CHUNK - 1) / QR_X_CHUNK);
  const size_t npub = (2 * FL * c->xtiles_s + FL * c->xtiles_p) * QR_X_PUBW;
  if (hipMalloc((void **)&d_iota, N * 4) != hipSuccess || hipMalloc((void **)&d_ks, N * 4) != hipSuccess ||
      hipMalloc((void **)&d_vs, N * 4) != hipSuccess)
    return fail("allocating the sort scratch of the pre-sorted lists failed");
  if (hipMalloc((void **)&c->d_xroot, FL * N * 8) != hipSuccess || hipMalloc((void **)&c->d_xlist[0], FL * N * 8) != hipSuccess ||
      hipMalloc((void **)&c->d_xlist[1], FL * N * 8) != hipSuccess ||
      hipMalloc((void **)&c->d_xtot, (2 + 2 * FL + 2) * 8) != hipSuccess ||  // [2] totals, then [2][F] best score bits (+ an experiment's counter)
      hipMalloc((void **)&c->d_xgoleft, N + 16) != hipSuccess || hipMalloc((void **)&c->d_xpub, npub * 8) != hipSuccess ||
      hipMalloc((void **)&c->d_xtbest, 2 * FL * c->xtiles_s * sizeof(XTileBest)) != hipSuccess ||
      hipMalloc((void **)&c->d_xnode_tot, (QR_MAXNODES + FL) * 8) != hipSuccess)  // per-node totals, then [F] winners' sums
    return fail("allocating the pre-sorted lists failed");
  c->d_xcs = c->d_xnode_tot + QR_MAXNODES;
  c->d_xgbest = c->d_xtot + 2;
  if (hipMemset(c->d_xpub, 0, npub * 8) != hipSuccess || hipMemset(c->d_xtot, 0, (2 + 2 * FL + 2) * 8) != hipSuccess)
    return fail("clearing the tiles' words failed");
  c->xepoch = 0;
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
  const unsigned nodes = root_mode ? 1 : 2;
  u64 *pub_scan = (u64 *)c->d_xpub, *pub_part = (u64 *)c->d_xpub + (size_t)2 * F * c->xtiles_s * QR_X_PUBW;
  // the launch epoch the tiles' words carry (16 bits, never 0: the words start as zeros)
  if (++c->xepoch > 0xFFFFu) {
    QR_CHECK(c, hipMemsetAsync(c->d_xpub, 0, ((size_t)2 * F * c->xtiles_s + (size_t)F * c->xtiles_p) * QR_X_PUBW * 8, c->stream));
    c->xepoch = 1;
  }
  const u64 epoch = c->xepoch;
  if (!root_mode) {
    const unsigned fg = (unsigned)std::min<size_t>((c->N + 2047) / 2048, 1024);
    if (c->world > 1)   // feature shards: the reduced go-left bits (the owner's k_mask), by position
      hipLaunchKernelGGL(k_xflag_mask, dim3(fg), dim3(256), 0, c->stream, c->d_tree, c->d_order[0], c->d_order[1],
                         (const uint32_t *)c->d_mask, c->d_xgoleft);
    else
      hipLaunchKernelGGL(k_xflag, dim3(fg), dim3(256), 0, c->stream, c->d_tree, xr, (const u64 *)x0, (const u64 *)x1,
                         c->N, c->d_xgoleft, 0);
    QR_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(k_xpart, dim3(F, c->xtiles_p), dim3(1024), 0, c->stream, c->d_tree, xr, x0, x1, c->N,
                       (const uint8_t *)c->d_xgoleft, pub_part, c->xtiles_p, epoch, 0, c->world > 1 ? 1 : 0);
    QR_CHECK(c, hipGetLastError());
  }
  QR_CHECK(c, hipMemsetAsync(c->d_xtot, 0, (2 + 2 * (size_t)F) * 8, c->stream));
  hipLaunchKernelGGL(k_xtotal, dim3(64, nodes), dim3(256), 0, c->stream, c->d_tree, mode, rootn, xr,
                     (const u64 *)x0, (const u64 *)x1, c->d_lambda, c->d_scalars, c->d_xtot);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_xscan, dim3(F, c->xtiles_s, nodes), dim3(QR_XS_T), 0, c->stream, c->d_tree, mode, rootn, xr,
                     (const u64 *)x0, (const u64 *)x1, c->N, c->d_lambda, c->d_scalars, c->d_xtot, c->d_woff,
                     c->flocal, pub_scan, c->xtiles_s, epoch, (unsigned long long *)(c->d_xtot + 2),
                     (XTileBest *)c->d_xtbest, c->batch_root ? (u64)c->cur_minls : ~0ull);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_xbest, dim3(F, nodes), dim3(64), 0, c->stream, c->d_tree, mode, rootn,
                     (const XTileBest *)c->d_xtbest, c->xtiles_s, c->d_woff, c->flocal, c->d_lf2gf, c->d_wthr,
                     c->d_featrec, c->d_featthr, (long long *)nullptr);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

// ---------------------------------------------------------------------------
// The whole tree with the split search at the pop (k_tree.hip k_xpop / k_xapply): per step
//   k_xscan + k_xbest (the popped node) -> k_xapply -> k_partition (document-order lists, child
//   sums) -> k_xflag -> k_xpart (every feature's segment) -> k_xpop (children, next pop)
// Lists move (pi N F) entries through the scans instead of (1 + pi) N F, and no pass for the
// children's gradient totals: the winner's cumulative sum is the left child's.
// ---------------------------------------------------------------------------
static int exact_step(qr_ctx *c, int root, int final_call) {
  const u64 *xr = (const u64 *)c->d_xroot;
  u64 *x0 = (u64 *)c->d_xlist[0], *x1 = (u64 *)c->d_xlist[1];
  const unsigned F = (unsigned)c->flocal;
  u64 *pub_scan = (u64 *)c->d_xpub, *pub_part = (u64 *)c->d_xpub + (size_t)2 * F * c->xtiles_s * QR_X_PUBW;
  if (++c->xepoch > 0xFFFFu) {
    QR_CHECK(c, hipMemsetAsync(c->d_xpub, 0, ((size_t)2 * F * c->xtiles_s + (size_t)F * c->xtiles_p) * QR_X_PUBW * 8, c->stream));
    c->xepoch = 1;
  }
  const u64 epoch = c->xepoch;
  hipLaunchKernelGGL(k_xscan, dim3(F, c->xtiles_s, 1), dim3(QR_XS_T), 0, c->stream, c->d_tree, 2, (uint32_t)c->N, xr,
                     (const u64 *)x0, (const u64 *)x1, c->N, c->d_lambda, c->d_scalars, (const long long *)c->d_xnode_tot,
                     c->d_woff, c->flocal, pub_scan, c->xtiles_s, epoch, (unsigned long long *)c->d_xgbest,
                     (XTileBest *)c->d_xtbest, (u64)c->cur_minls);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_xbest, dim3(F, 1), dim3(64), 0, c->stream, c->d_tree, 2, (uint32_t)c->N,
                     (const XTileBest *)c->d_xtbest, c->xtiles_s, c->d_woff, c->flocal, c->d_lf2gf, c->d_wthr,
                     c->d_featrec, c->d_featthr, c->d_xcs);
  QR_CHECK(c, hipGetLastError());
  int rc = qr_k_xapply(c, root);
  if (rc) return rc;
  if ((rc = qr_k_xpartition(c))) return rc;
  const unsigned fg = (unsigned)std::min<size_t>((c->N + 2047) / 2048, 1024);
  hipLaunchKernelGGL(k_xflag, dim3(fg), dim3(256), 0, c->stream, c->d_tree, xr, (const u64 *)x0, (const u64 *)x1,
                     c->N, c->d_xgoleft, 1);
  QR_CHECK(c, hipGetLastError());
  hipLaunchKernelGGL(k_xpart, dim3(F, c->xtiles_p), dim3(1024), 0, c->stream, c->d_tree, xr, x0, x1, c->N,
                     (const uint8_t *)c->d_xgoleft, pub_part, c->xtiles_p, epoch, 1);
  QR_CHECK(c, hipGetLastError());
  return qr_k_xpop(c, 0, c->cur_nleaves, c->cur_minls, final_call);
}

// --subsample (mart.cc:287-329): the tree grows on this iteration's sample.  Every feature's root
// list is cut down to the sample's documents -- a stable partition by the presence bytes the
// sample was marked with (k_xpart: "go left" = in the sample), into list buffer 0, where the
// root then is the segment [0, sub_n) -- one pass over the lists per tree, and everything below
// the root is what it is without a sample, on sub_n documents.
__global__ void k_xsample_desc(QrTreeState *ts, const uint32_t N, const uint32_t k) {
  QrSplitDesc &d = ts->desc;
  d.active = 1;
  d.owner_local = 0;
  d.begin = 0;
  d.end = N;
  d.src_buf = 2;
  d.dst_buf = 0;
  d.lcount = k;
  ts->xs_last = 0;
}

int qr_k_exact_fit(qr_ctx *c, size_t nleaves, uint64_t minls) {
  c->finish_in_decide = false;
  c->tree_counter += 0x9E3779B97F4A7C15ull;  // a fresh feature-subset stream per tree
  c->cur_minls = minls;
  const int frc = qr_k_prep_flush(c);  // (the iteration's scalars: the scans read the scale)
  if (frc) return frc;
  const uint32_t rootn = (uint32_t)(c->sub_k ? c->sub_n : c->N);
  const u64 *root_list = (const u64 *)(c->sub_k ? c->d_xlist[0] : c->d_xroot);
  if (c->sub_k) {
    if (++c->xepoch > 0xFFFFu) {
      QR_CHECK(c, hipMemsetAsync(c->d_xpub, 0,
                                 ((size_t)2 * c->flocal * c->xtiles_s + (size_t)c->flocal * c->xtiles_p) * QR_X_PUBW * 8,
                                 c->stream));
      c->xepoch = 1;
    }
    hipLaunchKernelGGL(k_xsample_desc, dim3(1), dim3(1), 0, c->stream, c->d_tree, (uint32_t)c->N, rootn);
    QR_CHECK(c, hipGetLastError());
    u64 *pub_part = (u64 *)c->d_xpub + (size_t)2 * c->flocal * c->xtiles_s * QR_X_PUBW;
    hipLaunchKernelGGL(k_xpart, dim3((unsigned)c->flocal, c->xtiles_p), dim3(1024), 0, c->stream, c->d_tree,
                       (const u64 *)c->d_xroot, (u64 *)c->d_xlist[0], (u64 *)c->d_xlist[1], c->N,
                       (const uint8_t *)c->d_present, pub_part, c->xtiles_p, (u64)c->xepoch, 0);
    QR_CHECK(c, hipGetLastError());
  }
  QR_CHECK(c, hipMemsetAsync(c->d_xnode_tot, 0, 8, c->stream));
  const unsigned tg = (unsigned)std::min<size_t>(((size_t)rootn + 2047) / 2048, 1024);
  hipLaunchKernelGGL(k_xtotal, dim3(tg ? tg : 1, 1), dim3(256), 0, c->stream, c->d_tree, 0, rootn, root_list,
                     (const u64 *)c->d_xlist[0], (const u64 *)c->d_xlist[1], c->d_lambda, c->d_scalars,
                     c->d_xnode_tot);
  QR_CHECK(c, hipGetLastError());
  int rc = qr_k_xpop(c, 1, nleaves, minls, 0);
  if (rc) return rc;
  const size_t steps = nleaves - 1;
  for (size_t s = 0; s < steps; ++s)
    if ((rc = exact_step(c, s == 0, s + 1 == steps))) return rc;
  c->tree_step = (int)steps;
  return QR_OK;
}

int qr_k_exact_continue(qr_ctx *c, size_t steps) {
  int rc;
  for (size_t s = 0; s < steps; ++s)
    if ((rc = exact_step(c, 0, s + 1 == steps))) return rc;
  return QR_OK;
}
