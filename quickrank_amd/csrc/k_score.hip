// k_score.hip -- tree-ensemble scoring (gfx950).
//
// Stands behind LTR_Algorithm::score_dataset (ltr_algorithm.cc:44-52) ->
// Ensemble::score_instance (ensemble.cc:111-118) -> RTNode::score_instance
// (rtnode.h:134-152):  score(x) = sum_t tree_t(x) * weight_t  in f64, tree order,
// f32 `x[f] <= threshold` at every internal node.
//
// Doc-parallel: a workgroup stages DOCS feature rows in LDS once and streams the
// model through LDS in batches of trees; one lane walks one doc.  The per-doc
// sum runs over the trees in ensemble order with a separate multiply and add
// (no FMA contraction), so scores are bit-identical to the reference's.
#include <algorithm>
#include <cstdlib>

#include "qr_internal.h"

struct DevNode {   // 16 B
  float thr;
  int32_t feat;    // -1 = leaf
  int32_t left, right;
};

#define SC_DOCS 64   // docs per workgroup (one wave)

__global__ __launch_bounds__(SC_DOCS) void k_ensemble_score(
    const float *__restrict__ x, const uint32_t N, const uint32_t F,
    const qr_node_t *__restrict__ nodes, const double *__restrict__ weights,
    const uint32_t ntrees, const uint32_t max_nodes, const uint32_t tbatch,
    double *__restrict__ out, double *__restrict__ partial, const int ignore_weights) {
  // partial != null: Ensemble::partial_scores_instance (ensemble.cc:120-131) -- the
  // per-tree outputs [N][ntrees], times the tree weight unless ignore_weights
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *rows = reinterpret_cast<float *>(smem);                 // [SC_DOCS][F+1]
  const uint32_t fs = F | 1;                                     // odd stride
  const size_t rows_bytes = ((size_t)SC_DOCS * fs * 4 + 15) & ~(size_t)15;
  double *tval = reinterpret_cast<double *>(smem + rows_bytes);  // [tbatch][max_nodes]
  DevNode *tn = reinterpret_cast<DevNode *>(tval + (size_t)tbatch * max_nodes);
  const uint32_t d0 = blockIdx.x * SC_DOCS;
  const uint32_t nd = d0 + SC_DOCS <= N ? SC_DOCS : N - d0;
  // coalesced row staging
  for (uint32_t i = threadIdx.x; i < nd * F; i += SC_DOCS) {
    const uint32_t r = i / F, f = i - r * F;
    rows[r * fs + f] = x[(size_t)d0 * F + i];
  }
  const float *my = rows + threadIdx.x * fs;
  double sum = 0.0;
  for (uint32_t t0 = 0; t0 < ntrees; t0 += tbatch) {
    const uint32_t tb = t0 + tbatch <= ntrees ? tbatch : ntrees - t0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < tb * max_nodes; i += SC_DOCS) {
      const qr_node_t nn = nodes[(size_t)t0 * max_nodes + i];
      DevNode dn;
      dn.thr = nn.threshold;
      dn.feat = nn.feature;
      dn.left = nn.left;
      dn.right = nn.right;
      tn[i] = dn;
      tval[i] = nn.value;
    }
    __syncthreads();
    if (threadIdx.x < nd) {
      for (uint32_t t = 0; t < tb; ++t) {
        const DevNode *tr = tn + t * max_nodes;
        int n = 0;
        DevNode cur = tr[0];
        while (cur.feat >= 0) {
          n = my[cur.feat] <= cur.thr ? cur.left : cur.right;
          cur = tr[n];
        }
        if (partial) {
          double pv = tval[t * max_nodes + n];
          if (!ignore_weights) pv *= weights[t0 + t];
          partial[(size_t)(d0 + threadIdx.x) * ntrees + t0 + t] = pv;
          continue;
        }
        const double v = tval[t * max_nodes + n] * weights[t0 + t];
        sum = sum + v;
      }
    }
  }
  if (threadIdx.x < nd && out) out[d0 + threadIdx.x] = sum;
}

int qr_k_ensemble_score(qr_ctx *c, const float *d_x, size_t N, size_t F,
                        double *d_out, double *d_partial, int ignore_weights) {
  if (!c->d_ens) QR_FAIL(c, QR_ERR_STATE, "no ensemble uploaded");
  const size_t fs = F | 1;
  const size_t rows_bytes = ((size_t)SC_DOCS * fs * 4 + 15) & ~(size_t)15;
  const size_t budget = c->lds_block - 10 * 1024;  // (150 KB of gfx950's 160: room for the static arrays)
  if (rows_bytes + c->ens_maxnodes * 24 > budget)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "feature rows / tree too large for LDS staging");
  size_t tbatch = (budget - rows_bytes) / (c->ens_maxnodes * 24);
  if (tbatch > c->ens_trees) tbatch = c->ens_trees;
  if (tbatch > 64) tbatch = 64;
  if (tbatch == 0) tbatch = 1;
  const size_t lds = rows_bytes + tbatch * c->ens_maxnodes * 24;
  QR_CHECK(c, hipFuncSetAttribute((const void *)k_ensemble_score,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  const unsigned grid = (unsigned)((N + SC_DOCS - 1) / SC_DOCS);
  hipLaunchKernelGGL(k_ensemble_score, dim3(grid), dim3(SC_DOCS), lds, c->stream,
                     d_x, (uint32_t)N, (uint32_t)F, c->d_ens, c->d_ens_w,
                     (uint32_t)c->ens_trees, (uint32_t)c->ens_maxnodes,
                     (uint32_t)tbatch, d_out, d_partial, ignore_weights);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}


// ---------------------------------------------------------------------------
// Oblivious ensembles, the scorer quicklearn --generator oblivious emits
// (generate_oblivious.cc:237-324): per tree m (feature, threshold) pairs, root
// first;  leafidx |= (v[fid[l]] > thr[l]) << (m-1-l);
// score += tree_weight (f32!, generate_oblivious.cc:166) * leaf_outputs[tree][leafidx].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(SC_DOCS) void k_obl_score(
    const float *__restrict__ x, const uint32_t N, const uint32_t F,
    const uint32_t *__restrict__ feat, const float *__restrict__ thr,
    const double *__restrict__ leaves, const float *__restrict__ weights,
    const uint32_t *__restrict__ depths, const uint32_t ntrees, const uint32_t D,
    const uint32_t tbatch, double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *rows = reinterpret_cast<float *>(smem);
  const uint32_t fs = F | 1;
  const size_t rows_bytes = ((size_t)SC_DOCS * fs * 4 + 15) & ~(size_t)15;
  const uint32_t nl = 1u << D;
  double *lv = reinterpret_cast<double *>(smem + rows_bytes);  // [tbatch][nl]
  uint32_t *lf = reinterpret_cast<uint32_t *>(lv + (size_t)tbatch * nl);  // [tbatch][D]
  float *lt = reinterpret_cast<float *>(lf + (size_t)tbatch * D);         // [tbatch][D]
  const uint32_t d0 = blockIdx.x * SC_DOCS;
  const uint32_t nd = d0 + SC_DOCS <= N ? SC_DOCS : N - d0;
  for (uint32_t i = threadIdx.x; i < nd * F; i += SC_DOCS) {
    const uint32_t r = i / F, f = i - r * F;
    rows[r * fs + f] = x[(size_t)d0 * F + i];
  }
  const float *my = rows + threadIdx.x * fs;
  double score = 0.0;
  for (uint32_t t0 = 0; t0 < ntrees; t0 += tbatch) {
    const uint32_t tb = t0 + tbatch <= ntrees ? tbatch : ntrees - t0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < tb * nl; i += SC_DOCS) lv[i] = leaves[(size_t)t0 * nl + i];
    for (uint32_t i = threadIdx.x; i < tb * D; i += SC_DOCS) {
      lf[i] = feat[(size_t)t0 * D + i];
      lt[i] = thr[(size_t)t0 * D + i];
    }
    __syncthreads();
    if (threadIdx.x < nd) {
      for (uint32_t t = 0; t < tb; ++t) {
        const uint32_t m = depths ? depths[t0 + t] : D;
        uint32_t leafidx = 0;
        for (uint32_t l = 0; l < m; ++l)
          leafidx |= (uint32_t)(my[lf[t * D + l]] > lt[t * D + l]) << (m - 1 - l);
        const double v = (double)weights[t0 + t] * lv[t * nl + leafidx];
        score = score + v;
      }
    }
  }
  if (threadIdx.x < nd) out[d0 + threadIdx.x] = score;
}

int qr_k_obl_score(qr_ctx *c, const float *d_x, size_t N, size_t F, double *d_out) {
  if (!c->d_obl_feat) QR_FAIL(c, QR_ERR_STATE, "no oblivious ensemble uploaded");
  const size_t fs = F | 1;
  const size_t rows_bytes = ((size_t)SC_DOCS * fs * 4 + 15) & ~(size_t)15;
  const size_t per_tree = ((size_t)8 << c->obl_depth) + c->obl_depth * 8;
  const size_t budget = c->lds_block - 10 * 1024;  // (150 KB of gfx950's 160: room for the static arrays)
  if (rows_bytes + per_tree > budget)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "feature rows / tree too large for LDS staging");
  size_t tbatch = (budget - rows_bytes) / per_tree;
  if (tbatch > c->obl_trees) tbatch = c->obl_trees;
  if (tbatch > 256) tbatch = 256;
  const size_t lds = rows_bytes + tbatch * per_tree;
  QR_CHECK(c, hipFuncSetAttribute((const void *)k_obl_score,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned grid = (unsigned)((N + SC_DOCS - 1) / SC_DOCS);
  hipLaunchKernelGGL(k_obl_score, dim3(grid), dim3(SC_DOCS), lds, c->stream, d_x, (uint32_t)N,
                     (uint32_t)F, c->d_obl_feat, c->d_obl_thr, c->d_obl_leaves, c->d_obl_w,
                     c->d_obl_depths, (uint32_t)c->obl_trees, (uint32_t)c->obl_depth,
                     (uint32_t)tbatch, d_out);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}


// ===========================================================================
// Binned doc-parallel scoring (the fast path of qr_ensemble_score)
// ===========================================================================
// The walk only ever evaluates `x[f] <= thr`; with the model's thresholds of
// feature f sorted (t_0 < t_1 < ...), b(x) = #{t_i < x} gives
//     x <= t_k  <=>  b(x) <= k                      (NaN -> b = count: always right)
// so a document is reduced ONCE to one small integer per feature (u8 when every
// feature has <= 255 distinct thresholds in the model -- any model trained with
// --num-thresholds <= 255 -- else u16) and the trees compare integers.  This
// quarters (halves) the LDS footprint of a document block, which is what limits
// occupancy here, and keeps the result bit-identical to the f32 walk.
//
// k_doc_bins : f32 rows -> [64-doc block][feature][lane] bins (coalesced both ways)
// k_score_bin: one wave = 64 documents whose bins sit in LDS as [feature][lane];
//              the model streams through LDS in batches of trees shared by the
//              workgroup's waves; each lane walks 4 trees at a time (independent
//              chains hide the LDS latency) and adds leaf * weight strictly in
//              tree order (ensemble.cc:111-118), separate multiply and add.
struct CNode {  // 8 B: internal node of a compact tree
  uint16_t feat, kbin, left, right;  // child: internal index, or 0x8000 | leaf index
};

// One workgroup = one group of DB_FG features x DB_BLOCKS 64-document blocks.  The
// group's threshold rows are staged in LDS once (a binary search out of L2 costs a
// 64-byte sector per step and made the first version L2-bandwidth-bound).
#define DB_FG 32
#define DB_BLOCKS 16
template <typename BT>
__global__ __launch_bounds__(256) void k_doc_bins(const float *__restrict__ x, const uint32_t N,
                                                  const uint32_t F, const uint32_t xstride,
                                                  const float *__restrict__ thr,
                                                  const uint32_t *__restrict__ thr_cnt,
                                                  const uint32_t tmax, const uint32_t lds_thr,
                                                  BT *__restrict__ out, const uint32_t nan_low) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [DB_FG][tmax | 1] when lds_thr: an odd row stride spreads the 32 features' rows over the banks
  const uint32_t tstr = tmax | 1u;
  float *lt = reinterpret_cast<float *>(smem);
  // [DB_FG][64], rows 4 bytes apart from a multiple of the bank row: the 32 features a wave
  // writes for one document land in 32 banks (at a stride of 64 bytes they shared two)
  constexpr uint32_t TS = 64 * sizeof(BT) + 4;
  BT *tile = reinterpret_cast<BT *>(smem + (lds_thr ? (size_t)DB_FG * tstr * 4 : 0));
  __shared__ uint32_t cnt[DB_FG];
  const uint32_t f0 = blockIdx.y * DB_FG;
  const uint32_t nf = f0 + DB_FG <= F ? DB_FG : F - f0;
  if (threadIdx.x < DB_FG) cnt[threadIdx.x] = threadIdx.x < nf ? thr_cnt[f0 + threadIdx.x] : 0;
  if (lds_thr)
    for (uint32_t i = threadIdx.x; i < nf * tmax; i += 256) lt[(i / tmax) * tstr + i % tmax] = thr[(size_t)f0 * tmax + i];
  __syncthreads();
  const uint32_t nblk = (N + 63) / 64;
  for (uint32_t bb = 0; bb < DB_BLOCKS; ++bb) {
    const uint32_t blk = blockIdx.x * DB_BLOCKS + bb;
    if (blk >= nblk) break;
    const uint32_t d0 = blk * 64;
    // a thread's eight (document, feature) pairs: the feature is the same for all of them
    // (256 threads = 8 rows x 32 features), the eight loads are requested before the first search
    constexpr uint32_t PP = 64 * DB_FG / 256;
    const uint32_t f = threadIdx.x % DB_FG, rr = threadIdx.x / DB_FG;
    float v[PP];
#pragma unroll
    for (uint32_t k = 0; k < PP; ++k) {
      const uint32_t r = rr + k * (256 / DB_FG);
      v[k] = (d0 + r < N && f < nf) ? x[(size_t)(d0 + r) * xstride + f0 + f] : 0.0f;
    }
    const float *t = lds_thr ? lt + (size_t)f * tstr : thr + (size_t)(f0 + (f < nf ? f : 0)) * tmax;
    BT *trow = reinterpret_cast<BT *>(reinterpret_cast<char *>(tile) + f * TS);
    const uint32_t cf = cnt[f];
    // lower bound (first index with t[idx] >= v) without data-dependent branches: the trip count
    // depends on the feature's threshold count only, so the eight searches advance together and
    // their LDS reads are in flight together
    uint32_t base[PP];
#pragma unroll
    for (uint32_t k = 0; k < PP; ++k) base[k] = 0;
    uint32_t len = cf;
    while (len > 1) {
      const uint32_t half = len >> 1;
#pragma unroll
      for (uint32_t k = 0; k < PP; ++k) base[k] += t[base[k] + half - 1] < v[k] ? half : 0u;
      len -= half;
    }
#pragma unroll
    for (uint32_t k = 0; k < PP; ++k) {
      const uint32_t r = rr + k * (256 / DB_FG);
      uint32_t lo = base[k] + ((len == 1 && t[base[k]] < v[k]) ? 1u : 0u);
      // NaN: `x <= thr` is false for every threshold (tree walk: always right) and so
      // is `x > thr` (the oblivious scorer's test: always left)
      if (v[k] != v[k]) lo = nan_low ? 0u : cf;
      trow[r] = (d0 + r < N && f < nf) ? (BT)lo : (BT)0;
    }
    __syncthreads();
    // out as dwords: row f of the tile is 64 * sizeof(BT) / 4 of them
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + ((size_t)blk * F + f0) * 64);
    constexpr uint32_t RW = 64 * sizeof(BT) / 4;
    for (uint32_t i = threadIdx.x; i < RW * nf; i += 256)
      dst[i] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(tile) + (i / RW) * TS + (i % RW) * 4);
    __syncthreads();
  }
}

// SELF: the node array of a tree holds its leaves too, as nodes that lead to themselves
// ({offset 0, slot 0xffff, left = right = own index}, entries NIint .. NIint + NL), and an
// internal node carries the BYTE OFFSET of its feature's bin row instead of the feature.  A
// step is then the same four operations for every chain, finished or not -- node address,
// bin address, compare, pick a child -- instead of twelve with the leaf test, the index
// clamp and the field unpacking (the walk was VALU-bound: 12.6 wave instructions per node
// visit); a wave's chains are done when the smallest index is a leaf's.  NI = entries per
// tree in `cnodes` (internal + leaves).  Used whenever a feature row's offset fits 16 bits.
template <typename BT, int NW, bool SELF>
__global__ __launch_bounds__(NW * 64) void k_score_bin(
    const BT *__restrict__ bins, const uint32_t N, const uint32_t F,
    const CNode *__restrict__ cnodes, const double *__restrict__ cleaves,
    const uint16_t *__restrict__ root_code, const double *__restrict__ weights,
    const uint32_t ntrees, const uint32_t NI, const uint32_t NL, const uint32_t tbatch,
    const uint32_t NIint, double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t doc_bytes = ((size_t)F * 64 * sizeof(BT) + 15) & ~(size_t)15;
  BT *mybins = reinterpret_cast<BT *>(smem + wave * doc_bytes);
  char *tb = smem + NW * doc_bytes;
  double *lv = reinterpret_cast<double *>(tb);                          // [tbatch][NL]
  CNode *ln = reinterpret_cast<CNode *>(lv + (size_t)tbatch * NL);      // [tbatch][NI]
  double *lw = reinterpret_cast<double *>(ln + (size_t)tbatch * NI);    // [tbatch]
  uint16_t *lr = reinterpret_cast<uint16_t *>(lw + tbatch);             // [tbatch]
  const uint32_t nblocks = (N + 63) / 64;
  const uint32_t blk = blockIdx.x * NW + wave;
  const bool have = blk < nblocks;
  if (have) {
    const BT *src = bins + (size_t)blk * 64 * F;
    // 16-byte copies of this wave's [F][64] block
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(mybins);
    const uint32_t n16 = (uint32_t)(F * 64 * sizeof(BT) / 16);
    for (uint32_t i = lane; i < n16; i += 64) d4[i] = s4[i];
  }
  const uint32_t doc = blk * 64 + lane;
  double sum = 0.0;
  for (uint32_t t0 = 0; t0 < ntrees; t0 += tbatch) {
    const uint32_t nb = t0 + tbatch <= ntrees ? tbatch : ntrees - t0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb * NL; i += NW * 64) lv[i] = cleaves[(size_t)t0 * NL + i];
    {
      const unsigned long long *sn = reinterpret_cast<const unsigned long long *>(cnodes + (size_t)t0 * NI);
      unsigned long long *dn = reinterpret_cast<unsigned long long *>(ln);
      for (uint32_t i = threadIdx.x; i < nb * NI; i += NW * 64) dn[i] = sn[i];
    }
    for (uint32_t i = threadIdx.x; i < nb; i += NW * 64) {
      lw[i] = weights[t0 + i];
      lr[i] = root_code[t0 + i];
    }
    __syncthreads();
    if (!have) continue;
    // branch-free step: a finished chain re-reads node 0 and keeps its leaf code
    const unsigned long long *ln64 = reinterpret_cast<const unsigned long long *>(ln);
    auto step = [&](const uint32_t c, const unsigned long long *nodes) -> uint32_t {
      const bool leaf = (c & 0x8000u) != 0;
      const unsigned long long nd = nodes[leaf ? 0u : c];
      const uint32_t feat = (uint32_t)nd & 0xffffu, kbin = ((uint32_t)nd >> 16);
      const uint32_t l = (uint32_t)(nd >> 32) & 0xffffu, r = (uint32_t)(nd >> 48);
      const uint32_t bv = mybins[feat * 64 + lane];
      const uint32_t nxt = bv <= kbin ? l : r;
      return leaf ? c : nxt;
    };
    uint32_t t = 0;
    if (SELF) {
      const char *mybytes = reinterpret_cast<const char *>(mybins) + lane * sizeof(BT);
      auto step2 = [&](const uint32_t c, const unsigned long long *nodes) -> uint32_t {
        const uint2 nd = *reinterpret_cast<const uint2 *>(nodes + c);
        const uint32_t bv = *reinterpret_cast<const BT *>(mybytes + (nd.x & 0xffffu));
        return bv <= (nd.x >> 16) ? (nd.y & 0xffffu) : (nd.y >> 16);
      };
      for (; t + 8 <= nb; t += 8) {
        uint32_t c[8];
        const unsigned long long *nn[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          c[j] = lr[t + j];
          nn[j] = ln64 + (size_t)(t + j) * NI;
        }
        for (;;) {
          uint32_t mn = c[0];
#pragma unroll
          for (int j = 1; j < 8; ++j) mn = mn < c[j] ? mn : c[j];
          if (!__any(mn < NIint)) break;
#pragma unroll
          for (int j = 0; j < 8; ++j) c[j] = step2(c[j], nn[j]);
        }
        const double *l0 = lv + (size_t)t * NL;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const double v = l0[j * NL + (c[j] - NIint)] * lw[t + j];
          sum = sum + v;
        }
      }
      for (; t < nb; ++t) {
        uint32_t c0 = lr[t];
        const unsigned long long *n0 = ln64 + (size_t)t * NI;
        while (__any(c0 < NIint)) c0 = step2(c0, n0);
        const double v0 = lv[(size_t)t * NL + (c0 - NIint)] * lw[t];
        sum = sum + v0;
      }
      continue;
    }
    for (; t + 8 <= nb; t += 8) {
      uint32_t c[8];
      const unsigned long long *nn[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        c[j] = lr[t + j];
        nn[j] = ln64 + (size_t)(t + j) * NI;
      }
      for (;;) {
        uint32_t all = c[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) all &= c[j];
        if (!__any((all & 0x8000u) == 0)) break;
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = step(c[j], nn[j]);
      }
      const double *l0 = lv + (size_t)t * NL;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double v = l0[j * NL + (c[j] & 0x7fffu)] * lw[t + j];
        sum = sum + v;
      }
    }
    for (; t + 4 <= nb; t += 4) {
      uint32_t c0 = lr[t], c1 = lr[t + 1], c2 = lr[t + 2], c3 = lr[t + 3];
      const unsigned long long *n0 = ln64 + (size_t)t * NI, *n1 = n0 + NI, *n2 = n1 + NI, *n3 = n2 + NI;
      while (__any(((c0 & c1 & c2 & c3) & 0x8000u) == 0)) {
        c0 = step(c0, n0);
        c1 = step(c1, n1);
        c2 = step(c2, n2);
        c3 = step(c3, n3);
      }
      const double *l0 = lv + (size_t)t * NL;
      const double v0 = l0[c0 & 0x7fffu] * lw[t];
      const double v1 = l0[NL + (c1 & 0x7fffu)] * lw[t + 1];
      const double v2 = l0[2 * NL + (c2 & 0x7fffu)] * lw[t + 2];
      const double v3 = l0[3 * NL + (c3 & 0x7fffu)] * lw[t + 3];
      sum = sum + v0;
      sum = sum + v1;
      sum = sum + v2;
      sum = sum + v3;
    }
    for (; t < nb; ++t) {
      uint32_t c0 = lr[t];
      const unsigned long long *n0 = ln64 + (size_t)t * NI;
      while (__any((c0 & 0x8000u) == 0)) c0 = step(c0, n0);
      const double v0 = lv[(size_t)t * NL + (c0 & 0x7fffu)] * lw[t];
      sum = sum + v0;
    }
  }
  if (have && doc < N) out[doc] = sum;
}

// LDS reads at an integer address (address space 3; the host pass of hipcc only parses these).
// A constant added to the address ends up in the instruction's offset field.
__device__ __forceinline__ uint32_t lds_read_u32(const uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(const __attribute__((address_space(3))) uint32_t *)(a);
#else
  return a;
#endif
}
__device__ __forceinline__ uint32_t lds_read_u8(const uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(const __attribute__((address_space(3))) uint8_t *)(a);
#else
  return a;
#endif
}
__device__ __forceinline__ double lds_read_f64(const uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(const __attribute__((address_space(3))) double *)(a);
#else
  return (double)a;
#endif
}
__device__ __forceinline__ uint32_t lds_address(const void *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)(size_t)((const __attribute__((address_space(3))) char *)p);
#else
  return 0u;
#endif
}

// k_score_p4: the same walk on 4-BYTE node records, for models with u8 bins, a bin-row offset
// that fits 16 bits and trees of at most 255 nodes (any leaf-wise tree of up to 128 leaves
// trained with up to 255 thresholds).  Round 3 rebuilt it around what the counters of the
// round-2 version said (config 5: LDS array 35 % busy, VALU 39 %, the SIMDs' issue slots 65 %,
// waves parked on `s_waitcnt` 57 % of their cycles: the walk is bound by the INSTRUCTIONS of a
// node visit, then by the bank conflicts of its two LDS reads):
//  * record = {row offset : 16, 255 - slot : 8, left child : 8}, the nodes of a tree in LEVEL
//    order with siblings adjacent (right = left + 1).  `w = record + (bin << 16)`: the byte sum
//    carries into the child field exactly when bin > slot, so BYTE 3 of w is the next node -- no
//    compare, no select, no VCC (an SDWA compare that writes VCC costs two wait states before
//    its reader).  A leaf is {0, 0, itself}: nothing carries, the walk stays.  There is no leaf
//    test: a group of eight trees runs for as many steps as its deepest tree has levels (one
//    byte per group, from the model); finished chains idle on their leaves.
//  * the tiles of a batch (16 trees x NNP records, NNP = 128 or 256) sit at the START of the
//    workgroup's LDS, so a tile's base is an immediate of the ds_read: a visit is
//    `ds_read_b32 record, [4 * node] offset:tile` -> `row + lane base` -> `ds_read_u8` ->
//    `w = (bin << 16) + record` -> `4 * node = byte 3 of w << 2`: three vector instructions
//    (five in round 2), both groups of a batch unrolled.
//  * both reads are laid out for the banks (ds_read_b32 / ds_read_u8: two groups of 32 lanes,
//    bank = (address / 4) mod 32, MI355X_MICROARCH.md "LDS"): a wave's chains move in lockstep,
//    one level per step, and the nodes of a level are consecutive dwords -- up to 32 nodes of a
//    level never share a bank (depth-first order scatters a level over the tree's dwords: ~3
//    addresses per bank at the deep levels); the bins of a wave's 64 documents sit as
//    [feature / 4][lane][feature % 4], one dword per lane and feature quad, so lane l always
//    reads bank l mod 32 whatever feature its node tests ([feature][lane] bytes put the four
//    lanes of a quad on bank 16 * feature + quad: features of one parity collide).  A node's
//    row offset is (feature / 4) * 256 + feature % 4.
//  * leaf values sit at the leaves' own positions, already multiplied by the tree's weight
//    (the same f64 product `tree(x) * weight` of ensemble.cc:111-118, done once per model, not
//    once per document), and are added strictly in tree order.
// The model is stored batch by batch, as the LDS image of a batch (16 tiles of records, then the
// 16 tiles of leaf values), padded to whole batches; `ntrees` says how many trees are real.
#define P4_TB 16
typedef uint32_t p4_u32x4 __attribute__((ext_vector_type(4)));  // (arrays of it stay in registers)
// one step of chains J0 .. J0 + NJ - 1 of a batch (chain j walks tile j)
template <int NNP, int J0, int NJ>
__device__ __forceinline__ void p4_step(uint32_t (&a)[P4_TB], const uint32_t mybase) {
  constexpr uint32_t TILE = NNP * 4;
#pragma unroll
  for (int j = J0; j < J0 + NJ; ++j) {
    const uint32_t nd = lds_read_u32(a[j] + j * TILE);
    const uint32_t bv = lds_read_u8(mybase + (nd & 0xffffu));
    const uint32_t w = nd + (bv << 16);
    // a = (w >> 24) << 2
    asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
        : "=v"(a[j])
        : "v"(w));
  }
}

// A batch: sixteen chains in flight for the levels both groups of eight trees have, then the
// deeper group alone (a group's depth is its deepest tree's: finished chains idle on their
// leaves); the sixteen leaf values are added in tree order.
template <int NNP>
__device__ __forceinline__ void p4_walk_batch(const uint32_t mybase, const uint32_t d0, const uint32_t d1,
                                              const int cnt, double &sum) {
  constexpr uint32_t TILE = NNP * 4, LV0 = P4_TB * TILE, LVT = NNP * 8;
  uint32_t a[P4_TB];  // 4 * node of each chain
#pragma unroll
  for (int j = 0; j < P4_TB; ++j) a[j] = 0;
  const uint32_t both = d0 < d1 ? d0 : d1;
  for (uint32_t k = 0; k < both; ++k) p4_step<NNP, 0, 16>(a, mybase);
  for (uint32_t k = both; k < d0; ++k) p4_step<NNP, 0, 8>(a, mybase);
  for (uint32_t k = both; k < d1; ++k) p4_step<NNP, 8, 8>(a, mybase);
  if (cnt >= P4_TB) {
#pragma unroll
    for (int j = 0; j < P4_TB; ++j) sum = sum + lds_read_f64(a[j] * 2 + (LV0 + j * LVT));
  } else {
#pragma unroll
    for (int j = 0; j < P4_TB; ++j)
      if (j < cnt) sum = sum + lds_read_f64(a[j] * 2 + (LV0 + j * LVT));
  }
}

template <int NW, int NNP>
__global__ __launch_bounds__(NW * 64) void k_score_p4(
    const uint8_t *__restrict__ bins, const uint32_t N, const uint32_t F,
    const p4_u32x4 *__restrict__ batches, const uint32_t *__restrict__ gdepth32, const uint32_t ntrees,
    double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr uint32_t TILE = NNP * 4, LV0 = P4_TB * TILE, LVT = NNP * 8, DOCS0 = LV0 + P4_TB * LVT;
  constexpr uint32_t B16 = DOCS0 / 16;                          // 16-byte pieces of one batch
  constexpr uint32_t PF = (B16 + NW * 64 - 1) / (NW * 64);      // ... per thread
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t FQ = (F + 3) / 4;  // feature quads
  const uint32_t doc_bytes = FQ * 256;
  // the tiles' addresses are instruction immediates: the dynamic LDS must start at 0 (it does
  // in a kernel without static __shared__ variables)
  {
    uint32_t sb = lds_address(smem);
    asm volatile("" : "+v"(sb));  // (the address of a variable never compares equal to 0 at compile time)
    if (sb != 0) __builtin_trap();
  }
  // The next batch travels from memory into registers while this one is walked: a workgroup has
  // the CU to itself (its documents fill the LDS), so nothing else would hide that latency.
  // (every thread loads PF pieces, the last ones clamped to the batch's end: no predicated
  // register writes, the pieces stay in registers)
  p4_u32x4 pf[PF];
  uint32_t pfi[PF];
#pragma unroll
  for (uint32_t r = 0; r < PF; ++r) {
    const uint32_t i = threadIdx.x + r * NW * 64;
    pfi[r] = i < B16 ? i : B16 - 1;
  }
#pragma unroll
  for (uint32_t r = 0; r < PF; ++r) pf[r] = batches[pfi[r]];
  const uint32_t nblocks = (N + 63) / 64;
  const uint32_t blk = blockIdx.x * NW + wave;
  const bool have = blk < nblocks;
  if (have) {
    // [feature][lane] bytes (k_doc_bins) -> [feature / 4][lane] dwords: once per wave and model
    const uint8_t *src = bins + (size_t)blk * 64 * F + lane;
    uint32_t *d = reinterpret_cast<uint32_t *>(smem + DOCS0 + wave * doc_bytes) + lane;
    const uint32_t full = F / 4;
#pragma unroll 4
    for (uint32_t q = 0; q < full; ++q) {
      const uint8_t *s4 = src + (size_t)q * 256;
      d[q * 64] = (uint32_t)s4[0] | ((uint32_t)s4[64] << 8) | ((uint32_t)s4[128] << 16) | ((uint32_t)s4[192] << 24);
    }
    if (full < FQ) {
      uint32_t w = 0;
      for (uint32_t f = full * 4; f < F; ++f) w |= (uint32_t)src[(size_t)f * 64] << (8 * (f & 3));
      d[full * 64] = w;
    }
  }
  const uint32_t doc = blk * 64 + lane;
  const uint32_t mybase = DOCS0 + wave * doc_bytes + lane * 4;
  p4_u32x4 *lds4 = reinterpret_cast<p4_u32x4 *>(smem);
  double sum = 0.0;
  for (uint32_t t0 = 0; t0 < ntrees; t0 += P4_TB) {
    // two groups' depths (t0 is a multiple of 16: the pair never straddles a dword)
    const uint32_t d01 = (gdepth32[t0 >> 5] >> ((t0 & 16u) ? 16 : 0)) & 0xffffu;
    __syncthreads();  // every wave is done with the previous batch's tiles
#pragma unroll
    for (uint32_t r = 0; r < PF; ++r) lds4[pfi[r]] = pf[r];  // (clamped pieces rewrite the last one)
    __syncthreads();
    if (t0 + P4_TB < ntrees) {
      const p4_u32x4 *src = batches + (size_t)(t0 / P4_TB + 1) * B16;
#pragma unroll
      for (uint32_t r = 0; r < PF; ++r) pf[r] = src[pfi[r]];
    }
    if (!have) continue;
    const int left = (int)(ntrees - t0);
    p4_walk_batch<NNP>(mybase, d01 & 0xffu, d01 >> 8, left, sum);
  }
  if (have && doc < N) out[doc] = sum;
}

template <int NW, int NNP>
static int launch_p4_nw(qr_ctx *c, size_t N, double *d_out) {
  const size_t F = c->sb_F;
  const size_t doc_bytes = ((F + 3) / 4) * 256;
  const size_t lds = NW * doc_bytes + (size_t)P4_TB * NNP * 12;
  const size_t nblk = (N + 63) / 64;
  QR_CHECK(c, hipFuncSetAttribute((const void *)k_score_p4<NW, NNP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL((k_score_p4<NW, NNP>), dim3((unsigned)((nblk + NW - 1) / NW)), dim3(NW * 64), lds, c->stream,
                     (const uint8_t *)c->d_sb_bins, (uint32_t)N, (uint32_t)F, (const p4_u32x4 *)c->d_p4_batches,
                     (const uint32_t *)c->d_p4_depth, (uint32_t)c->ens_trees, d_out);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

template <int NNP>
static int launch_p4(qr_ctx *c, size_t N, double *d_out, size_t nw) {
  switch (nw) {
    case 16: return launch_p4_nw<16, NNP>(c, N, d_out);
    case 12: return launch_p4_nw<12, NNP>(c, N, d_out);
    case 10: return launch_p4_nw<10, NNP>(c, N, d_out);
    case 8: return launch_p4_nw<8, NNP>(c, N, d_out);
    case 6: return launch_p4_nw<6, NNP>(c, N, d_out);
    default: return launch_p4_nw<4, NNP>(c, N, d_out);
  }
}

template <typename BT, int NW>
static int launch_binned_nw(qr_ctx *c, const float *d_x, size_t N, size_t xstride, double *d_out,
                            size_t tbatch) {
  const size_t F = c->sb_F;
  const size_t doc_bytes = (F * 64 * sizeof(BT) + 15) & ~(size_t)15;
  const size_t NN = c->sb_self ? c->sb_NI + c->sb_NL : c->sb_NI;  // node entries per tree
  const size_t per_tree = c->sb_NL * 8 + NN * 8 + 8 + 2;
  const size_t lds = NW * doc_bytes + tbatch * per_tree + 64;
  const size_t nblk = (N + 63) / 64;
  if (c->sb_self) {
    QR_CHECK(c, hipFuncSetAttribute((const void *)k_score_bin<BT, NW, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_score_bin<BT, NW, true>), dim3((unsigned)((nblk + NW - 1) / NW)), dim3(NW * 64), lds,
                       c->stream, (const BT *)c->d_sb_bins, (uint32_t)N, (uint32_t)F,
                       (const CNode *)c->d_sb_nodes, c->d_sb_leaves, c->d_sb_root, c->d_ens_w,
                       (uint32_t)c->ens_trees, (uint32_t)NN, (uint32_t)c->sb_NL, (uint32_t)tbatch,
                       (uint32_t)c->sb_NI, d_out);
  } else {
    QR_CHECK(c, hipFuncSetAttribute((const void *)k_score_bin<BT, NW, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_score_bin<BT, NW, false>), dim3((unsigned)((nblk + NW - 1) / NW)), dim3(NW * 64), lds,
                       c->stream, (const BT *)c->d_sb_bins, (uint32_t)N, (uint32_t)F,
                       (const CNode *)c->d_sb_nodes, c->d_sb_leaves, c->d_sb_root, c->d_ens_w,
                       (uint32_t)c->ens_trees, (uint32_t)NN, (uint32_t)c->sb_NL, (uint32_t)tbatch,
                       (uint32_t)c->sb_NI, d_out);
  }
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}

template <typename BT>
static int launch_binned(qr_ctx *c, const float *d_x, size_t N, size_t xstride, double *d_out) {
  const size_t F = c->sb_F;  // features the model tests; the rows may be wider
  const size_t doc_bytes = (F * 64 * sizeof(BT) + 15) & ~(size_t)15;
  const size_t per_tree = c->sb_NL * 8 + (c->sb_self ? c->sb_NI + c->sb_NL : c->sb_NI) * 8 + 8 + 2;
  const size_t budget = c->lds_block - 1024;
  // tree batch: 16..32 trees; the rest of the LDS goes to document blocks (occupancy)
  size_t tbatch = 32;
  while (tbatch > 8 && 4 * doc_bytes + tbatch * per_tree + 64 > budget) tbatch -= 8;
  if (4 * doc_bytes + tbatch * per_tree + 64 > budget) return -1;  // caller falls back
  size_t nw = (budget - tbatch * per_tree - 64) / doc_bytes;
  nw = nw >= 16 ? 16 : nw >= 12 ? 12 : nw >= 8 ? 8 : 4;
  const size_t nblk = (N + 63) / 64;
  const size_t need = nblk * 64 * F * sizeof(BT);
  if (need > c->sb_bins_bytes) {
    if (c->d_sb_bins) (void)hipFree(c->d_sb_bins);
    c->d_sb_bins = nullptr;
    QR_CHECK(c, hipMalloc(&c->d_sb_bins, need));
    c->sb_bins_bytes = need;
  }
  const uint32_t lds_thr = (size_t)DB_FG * (c->sb_tmax | 1) * 4 <= 96 * 1024 ? 1 : 0;
  const size_t lds_a = (lds_thr ? (size_t)DB_FG * (c->sb_tmax | 1) * 4 : 0) + DB_FG * (64 * sizeof(BT) + 4);
  if (lds_a > 64 * 1024)
    QR_CHECK(c, hipFuncSetAttribute((const void *)k_doc_bins<BT>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
  hipLaunchKernelGGL(k_doc_bins<BT>,
                     dim3((unsigned)((nblk + DB_BLOCKS - 1) / DB_BLOCKS), (unsigned)((F + DB_FG - 1) / DB_FG)),
                     dim3(256), lds_a, c->stream, d_x, (uint32_t)N, (uint32_t)F, (uint32_t)xstride,
                     c->d_sb_thr, c->d_sb_thr_cnt, (uint32_t)c->sb_tmax, lds_thr, (BT *)c->d_sb_bins, 0u);
  QR_CHECK(c, hipGetLastError());
  if (sizeof(BT) == 1 && c->p4_ready) {  // 4-byte node records (k_score_p4)
    // as many 64-document blocks per workgroup as the LDS holds next to one batch of tiles
    // (6 / 8 / 10 waves per CU: 353 / 257 / 249 ms at config 5 in round 2)
    const size_t doc4 = ((F + 3) / 4) * 256;
    const size_t tiles = (size_t)P4_TB * c->p4_NNP * 12;
    if (tiles + 4 * doc4 <= budget) {
      const size_t n = (budget - tiles) / doc4;
      const size_t nw4 = n >= 16 ? 16 : n >= 12 ? 12 : n >= 10 ? 10 : n >= 8 ? 8 : n >= 6 ? 6 : 4;
      return c->p4_NNP == 128 ? launch_p4<128>(c, N, d_out, nw4) : launch_p4<256>(c, N, d_out, nw4);
    }
  }
  switch (nw) {
    case 16: return launch_binned_nw<BT, 16>(c, d_x, N, xstride, d_out, tbatch);
    case 12: return launch_binned_nw<BT, 12>(c, d_x, N, xstride, d_out, tbatch);
    case 8: return launch_binned_nw<BT, 8>(c, d_x, N, xstride, d_out, tbatch);
    default: return launch_binned_nw<BT, 4>(c, d_x, N, xstride, d_out, tbatch);
  }
}

// ===========================================================================
// Oblivious ensembles on binned documents (generate_oblivious.cc:237-324): every
// level of a tree tests ONE (feature, threshold) for all documents, so with the
// documents' bins in LDS as [feature][lane] a level costs one conflict-free byte
// read, a compare and a shift-or; the (feature, threshold index) pairs are uniform
// over the wave.  `x > thr`  <=>  bin(x) > index(thr)  (NaN -> bin 0 -> false, as
// the f32 comparison).  Four trees are in flight per lane; leaf * weight (f32
// weight promoted, :312-324) is added strictly in tree order.
// ===========================================================================
// DC: the array depth when it is a compile-time constant (unrolled level loop), 0 = any
template <typename BT, int NW, int DC>
__global__ __launch_bounds__(NW * 64) void k_obl_score_bin(
    const BT *__restrict__ bins, const uint32_t N, const uint32_t F,
    const uint32_t *__restrict__ fk, const double *__restrict__ leaves,
    const float *__restrict__ weights, const uint32_t *__restrict__ depths,
    const uint32_t ntrees, const uint32_t Drt, const uint32_t tbatch, double *__restrict__ out) {
  const uint32_t D = DC ? (uint32_t)DC : Drt;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t doc_bytes = ((size_t)F * 64 * sizeof(BT) + 15) & ~(size_t)15;
  BT *mybins = reinterpret_cast<BT *>(smem + wave * doc_bytes);
  const uint32_t nl = 1u << D;
  double *lv = reinterpret_cast<double *>(smem + NW * doc_bytes);  // [tbatch][nl]
  uint32_t *lfk = reinterpret_cast<uint32_t *>(lv + (size_t)tbatch * nl);  // [tbatch][D]
  float *lw = reinterpret_cast<float *>(lfk + (size_t)tbatch * D);         // [tbatch]
  uint32_t *lm = reinterpret_cast<uint32_t *>(lw + tbatch);                // [tbatch] actual depth
  const uint32_t nblk = (N + 63) / 64;
  const uint32_t blk = blockIdx.x * NW + wave;
  const bool live = blk < nblk;
  if (live) {
    const BT *src = bins + (size_t)blk * F * 64;
    for (uint32_t i = lane; i < F * 64; i += 64) mybins[i] = src[i];
  }
  double score = 0.0;
  for (uint32_t t0 = 0; t0 < ntrees; t0 += tbatch) {
    const uint32_t tb = t0 + tbatch <= ntrees ? tbatch : ntrees - t0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < tb * nl; i += NW * 64) lv[i] = leaves[(size_t)t0 * nl + i];
    for (uint32_t i = threadIdx.x; i < tb * D; i += NW * 64) lfk[i] = fk[(size_t)t0 * D + i];
    for (uint32_t i = threadIdx.x; i < tb; i += NW * 64) {
      lw[i] = weights[t0 + i];
      lm[i] = depths ? depths[t0 + i] : D;
    }
    __syncthreads();
    if (!live) continue;
    uint32_t t = 0;
    for (; t + 4 <= tb; t += 4) {
      uint32_t idx[4] = {0, 0, 0, 0};
      uint32_t m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) m[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lm[t + u]);
      auto level = [&](const uint32_t l) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (l < m[u]) {  // uniform over the wave
            // (feature, index) is the same for every lane: keep it in scalar registers
            const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)lfk[(t + u) * D + l]);
            const uint32_t b = mybins[(v & 0xffffu) * 64 + lane];
            idx[u] |= (uint32_t)(b > (v >> 16)) << (m[u] - 1 - l);
          }
        }
      };
      if constexpr (DC != 0) {
#pragma unroll
        for (uint32_t l = 0; l < (uint32_t)DC; ++l) level(l);
      } else {
        for (uint32_t l = 0; l < D; ++l) level(l);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double v = (double)lw[t + u] * lv[(size_t)(t + u) * nl + idx[u]];
        score = score + v;
      }
    }
    for (; t < tb; ++t) {
      const uint32_t m = lm[t];
      uint32_t idx = 0;
      for (uint32_t l = 0; l < m; ++l) {
        const uint32_t v = lfk[t * D + l];
        const uint32_t b = mybins[(v & 0xffffu) * 64 + lane];
        idx |= (uint32_t)(b > (v >> 16)) << (m - 1 - l);
      }
      const double v = (double)lw[t] * lv[(size_t)t * nl + idx];
      score = score + v;
    }
  }
  if (live && blk * 64 + lane < N) out[(size_t)blk * 64 + lane] = score;
}

// k_obl_score_s: the same scorer with a tree's level tests in SCALAR registers (u8 bins, depth
// <= 8).  k_obl_score_bin fetches every (feature, slot) pair from LDS and broadcasts it with
// v_readfirstlane -- nine instructions and two waits per level test, 15 CU cycles per wave and
// test; here a tree is sixteen dwords {row offset[8], slot[8]} that one s_load_dwordx16 brings
// in, and a level test is `row + lane base` -> ds_read_u8 -> compare with the slot ->
// `index = 2 * index + carry`: three vector instructions and one conflict-free LDS read (the
// bins of a wave's 64 documents are [feature][lane] bytes: a level's feature is the same for
// every lane).  Four trees are in flight per lane.  Leaf values are stored times the tree's
// (f32, promoted) weight -- the product of generate_oblivious.cc:312-324, once per model -- and
// are added strictly in tree order; shallower trees are padded at the END with levels that
// are never true, their leaves stored at the shifted indices; the model is padded to whole
// batches with all-zero trees (score + 0.0 == score: the sum is never -0.0).
struct ObsTree {
  uint32_t row[8], slot[8];
};
template <int NW, int D>
__global__ __launch_bounds__(NW * 64) void k_obl_score_s(
    const uint8_t *__restrict__ bins, const uint32_t N, const uint32_t F,
    const ObsTree *__restrict__ trees, const p4_u32x4 *__restrict__ leaves, const uint32_t tpad,
    const uint32_t tb, double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr uint32_t NL8 = (1u << D) * 8;  // bytes of a tree's leaf values
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tile = tb * NL8;                   // leaf values of a batch, at the start of the LDS
  const uint32_t doc_bytes = (F * 64 + 15) & ~15u;
  {
    uint32_t sb = lds_address(smem);
    asm volatile("" : "+v"(sb));
    if (sb != 0) __builtin_trap();  // (LDS addresses below are absolute)
  }
  const uint32_t nblk = (N + 63) / 64;
  const uint32_t blk = blockIdx.x * NW + wave;
  const bool live = blk < nblk;
  // the next batch's leaf values travel into registers while this one is walked
  constexpr uint32_t PF = 4;  // 16-byte pieces per thread (the host sizes the batch for it)
  const uint32_t b16 = tile / 16;
  if (b16 > PF * NW * 64) __builtin_trap();
  p4_u32x4 pf[PF];
  uint32_t pfi[PF];
#pragma unroll
  for (uint32_t r = 0; r < PF; ++r) {
    const uint32_t i = threadIdx.x + r * NW * 64;
    pfi[r] = i < b16 ? i : b16 - 1;
  }
#pragma unroll
  for (uint32_t r = 0; r < PF; ++r) pf[r] = leaves[pfi[r]];
  if (live) {
    const p4_u32x4 *src = reinterpret_cast<const p4_u32x4 *>(bins + (size_t)blk * F * 64);
    p4_u32x4 *dst = reinterpret_cast<p4_u32x4 *>(smem + tile + wave * doc_bytes);
    for (uint32_t i = lane; i < F * 4; i += 64) dst[i] = src[i];
  }
  const uint32_t mybase = tile + wave * doc_bytes + lane;
  p4_u32x4 *lds4 = reinterpret_cast<p4_u32x4 *>(smem);
  double score = 0.0;
  for (uint32_t t0 = 0; t0 < tpad; t0 += tb) {
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < PF; ++r) lds4[pfi[r]] = pf[r];
    __syncthreads();
    if (t0 + tb < tpad) {
      const p4_u32x4 *src = leaves + (size_t)(t0 + tb) * (NL8 / 16);
#pragma unroll
      for (uint32_t r = 0; r < PF; ++r) pf[r] = src[pfi[r]];
    }
    if (!live) continue;
    for (uint32_t t = 0; t < tb; t += 4) {
      ObsTree tr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) tr[u] = trees[t0 + t + u];  // uniform: scalar loads
      uint32_t idx[4] = {0, 0, 0, 0};
#pragma unroll
      for (int l = 0; l < D; ++l) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t b = lds_read_u8(mybase + tr[u].row[l]);
          // index = 2 * index + (bin > slot): compare into the carry, add with carry
          asm("v_cmp_lt_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
              : "+v"(idx[u])
              : "v"(b), "s"(tr[u].slot[l])
              : "vcc");
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) score = score + lds_read_f64(idx[u] * 8 + (t + u) * NL8);
    }
  }
  if (live && blk * 64 + lane < N) out[(size_t)blk * 64 + lane] = score;
}

template <int NW>
static int launch_obl_s(qr_ctx *c, size_t N, double *d_out) {
  const size_t F = c->ob_F;
  const size_t doc_bytes = (F * 64 + 15) & ~(size_t)15;
  const size_t nl8 = ((size_t)1 << c->obl_depth) * 8;
  const size_t lds = c->obs_tb * nl8 + NW * doc_bytes;
  const size_t nblk = (N + 63) / 64;
  auto launch = [&](auto kernel) -> int {
    QR_CHECK(c, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)((nblk + NW - 1) / NW)), dim3(NW * 64), lds, c->stream,
                       (const uint8_t *)c->d_sb_bins, (uint32_t)N, (uint32_t)F,
                       (const ObsTree *)c->d_obs_trees, (const p4_u32x4 *)c->d_obs_leaves,
                       (uint32_t)c->obs_tpad, (uint32_t)c->obs_tb, d_out);
    QR_CHECK(c, hipGetLastError());
    return QR_OK;
  };
  switch (c->obl_depth) {
    case 1: return launch(k_obl_score_s<NW, 1>);
    case 2: return launch(k_obl_score_s<NW, 2>);
    case 3: return launch(k_obl_score_s<NW, 3>);
    case 4: return launch(k_obl_score_s<NW, 4>);
    case 5: return launch(k_obl_score_s<NW, 5>);
    case 6: return launch(k_obl_score_s<NW, 6>);
    case 7: return launch(k_obl_score_s<NW, 7>);
    default: return launch(k_obl_score_s<NW, 8>);
  }
}

template <typename BT>
static int launch_obl_binned(qr_ctx *c, const float *d_x, size_t N, size_t xstride, double *d_out) {
  constexpr int NW = 8;
  const size_t F = c->ob_F;
  const size_t doc_bytes = (F * 64 * sizeof(BT) + 15) & ~(size_t)15;
  const size_t nl = (size_t)1 << c->obl_depth;
  const size_t per_tree = nl * 8 + c->obl_depth * 4 + 8;
  const size_t budget = c->lds_block - 10 * 1024;  // (150 KB of gfx950's 160: room for the static arrays)
  if (NW * doc_bytes + 4 * per_tree + 64 > budget) return -1;  // caller falls back
  // two workgroups per CU when the document tiles allow it (occupancy hides the
  // dependent LDS reads of the level loop); the rest of the LDS goes to the tree batch
  const size_t half = c->lds_cu / 2 - 2 * 1024;
  const size_t room = NW * doc_bytes + 8 * per_tree + 64 <= half ? half : budget;
  size_t tbatch = (room - NW * doc_bytes - 64) / per_tree;
  tbatch = std::min<size_t>(tbatch, 128) & ~(size_t)3;
  if (tbatch > c->obl_trees) tbatch = (c->obl_trees + 3) & ~(size_t)3;
  const size_t nblk = (N + 63) / 64;
  const size_t need = nblk * 64 * F * sizeof(BT);
  if (need > c->sb_bins_bytes) {
    if (c->d_sb_bins) (void)hipFree(c->d_sb_bins);
    c->d_sb_bins = nullptr;
    QR_CHECK(c, hipMalloc(&c->d_sb_bins, need));
    c->sb_bins_bytes = need;
  }
  const uint32_t lds_thr = (size_t)DB_FG * (c->ob_tmax | 1) * 4 <= 96 * 1024 ? 1 : 0;
  const size_t lds_a = (lds_thr ? (size_t)DB_FG * (c->ob_tmax | 1) * 4 : 0) + DB_FG * (64 * sizeof(BT) + 4);
  if (lds_a > 64 * 1024)
    QR_CHECK(c, hipFuncSetAttribute((const void *)k_doc_bins<BT>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
  hipLaunchKernelGGL(k_doc_bins<BT>,
                     dim3((unsigned)((nblk + DB_BLOCKS - 1) / DB_BLOCKS), (unsigned)((F + DB_FG - 1) / DB_FG)),
                     dim3(256), lds_a, c->stream, d_x, (uint32_t)N, (uint32_t)F, (uint32_t)xstride,
                     c->d_ob_thr, c->d_ob_thr_cnt, (uint32_t)c->ob_tmax, lds_thr, (BT *)c->d_sb_bins, 1u);
  QR_CHECK(c, hipGetLastError());
  if (sizeof(BT) == 1 && c->obs_ready) {
    // the workgroups' documents next to one batch of leaf values; two workgroups per CU
    // (the batch was sized at upload for obs_nw 64-document blocks per workgroup)
    return c->obs_nw == 8 ? launch_obl_s<8>(c, N, d_out) : launch_obl_s<4>(c, N, d_out);
  }
  const size_t lds = NW * doc_bytes + tbatch * per_tree + 64;
  auto launch = [&](auto kernel) -> int {
    QR_CHECK(c, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)((nblk + NW - 1) / NW)), dim3(NW * 64), lds, c->stream,
                       (const BT *)c->d_sb_bins, (uint32_t)N, (uint32_t)F, c->d_ob_fk, c->d_obl_leaves,
                       c->d_obl_w, c->d_obl_depths, (uint32_t)c->obl_trees, (uint32_t)c->obl_depth,
                       (uint32_t)tbatch, d_out);
    QR_CHECK(c, hipGetLastError());
    return QR_OK;
  };
  switch (c->obl_depth) {  // the depths quicklearn's --tree-depth usually takes
    case 3: return launch(k_obl_score_bin<BT, NW, 3>);
    case 4: return launch(k_obl_score_bin<BT, NW, 4>);
    case 5: return launch(k_obl_score_bin<BT, NW, 5>);
    case 6: return launch(k_obl_score_bin<BT, NW, 6>);
    case 8: return launch(k_obl_score_bin<BT, NW, 8>);
    default: return launch(k_obl_score_bin<BT, NW, 0>);
  }
}

int qr_k_obl_score_fast(qr_ctx *c, const float *d_x, size_t N, size_t F, double *d_out) {
  if (!c->ob_ready || F < c->ob_F) return -1;
  return c->ob_u8 ? launch_obl_binned<uint8_t>(c, d_x, N, F, d_out)
                  : launch_obl_binned<uint16_t>(c, d_x, N, F, d_out);
}

int qr_k_ensemble_score_fast(qr_ctx *c, const float *d_x, size_t N, size_t F, double *d_out) {
  if (!c->sb_ready || F < c->sb_F) return -1;
  return c->sb_u8 ? launch_binned<uint8_t>(c, d_x, N, F, d_out)
                  : launch_binned<uint16_t>(c, d_x, N, F, d_out);
}
