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
    double *__restrict__ out) {
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
        const double v = tval[t * max_nodes + n] * weights[t0 + t];
        sum = sum + v;
      }
    }
  }
  if (threadIdx.x < nd) out[d0 + threadIdx.x] = sum;
}

int qr_k_ensemble_score(qr_ctx *c, const float *d_x, size_t N, size_t F,
                        double *d_out) {
  if (!c->d_ens) QR_FAIL(c, QR_ERR_STATE, "no ensemble uploaded");
  const size_t fs = F | 1;
  const size_t rows_bytes = ((size_t)SC_DOCS * fs * 4 + 15) & ~(size_t)15;
  const size_t budget = 150 * 1024;
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
                     (uint32_t)tbatch, d_out);
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
  const size_t budget = 150 * 1024;
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
