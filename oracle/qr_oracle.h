/*
 * qr_oracle.h -- CPU restatement of QuickRank's LambdaMART/GBRT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product path (quickrank_amd/, include/qr_hip.h) never links or calls it.
 *
 * Every function cites the reference file:line (relative to the upstream
 * hpclab/quickrank checkout) whose behaviour it restates.  Arithmetic types
 * follow include/types.h:28-35 of the reference: Feature/Label = f32,
 * Score/MetricScore = f64, counts = 64-bit unsigned.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - ranking sort, DCG/NDCG value, NDCG jacobian, root/child/sibling
 *     histograms, radix argsort order: PINNED against the reference's own
 *     translation units compiled unmodified into oracle/_ref (no stand-in
 *     headers), see oracle/Makefile and tests/test_oracle_vs_ref.py, and
 *     against the reference's known-answer tests (test-dcg.cc, test-ndcg.cc).
 *   - pair loop of LambdaMart::compute_pseudoresponses, RegressionTree::{fit,
 *     split,update_output}, ObliviousRT::fit, Mart::{init,learn,
 *     update_modelscores}, Ensemble::score_instance: restatement only --
 *     those translation units include pugixml (an un-vendored submodule) and
 *     are unbuildable here without stand-in headers: PARITY UNPINNED for them.
 */
#ifndef QR_ORACLE_H_
#define QR_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ranking sort (libstdc++ 11 std::sort behaviour) -- */
/* queryresults.cc:37-53: idx = 0..n-1 sorted with comp(i,j) = s[i] > s[j]   */
void qro_rank_by_score(const double *scores, size_t n, uint64_t *idx);
/* ndcg.cc:40-41: std::sort(labels, labels+n, std::greater<int>())           */
void qro_sort_labels_desc_int(float *labels, size_t n);
/* std::partial_sort(first,last,last) == the introsort depth-limit fallback   */
void qro_heapsort_by_score(const double *scores, size_t n, uint64_t *idx);
/* maxheap.h:58-88 driven by a push (ops[i] >= 0: value) / pop (ops[i] < 0) sequence;  */
/* symmatrix.h:29-89 packed index of every (i, j).  Test hooks for the pins.           */
void qro_heap_trace(const double *keys, const int32_t *ops, size_t n, size_t initsize,
                    int32_t *top_out, uint64_t *size_out);
void qro_sym_index(size_t size, uint64_t *out);

/* ---- metric (dcg.cc:33-57, ndcg.cc:35-58, metric.h:77-106) --------------- */
double qro_dcg(const float *labels, size_t len, size_t cutoff);
double qro_idcg(const float *labels, size_t n, size_t cutoff);
double qro_dcg_query(const float *labels, const double *scores, size_t n,
                     size_t cutoff);
double qro_ndcg_query(const float *labels, const double *scores, size_t n,
                      size_t cutoff);
/* metric: 0 = DCG, 1 = NDCG.  cutoff 0 means no cutoff (metric.h:65-67).    */
double qro_eval_dataset(int metric, const float *labels, const double *scores,
                        const uint64_t *qoff, size_t nq, size_t cutoff);
/* ndcg.cc:60-93 / dcg.cc:59-83.  out = packed upper triangle, n(n+1)/2      */
void qro_jacobian(int metric, const float *sorted_labels, size_t n,
                  size_t cutoff, double *out);

/* ---- pseudo-responses ---------------------------------------------------- */
/* lambdamart.cc:62-152 (sample_presence == NULL path)                        */
void qro_lambdas(int metric, const float *labels, const double *scores,
                 const uint64_t *qoff, size_t nq, size_t cutoff,
                 double *lambda, double *weight);
/* mart.cc:418-431                                                            */
void qro_residuals(const float *labels, const double *scores, size_t n,
                   double *out);

/* ---- binning (mart.cc:117-176, radix.cc:28-73, rtnode_histogram.cc:227-253) */
void qro_argsort_f32(const float *v, size_t n, uint64_t *idx);
/* thr: [F][cap] with cap = nthresholds ? nthresholds+1 : N+1; thr_size[F]    */
void qro_thresholds(const float *colmajor, size_t N, size_t F,
                    size_t nthresholds, float *thr, uint64_t *thr_size,
                    size_t cap);
/* stmap: [F][N] (u32), count0: [F][cap] cumulative root counts               */
void qro_binmap(const float *colmajor, size_t N, size_t F, const float *thr,
                const uint64_t *thr_size, size_t cap, uint32_t *stmap,
                uint64_t *count0);

/* ---- histograms (rtnode_histogram.cc:41-87, 172-217) --------------------- */
/* sum/count: [F][cap] cumulative; sampleids NULL == identity 0..n-1.         */
/* returns squares_sum_                                                       */
double qro_hist_build(const uint32_t *stmap, size_t N, size_t F,
                      const uint64_t *thr_size, size_t cap,
                      const double *labels, const uint64_t *sampleids,
                      size_t nsamples, double *sum, uint64_t *count);
void qro_hist_subtract(size_t F, const uint64_t *thr_size, size_t cap,
                       const double *psum, const uint64_t *pcount,
                       const double *lsum, const uint64_t *lcount, double *rsum,
                       uint64_t *rcount);

/* ---- split scan (rt.cc:257-312) over features [f0, f1) ------------------- */
typedef struct {
  double score;      /* -1 (initvar) when no valid split                     */
  uint64_t feature;  /* UINT64_MAX when none                                 */
  uint64_t thr_id;
  uint64_t lcount, rcount;
} qro_split_t;
void qro_split_find(size_t f0, size_t f1, const uint64_t *thr_size, size_t cap,
                    const double *sum, const uint64_t *count, uint64_t minls,
                    qro_split_t *out);

/* ---- trees ---------------------------------------------------------------- */
typedef struct {
  int32_t feature;   /* -1 = leaf                                            */
  int32_t thr_id;
  float threshold;
  int32_t left, right;
  double value;      /* RTNode::avglabel (leaf output after update_output)   */
  double deviance;
  uint64_t nsamples;
} qro_node_t;

typedef struct {
  size_t N, F, cap;
  const float *colmajor;       /* [F][N] raw features                        */
  const uint32_t *stmap;       /* [F][N]                                     */
  const float *thr;            /* [F][cap]                                   */
  const uint64_t *thr_size;    /* [F]                                        */
} qro_train_data_t;

/* rt.cc:49-90,154-160 + split :209-362 + rtnode.cc:34-46.                    */
/* nodes: capacity 2*nleaves+1; leaf_of_doc[N] = index into DFS leaf order    */
/* (or -1); leaf_nodes[nleaves] = node index of each DFS leaf.                */
/* f0,f1: feature range considered by the split scan (whole = 0,F).           */
/* split_log (optional, capacity nleaves): records of accepted splits.        */
/* returns number of nodes; *nleaves_out = number of leaves.                  */
size_t qro_tree_fit(const qro_train_data_t *d, const double *labels,
                    size_t nleaves, uint64_t minls, qro_node_t *nodes,
                    int32_t *leaf_of_doc, int32_t *leaf_nodes,
                    size_t *nleaves_out, qro_split_t *split_log,
                    size_t *nsplits_out);
/* ot.cc:32-201: oblivious tree, nodes in heap order (2i+1, 2i+2).            */
size_t qro_oblivious_fit(const qro_train_data_t *d, const double *labels,
                         size_t depth, uint64_t minls, qro_node_t *nodes,
                         int32_t *leaf_of_doc, int32_t *leaf_nodes,
                         size_t *nleaves_out, qro_split_t *split_log,
                         size_t *nsplits_out);
/* rt.cc:165-207.  weights NULL = MART mean; else LambdaMART Newton step.     */
void qro_update_output(qro_node_t *nodes, const int32_t *leaf_nodes,
                       size_t nleaves, const int32_t *leaf_of_doc, size_t N,
                       const double *pseudo, const double *weights);
/* rtnode.h:134-152: walk one tree. x + f*stride is feature f.                */
double qro_tree_score(const qro_node_t *nodes, const float *x, size_t stride);
/* mart.cc:447-468: scores[i] += shrinkage * tree(x_i)                        */
void qro_update_scores(const qro_node_t *nodes, const float *data, size_t N,
                       size_t F, int colmajor, double shrinkage, double *scores);

/* ---- full training loop (mart.cc:208-416) -------------------------------- */
typedef struct {
  int algo;           /* 0 MART, 1 LAMBDAMART, 2 OBVMART, 3 OBVLAMBDAMART   */
  size_t ntrees;
  double shrinkage;
  size_t nthresholds; /* 0 = every unique value                             */
  size_t nleaves;     /* leaf-wise                                           */
  size_t depth;       /* oblivious                                           */
  uint64_t minls;
  size_t esr;         /* valid_iterations_ (0 = off)                         */
  int metric;         /* 0 DCG, 1 NDCG                                       */
  size_t cutoff;
} qro_params_t;

typedef struct {
  size_t ntrees;          /* trees kept after rollback                       */
  size_t ntrees_built;
  size_t max_nodes;       /* stride of nodes per tree                        */
  qro_node_t *nodes;      /* [ntrees_built][max_nodes] (malloc'd)            */
  uint64_t *nnodes;       /* [ntrees_built]                                  */
  double *train_metric;   /* [ntrees_built]                                  */
  double *valid_metric;   /* [ntrees_built] (if validation)                  */
  double *train_scores;   /* [N] final                                       */
  float *thr;             /* [F][cap]                                        */
  uint64_t *thr_size;
  size_t cap;
  size_t best_model;
  double *iter_seconds;   /* [ntrees_built] wall time per boosting iteration */
} qro_model_t;

/* row-major inputs like Dataset (dataset.h:65-67).  valid may be NULL.       */
int qro_train(const qro_params_t *p, const float *train_rowmajor,
              const float *train_labels, const uint64_t *train_qoff, size_t nq,
              size_t N, size_t F, const float *valid_rowmajor,
              const float *valid_labels, const uint64_t *valid_qoff,
              size_t vnq, size_t vN, qro_model_t *out);
void qro_model_free(qro_model_t *m);

/* ---- inference (ensemble.cc:111-118, ltr_algorithm.cc:44-52) ------------- */
void qro_ensemble_score(const qro_node_t *nodes, const uint64_t *nnodes,
                        size_t ntrees, size_t max_nodes, const double *weights,
                        const float *rowmajor, size_t N, size_t F,
                        double *scores);
/* generate_oblivious.cc:237-324: bit-interleaved oblivious scorer            */
void qro_oblivious_score(const uint32_t *feat, const float *thr,
                         const double *leaves, const float *weights,
                         size_t ntrees, size_t depth, const float *rowmajor,
                         size_t N, size_t F, double *scores);

void qro_set_threads(int n);

/* Self-checks (qr_oracle.c, top): how many times since the last call two views the
 * oracle holds of one fact disagreed (0 in every healthy run), and the first one's text. */
int qro_self_check(char *msg, size_t n);

#ifdef __cplusplus
}
#endif
#endif
