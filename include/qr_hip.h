/*
 * qr_hip.h -- C-ABI of the MI355X (gfx950) LambdaMART/GBRT training + scoring
 * device layer.  Plain pointers and sizes only; no C++/torch types cross it.
 *
 * QuickRank has no FFI layer: the seam is the set of protected virtual hooks
 * that Mart::learn calls in order.  Each entry point
 * below names the reference call site it stands behind (file:line relative to
 * the upstream hpclab/quickrank tree) so a maintainer can bind it from
 * Mart/LambdaMart/RegressionTree/ObliviousRT/Ensemble.
 *
 * Conventions
 *   - every function returns 0 on success, a QR_ERR_* code otherwise;
 *     qr_last_error(ctx) gives the message (the host CLI keeps the reference
 *     behaviour: message on stderr + exit(EXIT_FAILURE), e.g. dataset.cc:37-43)
 *   - the context owns all device memory; host buffers belong to the caller
 *   - one host thread drives a context; calls are synchronous at return unless
 *     the name ends in _async; a context is not thread-safe (the reference is
 *     single-caller too: one thread drives OpenMP regions, rt.cc:247)
 *   - types follow include/types.h:28-35 of the reference: Feature/Label f32,
 *     Score/MetricScore f64; counts are 64-bit at the boundary
 *   - there is NO CPU fallback: without a visible gfx950 device
 *     qr_ctx_create fails with QR_ERR_NO_DEVICE
 */
#ifndef QR_HIP_H_
#define QR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QR_OK 0
#define QR_ERR_NO_DEVICE 1
#define QR_ERR_HIP 2
#define QR_ERR_ARG 3
#define QR_ERR_STATE 4
#define QR_ERR_UNSUPPORTED 5

#define QR_MAX_BINS 256 /* threshold slots per feature: 255 + FLT_MAX sentinel */

#define QR_ALGO_MART 0          /* mart.cc                                    */
#define QR_ALGO_LAMBDAMART 1    /* lambdamart.cc                              */
#define QR_ALGO_OBVMART 2       /* obliviousmart.cc                           */
#define QR_ALGO_OBVLAMBDAMART 3 /* obliviouslambdamart.cc                     */

#define QR_METRIC_DCG 0  /* dcg.cc   */
#define QR_METRIC_NDCG 1 /* ndcg.cc  */

typedef struct qr_ctx qr_ctx;

/* One tree node as the host RTNode mirror needs it (rtnode.h:37-52).          */
typedef struct {
  int32_t feature;  /* featureidx, -1 for a leaf (uint_max in the reference)   */
  int32_t thr_id;   /* threshold slot                                          */
  float threshold;  /* RTNode::threshold                                       */
  int32_t left, right; /* node indices, -1 for a leaf                          */
  double value;     /* RTNode::avglabel (leaf output after update_output)      */
  double deviance;  /* RTNode::deviance                                        */
  uint64_t nsamples;
} qr_node_t;

/* A best-split record (rt.cc:257-312), also the unit exchanged between GPUs. */
typedef struct {
  double score;     /* -1 when the node has no valid split (rt.cc:213,312)     */
  uint32_t feature; /* global feature index, UINT32_MAX when none              */
  uint32_t thr_id;
  uint64_t lcount;
  uint64_t rcount;
} qr_split_t;

/* ---- context ---------------------------------------------------------------- */
int qr_ctx_create(int device, qr_ctx **out);
void qr_ctx_destroy(qr_ctx *ctx);
const char *qr_last_error(const qr_ctx *ctx);
/* run every launch on the caller's HIP stream (e.g. torch's current stream)    */
int qr_ctx_set_stream(qr_ctx *ctx, void *hip_stream);
/* the HIP stream the context launches on (a hipStream_t): what a host hands to      */
/* ncclAllReduce / ncclAllGather for the exchanges of the multi-GPU protocols below   */
int qr_ctx_stream(qr_ctx *ctx, void **hip_stream_out);
/* feature-block sharding for the multi-GPU path: this   */
/* rank owns features [rank*ceil(F/world), (rank+1)*ceil(F/world)) of the bins.  */
/* Must be called before qr_bins_build.  Default rank 0 / world 1.              */
int qr_ctx_set_shard(qr_ctx *ctx, int rank, int world);
/* document sharding (the second multi-GPU layout, section "document-sharded      */
/* protocol" below): this rank holds whole queries of its own -- n_global /       */
/* q_global documents / queries over all ranks -- and every feature of them.      */
/* Must precede the bin build.                                                    */
int qr_ctx_set_doc_shard(qr_ctx *ctx, int rank, int world, uint64_t n_global,
                         uint64_t q_global);
/* Drains the context's stream.  A tree whose enqueued steps did not suffice is carried on   */
/* first -- EXCEPT on a document-sharded context whose last tree ended behind a guessed      */
/* number of steps (qr_tree_batch_*, qr_tree_end before qr_tree_batch_settle): carrying that */
/* tree on takes all-reduces only the caller can enqueue, so qr_synchronize drains what is   */
/* enqueued and returns QR_OK with the tree possibly incomplete (its leaf and score kernels   */
/* returned at once).  qr_tree_pending says so; qr_tree_batch_settle (+ the steps it asks    */
/* for) must come before the tree's records or the scores are used.                          */
int qr_synchronize(qr_ctx *ctx);
/* Debugging aid: drains the DEVICE and returns the error of any launch since the last call; */
/* in a library built with -DQR_DEBUG_CHECKS also the first bounds violation a growth kernel  */
/* recorded (the checked stores are skipped, not performed).  quickrank_amd/_capi.py calls it */
/* after every C-ABI call when QR_DEBUG=1.                                                    */
int qr_debug_check(qr_ctx *ctx);
/* *count = how often a polled read-back (a tree's records, an iteration's scalars) did not  */
/* fit the sequence number it was polled for at first sight and was read again (the records   */
/* carry a tag of themselves, QrNodeWire in qr_internal.h).  Expected: 0 -- the tests and     */
/* tests/tools/abort_hunt.py assert it, so that a recurrence is caught rather than absorbed.  */
int qr_readback_retries(qr_ctx *ctx, unsigned long long *count);
/* *pending = 1: the last tree ended behind a guess that nobody has looked at yet             */
int qr_tree_pending(qr_ctx *ctx, int *pending);

/* ---- data: replaces Dataset -> VerticalDataset (vertical_dataset.cc:29-66)    */
/*      and Mart::init (mart.cc:117-176) + RTRootHistogram (rtnode_histogram.cc */
/*      :227-253)                                                                */
/* rowmajor: f32 [N][F] (Dataset::at, dataset.h:65-67) in host memory, or rows    */
/* already resident on the context's device (copied either way); qoff: [Q+1];     */
/* labels and qoff: host memory                                                   */
int qr_dataset_upload(qr_ctx *ctx, const float *rowmajor, size_t N, size_t F,
                      const float *labels, const uint64_t *qoff, size_t Q);
/* optional validation set (mart.cc:231-233, 354-360).  rowmajor: host f32 [N][F]  */
/* with the validation file's OWN width F (the SVMLight reader sets it to the       */
/* largest feature id of each file): columns the training set has and this one      */
/* lacks read as 0 (SVMLight's meaning of an absent feature), columns beyond the     */
/* training width are never tested by a tree and are dropped.                        */
int qr_valid_upload(qr_ctx *ctx, const float *rowmajor, size_t N, size_t F,
                    const float *labels, const uint64_t *qoff, size_t Q);
/* thresholds + uint8 bin map: the FAST path, for up to 255 thresholds per feature. */
/* nthresholds in [1,255], or 0 (= every unique value) while no feature has more   */
/* than 255 uniques.  Anything beyond that -- nthresholds > 255, or 0 on a column  */
/* with more distinct values -- returns QR_ERR_UNSUPPORTED here and is served by   */
/* qr_bins_build_wide below (the hosts fall through to it automatically).          */
/* thr_out: host [F][QR_MAX_BINS] (padded with FLT_MAX), thr_size_out: [F].        */
int qr_bins_build(qr_ctx *ctx, size_t nthresholds, float *thr_out,
                  uint32_t *thr_size_out);
/* The same in three steps, for document-sharded contexts whose thresholds must   */
/* come from the WHOLE training set (mart.cc:147-169 runs over every document):  */
/*   qr_bins_stats on every rank -> all_gather the three arrays ->                */
/*   qr_thresholds_from_stats (pure host, same result on every rank) ->          */
/*   qr_bins_build_with.                                                         */
/* limit = nthresholds ? nthresholds + 1 : 256.  vals_out: [F][limit+1] f32 bit   */
/* patterns of the first distinct values met, cnt_out: [F] number of distinct    */
/* values (saturating at limit+1), minmax_out: [F][2] radix-ordered keys.        */
int qr_bins_stats(qr_ctx *ctx, size_t nthresholds, uint32_t *vals_out,
                  uint32_t *cnt_out, uint32_t *minmax_out);
/* the arrays of `nranks` ranks concatenated rank-major                          */
int qr_thresholds_from_stats(size_t F, size_t nthresholds, size_t nranks,
                             const uint32_t *vals, const uint32_t *cnt,
                             const uint32_t *minmax, float *thr_out,
                             uint32_t *thr_size_out);
/* bin map for caller-supplied thresholds ([F][QR_MAX_BINS], rows end in FLT_MAX) */
int qr_bins_build_with(qr_ctx *ctx, const float *thr, const uint32_t *thr_size);
/* More than 255 thresholds per feature -- QuickRank's default `--num-thresholds 0` on   */
/* real-valued columns (every distinct value a candidate, quicklearn.cc:103,              */
/* mart.cc:147-158) and any --num-thresholds > 255 (mart.cc:159-169).  Same thresholds,   */
/* same bin rule; rows are ragged: feature f has thr_size[f] slots, *cells_out in all,    */
/* the longest *max_slots_out.  u32 bins, histograms of any row length (k_wide.hip; rows   */
/* of up to 1152 slots take the LDS-tiled kernel).  Single-GPU contexts and feature-      */
/* sharded ones (qr_ctx_set_shard: the rank builds its own feature range, cells_out counts */
/* its own cells; the phase calls qr_tree_begin / decide / apply / end work unchanged, the   */
/* go-left mask carries one more word -- qr_exchange_buffers reports the size).  Document-   */
/* sharded contexts: qr_bins_build_wide_with.  Everything after the bin build -- lambdas,         */
/* qr_tree_fit, qr_oblivious_fit, score updates, metrics -- is called as on a u8 context.    */
int qr_bins_build_wide(qr_ctx *ctx, size_t nthresholds, size_t *cells_out, size_t *max_slots_out);
/* The same bins for GIVEN thresholds (the wide counterpart of qr_bins_build_with): ragged rows,  */
/* feature f's thr_size[f] ascending values one after the other, each row ending in FLT_MAX.      */
/* Single-GPU contexts and DOCUMENT-SHARDED ones (round 4): every rank passes the thresholds of    */
/* the WHOLE training set -- qr_bins_stats_wide on every rank -> [all_gather] ->                    */
/* qr_thresholds_from_stats_wide -> here -- and then drives the document-sharded protocol below    */
/* unchanged (qr_tree_begin / decide / apply: the histogram exchange buffer holds 2 x cells + 2 x   */
/* world int64, at most 4M cells in all; qr_tree_batch_* and the level-wise calls need u8 bins).   */
int qr_bins_build_wide_with(qr_ctx *ctx, const float *thr, const uint32_t *thr_size, size_t *cells_out,
                            size_t *max_slots_out);
/* Column statistics of this rank's documents for those thresholds (mart.cc:136-152 on its shard): */
/* vals_out u32 [F][limit] = the first `limit` distinct values of every column in sorted order (raw */
/* f32 bits), cnt_out [F] = how many there are (limit + 1: more than `limit`), minmax_out [F][2] =  */
/* the radix keys (radix.cc:28-30) of the smallest / largest value.  limit = nthresholds + 1, or     */
/* the most distinct values per column the caller will gather when nthresholds == 0.                 */
int qr_bins_stats_wide(qr_ctx *ctx, size_t limit, uint32_t *vals_out, uint32_t *cnt_out, uint32_t *minmax_out);
/* mart.cc:140-169 over the union of `nranks` ranks' statistics ([rank][F][limit] etc., as gathered): */
/* ragged rows into thr_out (thr_cap floats; NULL: sizes only), thr_size_out [F], *cells_out = their   */
/* total.  QR_ERR_UNSUPPORTED: nthresholds == 0 and a column with more than `limit` distinct values.   */
/* Pure host code: no context, no GPU.                                                                 */
int qr_thresholds_from_stats_wide(size_t F, size_t nthresholds, size_t nranks, size_t limit,
                                  const uint32_t *vals, const uint32_t *cnt, const uint32_t *minmax,
                                  float *thr_out, size_t thr_cap, uint32_t *thr_size_out, size_t *cells_out);
/* thresholds of a binned context (u8 or wide) as ragged rows: thr_out holds              */
/* sum(thr_size) floats, feature after feature; thr_size_out [F].  NULL = skip.           */
int qr_thresholds_read(qr_ctx *ctx, float *thr_out, uint32_t *thr_size_out);
/* debug/parity: bin ids as u32 [N][F] row-major (u8 or wide contexts)                    */
int qr_bins_read_u32(qr_ctx *ctx, uint32_t *out);
/* debug/parity: bin ids as u8 [N][F] row-major (global features; features this  */
/* rank does not own read back as 0xFF)                                          */
int qr_bins_read(qr_ctx *ctx, uint8_t *out);
/* debug/parity: the same from the FEATURE-MAJOR copy of the u8 bins (the one the partition    */
/* of rt.cc:325-334 and the leaf walk read); tests/tools/repro_first_tree.py compares the two  */
int qr_bins_read_fm(qr_ctx *ctx, uint8_t *out);
/* verify-after-write of the resident u8 bin map (round 6): every cell recomputed from the raw   */
/* rows on the device and compared with the block rows and with the feature-major copy; the      */
/* counts of cells that differ.  qr_bins_build / qr_bins_build_with run it themselves and build  */
/* the map again (up to three times) when it reports any: RTRootHistogram's stmap                */
/* (rtnode_histogram.cc:227-253) is written once and read by every tree.                         */
int qr_bins_verify(qr_ctx *ctx, unsigned long long *bad_block_rows, unsigned long long *bad_feature_major);
/* test aid: zero the bin rows of documents [first_doc, first_doc + ndocs) in the block rows     */
/* (which = 0) or in the feature-major copy (which = 1) -- what a lost store looks like;         */
/* which = 2: the next `first_doc` builds of the map lose documents [8, 16) of the block rows    */
/* behind their kernels (the rebuild path of qr_bins_build); which = 3: the next tree's records */
/* are read as if the root histogram had lost `first_doc` documents (qr_tree_nodes then fails     */
/* with QR_ERR_HIP: the records' counts do not add up)                                            */
int qr_debug_bins_clobber(qr_ctx *ctx, int which, size_t first_doc, size_t ndocs);

/* ---- model state: scores_on_training_ (mart.cc:121), pseudoresponses_,        */
/*      instance_weights_ (lambdamart.cc:34-39)                                  */
int qr_scores_reset(qr_ctx *ctx);                       /* zero train+valid      */
int qr_scores_set(qr_ctx *ctx, const double *scores);   /* host [N] -> device    */
int qr_scores_get(qr_ctx *ctx, double *scores);         /* device -> host [N]    */
int qr_valid_scores_get(qr_ctx *ctx, double *scores);
/* restart from a loaded model (mart.cc:237-253): validation scores of that model */
int qr_valid_scores_set(qr_ctx *ctx, const double *scores);
int qr_pseudo_get(qr_ctx *ctx, double *lambda, double *weight); /* NULL = skip   */
int qr_pseudo_set(qr_ctx *ctx, const double *lambda, const double *weight);

/* ---- pseudo-responses --------------------------------------------------------*/
/* LambdaMart::compute_pseudoresponses (lambdamart.cc:62-152) for metric@cutoff. */
/* cutoff 0 = no cutoff (metric.h:65-67).  Also leaves the training metric of    */
/* the CURRENT scores in the context (same ranking), see qr_metric_last.         */
/* (The iteration's scalars -- quantisation scale, the root's sums, that metric -- */
/* are finished lazily: by workgroups riding in the root scan launch of the       */
/* qr_tree_fit / qr_oblivious_fit that follows, or, when anything else asks first */
/* (qr_metric_last, the phase calls, ...), in a launch of their own.  Nothing a   */
/* caller can observe depends on which; a host that reads qr_metric_last AFTER it  */
/* has enqueued the tree saves that launch.  csrc/qr_prep.h)                       */
int qr_lambda_compute(qr_ctx *ctx, int metric, size_t cutoff);
/* Mart::compute_pseudoresponses (mart.cc:418-431)                               */
int qr_residual_compute(qr_ctx *ctx);

/* ---- metric: Metric::evaluate_dataset (metric.h:77-106) over the device       */
/*      scores.  which: 0 = training set, 1 = validation set.                    */
int qr_metric_eval(qr_ctx *ctx, int which, int metric, size_t cutoff,
                   double *out);
/* training metric of the scores the last qr_lambda_compute ranked (the lambda   */
/* kernel evaluates it on the way: same ranking, no second sort).  Waits for the */
/* lambda pass only (pinned snapshot + event), not for work enqueued after it.   */
int qr_metric_last(qr_ctx *ctx, double *out);

/* ---- regression tree: RegressionTree::fit (rt.cc:49-90) + split (:209-362) +  */
/*      RTNodeHistogram (rtnode_histogram.cc:41-87,172-217) + update_output      */
/*      (rt.cc:165-207) on the current pseudo-responses.                         */
/* newton != 0: LambdaMART leaf = sum(lambda)/sum(weight) (rt.cc:186-207),       */
/* else MART mean (rt.cc:165-184).  nodes_out: capacity 2*nleaves+1, creation    */
/* order (root 0; each split appends left,right).  leaf ids are DFS left-first   */
/* (rtnode.cc:34-46).                                                            */
/* nodes_out == NULL and nnodes_out == NULL: enqueue only (no wait); fetch the    */
/* records later with qr_tree_nodes.                                             */
int qr_tree_fit(qr_ctx *ctx, size_t nleaves, uint64_t minls, int newton,
                qr_node_t *nodes_out, size_t *nnodes_out);
/* records of the last fitted tree.  The tree's last kernel writes them into      */
/* pinned host memory, then a sequence number this call polls: it waits for the   */
/* tree only -- not for work enqueued after it (score update, the next lambdas).  */
/* They stay valid until the next tree is fitted: a host may enqueue the next      */
/* iteration's qr_lambda_compute first and fetch them under it.   */
/* The step count of an enqueued tree is a guess; a tree the guess   */
/* cut short is completed here -- and by every other call that reads or builds on */
/* its results -- before anything is returned: callers never see a partial tree.  */
/* The records' counts are checked before they are handed out (round 6): the root   */
/* holds the documents the tree was grown on, every internal node as many as its    */
/* two children (rtnode.h:97-107: a node's count is its histogram's) -- QR_ERR_HIP  */
/* otherwise: a histogram launch lost stores (profiles/r06_hunt.md).                */
int qr_tree_nodes(qr_ctx *ctx, qr_node_t *nodes_out, size_t *nnodes_out);
/* --subsample (mart.cc:287-329, lambdamart.cc:85-102): every iteration fits its  */
/* tree on a fresh uniform sample of the training documents: subsample > 1 = that */
/* many, < 1 = that fraction (rounded down), 1 = all.  The next                   */
/* qr_lambda_compute / qr_residual_compute draws the sample; lambdas are computed */
/* on the queries cleaned of the other documents, the tree and its leaf outputs   */
/* see the sample only, qr_scores_update updates every document.  The reference   */
/* shuffles with a clock-seeded engine (and indexes the cleaned scores without    */
/* the query offset, lambdamart.cc:94); here the sample is a pure function of     */
/* (seed, iteration) and the scores are the query's own.  Call after the bin      */
/* build.  qr_metric_last then reports the cleaned rankings: use qr_metric_eval.  */
int qr_subsample_set(qr_ctx *ctx, float subsample, uint64_t seed);
/* document-sharded contexts: `subsample` refers to ALL ranks' documents, of which   */
/* this rank's are [first_doc, first_doc + N).  A document's key is a function of its  */
/* GLOBAL index, so every rank finds the same sample (the one a single GPU draws from */
/* the whole set) without an exchange, and keeps its own part of it.                  */
int qr_subsample_set_doc(qr_ctx *ctx, float subsample, uint64_t seed, size_t first_doc);
/* test aid: the sample's keys ANDed with `mask` (all ones by default) -- narrow keys meet equal ones, */
/* which the selection breaks by ascending document as a stable sort of (key, document) would        */
int qr_debug_sample_key_mask(qr_ctx *ctx, uint32_t mask);
/* --max-features (rt.cc:222-243): every node's split search sees a random       */
/* subset of the features: max_features > 1 = that many, < 1 = that fraction     */
/* (rounded up), 1 = all.  The reference draws it from a clock-seeded engine at  */
/* every split; here it is a pure function of (seed, tree number, node, feature) */
/* -- reproducible, identical on every rank.  Leaf-wise trees only.              */
int qr_tree_set_max_features(qr_ctx *ctx, float max_features, uint64_t seed);
/* ObliviousRT::fit (ot.cc:32-201): nodes_out in heap order (2i+1, 2i+2),        */
/* capacity 2^(depth+1)-1; absent nodes have feature == -2.                      */
int qr_oblivious_fit(qr_ctx *ctx, size_t depth, uint64_t minls, int newton,
                     qr_node_t *nodes_out, size_t *nnodes_out);
/* Mart::update_modelscores (mart.cc:447-468) with the tree just fitted:         */
/* training scores through the leaf membership, validation scores by walking     */
/* the tree on raw f32 features.  shrinkage is f64 (mart.h).                     */
int qr_scores_update(qr_ctx *ctx, double shrinkage);

/* ---- multi-GPU split protocol -------------------------*/
/* The feature-sharded path runs qr_tree_fit as explicit phases so the caller    */
/* can put its collectives (RCCL via torch.distributed) in between, all on the   */
/* context's stream, no host sync:                                               */
/*   qr_tree_begin -> [all_gather recs] -> loop { qr_tree_decide ->              */
/*   [all_reduce mask] -> qr_tree_apply -> [all_gather recs] } -> qr_tree_end    */
/* recs: device buffer of 2 qr_split_t per rank (left, right child of the last   */
/* split; root in slot 0 after begin).  mask: device u32 words, bit d = doc at   */
/* position d of the split node's segment goes left (zeros on non-owners).       */
int qr_tree_begin(qr_ctx *ctx, size_t nleaves, uint64_t minls);
int qr_tree_decide(qr_ctx *ctx);
int qr_tree_apply(qr_ctx *ctx);
int qr_tree_end(qr_ctx *ctx, int newton, qr_node_t *nodes_out,
                size_t *nnodes_out);
/* device pointers for the exchange buffers (valid after qr_bins_build)           */
int qr_exchange_buffers(qr_ctx *ctx, void **recs_local, void **recs_all,
                        size_t *rec_bytes_per_rank, void **mask,
                        size_t *mask_bytes);

/* Feature-sharded OBLIVIOUS trees (ot.cc:32-201), one exchange pair per level:        */
/*   qr_obl_begin -> for level in 0 .. depth-1 { qr_obl_propose -> [all_gather recs] ->  */
/*   qr_obl_mark -> [all_reduce mask, int32 sum] -> qr_obl_apply } -> qr_tree_end        */
/* recs: the buffers of qr_exchange_buffers (every rank's best (feature, slot) of the    */
/* level; all ranks then pick the same one).  mask: here the go-left bit of every        */
/* DOCUMENT (all nodes of a level take the same split) followed by the left count of     */
/* every node of the level -- qr_obl_exchange_buffers gives the pointer and the size     */
/* (a multiple of 4 bytes) to reduce; only the owner of the chosen feature contributes.  */
int qr_obl_begin(qr_ctx *ctx, size_t depth, uint64_t minls);
int qr_obl_propose(qr_ctx *ctx, size_t level);
int qr_obl_mark(qr_ctx *ctx, size_t level);
int qr_obl_apply(qr_ctx *ctx, size_t level);
int qr_obl_exchange_buffers(qr_ctx *ctx, void **recs_local, void **recs_all,
                            size_t *rec_bytes_per_rank, void **mask, size_t *mask_bytes);
/* Document-sharded OBLIVIOUS trees: every rank holds every feature of its own documents, */
/* so what a level exchanges is its directly built children's histogram cells:              */
/*   qr_obl_begin -> [all_reduce hist (qr_doc_exchange_buffers), int64 sum] ->              */
/*   for level in 0 .. depth-1 { qr_obl_propose -> qr_obl_apply ->                          */
/*     [all_reduce the level's cells (qr_obl_level_exchange), int64 sum; not after the last */
/*      level: ot.cc:127 builds no histograms for the leaves] } ->                          */
/*   qr_tree_end -> [all_reduce leaf] -> qr_tree_leaves_finish                               */
/* (qr_obl_mark is not part of this protocol: the partition is local.)                      */
int qr_obl_level_exchange(qr_ctx *ctx, size_t level, void **cells, size_t *cells_i64);

/* ---- document-sharded protocol ------------------------------------------------*/
/* Each rank holds its own queries and all features; what is exchanged is the    */
/* node histogram itself -- exact fixed-point integers, so ONE int64 sum          */
/* all-reduce per histogram gives every rank the same bits as a single GPU would */
/* compute, and every rank then scans / decides redundantly with no further      */
/* exchange.  f64 quantities (sum / sum of squares of the pseudo-responses, leaf */
/* sums, the metric) travel as bit patterns in per-rank slots of the same int64  */
/* buffers (zeros elsewhere: sum == gather) and are added in rank order.         */
/*   per iteration:                                                              */
/*     qr_lambda_compute -> [all_reduce scal] -> qr_lambda_finish                */
/*     qr_tree_begin     -> [all_reduce hist]                                    */
/*     (nleaves-1) x { qr_tree_decide -> qr_tree_apply -> [all_reduce hist] }    */
/*     qr_tree_decide -> qr_tree_end(nodes_out = NULL) -> [all_reduce leaf]      */
/*     qr_tree_leaves_finish -> qr_scores_update                                 */
/* all on the context's stream (int64 sum all-reduces, element counts below).    */
int qr_lambda_finish(qr_ctx *ctx);
int qr_tree_leaves_finish(qr_ctx *ctx, int newton, qr_node_t *nodes_out,
                          size_t *nnodes_out);
/* The same trees with up to TWO splits per exchange (what qr_tree_fit does on one GPU --  */
/* rt.cc:58-90's loop with the most promising other heap entry split ahead of its turn,    */
/* invisible in the result -- cut at the all-reduces).  A tree of L leaves then costs      */
/* 1 + steps all-reduces (5-6 steps for L = 10) instead of L:                               */
/*     qr_tree_batch_begin(&steps) -> [all_reduce hist] -> qr_tree_batch_root               */
/*     steps x { qr_tree_batch_apply -> [all_reduce batch cells, qr_tree_batch_exchange]    */
/*               -> qr_tree_batch_decide(last = the final one of the sequence) }            */
/*     qr_tree_batch_settle(&incomplete): waits for the last control step; `steps` was a    */
/*       guess (the previous tree's count; every rank grows the same trees, so every rank   */
/*       guesses and settles alike) -- if incomplete, repeat { apply, all_reduce, decide }  */
/*       with last = 1 on the piece's final step and settle again                           */
/*     qr_tree_end(nodes_out = NULL) -> [all_reduce leaf] -> qr_tree_leaves_finish          */
/* qr_tree_batch_supported: 1 if the context can grow a tree of `nleaves` this way (u8      */
/* bins, no --max-features, 2 <= nleaves <= 255), else 0: use qr_tree_begin / decide / apply */
int qr_tree_batch_supported(qr_ctx *ctx, size_t nleaves);
int qr_tree_batch_begin(qr_ctx *ctx, size_t nleaves, uint64_t minls, size_t *steps_out);
int qr_tree_batch_root(qr_ctx *ctx);
int qr_tree_batch_apply(qr_ctx *ctx);
int qr_tree_batch_decide(qr_ctx *ctx, int last);
int qr_tree_batch_settle(qr_ctx *ctx, int *incomplete, size_t *steps_used);
/* device pointer + int64 element count of the batch's cells (valid after batch_begin)     */
int qr_tree_batch_exchange(qr_ctx *ctx, void **cells, size_t *cells_i64);
/* device pointers + int64 element counts; hist/scal valid after the bin build,  */
/* leaf after qr_tree_begin                                                      */
int qr_doc_exchange_buffers(qr_ctx *ctx, void **hist, size_t *hist_i64, void **scal,
                            size_t *scal_i64, void **leaf, size_t *leaf_i64);

/* ---- parity/debug read-backs --------------------------------------------------*/
/* cumulative histogram of a node slot of the LAST fitted tree, as the           */
/* reference keeps it (sumlbl f64 / count u64, [F][QR_MAX_BINS], owned features  */
/* only, others zero).  The device accumulates order-independent fixed-point     */
/* sums; sum_out = fixed * 2^-scale_exp.                                         */
int qr_node_hist_read(qr_ctx *ctx, int node, double *sum_out,
                      uint64_t *count_out);
/* the same as ragged rows (sum(thr_size) cells, feature after feature): u8 or wide */
int qr_node_hist_read_ragged(qr_ctx *ctx, int node, double *sum_out, uint64_t *count_out);
/* sample ids (ascending doc ids) of a node of the last fitted tree              */
int qr_node_samples_read(qr_ctx *ctx, int node, uint32_t *ids_out,
                         size_t *n_out);
/* accepted splits of the last fitted tree in order                              */
int qr_tree_split_log(qr_ctx *ctx, qr_split_t *out, size_t *n_out);
/* per-query metric of the last qr_lambda_compute / qr_metric_eval(which=0)      */
int qr_metric_per_query(qr_ctx *ctx, double *out);
/* rank permutation (pos_of_rank, rankedresults.cc:27-41) of the last            */
/* qr_lambda_compute: host u32 [N], query-local doc index per rank.  The first    */
/* `cutoff` ranks of a query are std::sort's bit for bit; beyond the cutoff (where */
/* no discount applies) the documents come in non-increasing score order with an  */
/* unspecified order among equal scores, unless QR_EXACT_TAIL=1 is set.           */
int qr_ranks_read(qr_ctx *ctx, uint32_t *out);

/* ---- inference: Ensemble::score_instance (ensemble.cc:111-118) over           */
/*      LTR_Algorithm::score_dataset (ltr_algorithm.cc:44-52)                    */
/* nodes: [ntrees][max_nodes] in the qr_node_t layout; weights f64 [ntrees].     */
/* qr_ensemble_set_depth_order(1) BEFORE the upload (or QR_SCORE_DEPTH_ORDER=1): the trees  */
/* are walked and their contributions added in ascending order of depth (stable), so that  */
/* the trees a wave walks in lockstep end together -- 1.3x on leaf-wise 64-leaf trees.  The */
/* f64 sum is then taken in another order than ensemble.cc:111-118's: equal to f64 rounding */
/* (north_star: 1e-5), not bit for bit.  Default 0: the model's order, bit for bit.         */
int qr_ensemble_set_depth_order(qr_ctx *ctx, int on);
int qr_ensemble_upload(qr_ctx *ctx, const qr_node_t *nodes, size_t ntrees,
                       size_t max_nodes, const double *weights);
/* rowmajor: HOST f32 [N][F]; scores_out: host f64 [N].  Timing of the kernel    */
/* alone (features resident) is returned in *kernel_ms when non-NULL.  A matrix   */
/* narrower than the model's largest tested feature is zero-padded on the way up   */
/* (an absent SVMLight feature is 0); wider is fine.                              */
int qr_ensemble_score(qr_ctx *ctx, const float *rowmajor, size_t N, size_t F,
                      double *scores_out, float *kernel_ms);
/* same, features already resident on the device (device pointer); F must cover   */
/* every feature the model tests (QR_ERR_ARG otherwise: nothing to pad in place)  */
int qr_ensemble_score_device(qr_ctx *ctx, const void *d_rowmajor, size_t N,
                             size_t F, void *d_scores_out);

/* Ensemble::partial_scores_instance (ensemble.cc:120-131), the per-tree outputs     */
/* behind `--detailed` (driver.cc:326-347, 411-445): out is host f64 [N][ntrees],    */
/* out[d][t] = tree_t(x_d) * weight_t, or tree_t(x_d) alone when ignore_weights.      */
int qr_ensemble_partial_scores(qr_ctx *ctx, const float *rowmajor, size_t N, size_t F,
                               int ignore_weights, double *out);

/* ---- oblivious ensembles: the bit-interleaved scorer `quicklearn --generator    */
/*      oblivious` emits (generate_oblivious.cc:237-324).  feat/thr: [ntrees][depth]*/
/*      root level first; leaves: [ntrees][2^depth] DFS left-first; weights f32     */
/*      (generate_oblivious.cc:166 parses them with as_float); depths: per-tree     */
/*      number of levels actually used (NULL = all `depth`).                        */
int qr_oblivious_upload(qr_ctx *ctx, const uint32_t *feat, const float *thr,
                        const double *leaves, const float *weights,
                        const uint32_t *depths, size_t ntrees, size_t depth);
int qr_oblivious_score(qr_ctx *ctx, const float *rowmajor, size_t N, size_t F,
                       double *scores_out, float *kernel_ms);

/* ---- instrumentation ---------------------------------------------------------*/
/* HIP-event timing of the dominant kernel (root histogram build) accumulated    */
/* since the last reset: launches, total ms, algorithmic bytes per launch.       */
int qr_prof_reset(qr_ctx *ctx);
int qr_prof_get(qr_ctx *ctx, uint64_t *launches, double *total_ms,
                double *alg_bytes_per_launch);
/* on: bit 0 = time the root histogram launches, bit 1 = also the child launches, */
/* bit 2 (without bit 1) = the lambda pass's launch instead of the child launches  */
/* (LambdaMart::compute_pseudoresponses, lambdamart.cc:62-152; equal-length query  */
/* sets: one launch), read through qr_prof_get_child as well;                      */
/* bits 8..15 = k: events on every k-th root launch only (0 = every launch; an     */
/* evented launch costs the stream ~7.5 us)                                          */
int qr_prof_enable(qr_ctx *ctx, int on);
/* child-histogram launches (k_hist_batch) since the last reset; their algorithmic */
/* bytes depend on the trees: sum over splits of n_small * (F + 12), from the       */
/* records qr_tree_nodes returns                                                     */
int qr_prof_get_child(qr_ctx *ctx, uint64_t *launches, double *total_ms);
/* The second bound of the root histogram launch, measured in the caller's process   */
/* (bench.py's roofline.lds_atomic_bound): the launch issues one ds_add_u64 per       */
/* (document, accumulated column) -- rtnode_histogram.cc:172-204's `sumlbl[f][t] +=`  */
/* and `count[f][t]++` in one LDS atomic -- and a CU retires them at the rate this    */
/* microbenchmark reaches with nothing else in its loop: shader cycles per wave       */
/* instruction, the shader clock under that load, their quotient in ns, and the wave  */
/* instructions the root launch of the context's data set asks of one CU.             */
int qr_prof_lds_atomic(qr_ctx *ctx, double *cycles_per_instr, double *shader_ghz,
                       double *ns_per_instr, double *root_wave_instr_per_cu);

#ifdef __cplusplus
}
#endif
#endif /* QR_HIP_H_ */
