// qr_internal.h -- shared declarations of the gfx950 device layer (not part of
// the C-ABI; see include/qr_hip.h for the boundary).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/qr_hip.h"

// ---------------------------------------------------------------------------
// Fixed-point histogram accumulators.
// A (feature, bin) cell of a workgroup's LDS histogram is ONE u64:
//     cell = count * 2^QR_SB + sum_q        (sum_q signed, two's complement)
// where sum_q = sum of q_d = rint(lambda_d * 2^scale_exp), |q_d| < 2^33.
// A workgroup flushes after at most QR_DPW docs, so count < 2^15 and
// |sum_q| < 2^47: one ds_add_u64 per (doc, feature) carries gradient and count.
// Integer sums are associative => any accumulation order gives identical bits
// (deterministic, and structural ties between split candidates are preserved
// exactly -- SURVEY.md section 7 hard part 1).
// ---------------------------------------------------------------------------
#define QR_SB 49
#define QR_DPW 16384u     /* docs per workgroup between flushes              */
#define QR_QBITS 33       /* |q| < 2^33                                      */
#define QR_SLICE 1024u    /* doc-range alignment of hist workgroups          */
#ifndef QR_HIST_MIN_PER
#define QR_HIST_MIN_PER QR_SLICE /* fewest documents a histogram workgroup takes (A/B: 512 / 2048) */
#endif
#define QR_MAXBLK 64      /* max 64-feature blocks per rank (F <= 4096)      */
#ifndef QR_PART_SLICE
#define QR_PART_SLICE 2048u /* positions per partition workgroup             */
#endif

struct QrBlock {
  int f0;        // first global feature
  int fw;        // padded width: 16,32,48,64
  int nreal;     // real features in the block
  int lf0;       // first local feature index
  size_t off;    // byte offset of the block's [N][fw] u8 matrix in d_bins
};

// Work plan of one histogram launch: which workgroups serve which block.
struct alignas(16) QrPlan {
  int wg_start[QR_MAXBLK + 1];
  uint32_t per[QR_MAXBLK];  // docs per workgroup (multiple of QR_SLICE)
  int kmax;                 // flushes per workgroup
  int pad[2];               // sizeof == 528: keeps the dynamic-LDS base 16-B aligned
};

// The unit of histogram work is one document x 16 feature columns.  A launch is
// planned by its quantum q = units per workgroup: a block of width fw gives each
// of its workgroups ceil(q / (fw/16)) documents (rounded up to QR_SLICE), so every
// workgroup of every block -- and, in level-wise growth, of every node of the
// level -- carries the same load.
__host__ __device__ inline uint32_t qr_plan_quantum(unsigned long long units_total, int G) {
  if (G < 1) G = 1;
  const unsigned long long q = (units_total + (unsigned)G - 1) / (unsigned)G;
  return q < 1 ? 1u : (q > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)q);
}
__host__ __device__ inline int qr_plan_wsum(int nblocks, const QrBlock *blk) {
  int wsum = 0;
  for (int b = 0; b < nblocks; ++b) wsum += blk[b].fw / 16;
  return wsum;
}
__host__ __device__ inline void qr_make_plan(uint32_t n, int nblocks,
                                             const QrBlock *blk, uint32_t q,
                                             QrPlan *p) {
  p->wg_start[0] = 0;
  p->kmax = 1;
  for (int b = 0; b < nblocks; ++b) {
    // (32-bit arithmetic throughout: the control lane of k_decide_batch runs this, and
    // 64-bit divisions cost it ~200 cycles each)
    const uint32_t u = (uint32_t)(blk[b].fw / 16);
    uint32_t per = q / u + (q % u ? 1u : 0u);                 // ceil(q / u)
    per = per > 0x7FFFFC00u ? 0x7FFFFC00u : ((per + 255u) & ~255u);  // fine enough to fill the CUs evenly ...
    if (per < QR_HIST_MIN_PER) per = QR_HIST_MIN_PER;  // ... but never less than 1024 documents per workgroup
    const int weff = (int)(n / per + (n % per ? 1u : 0u));    // ceil(n / per)
    p->per[b] = per;
    p->wg_start[b + 1] = p->wg_start[b] + weff;
    const int k = (int)((per + QR_DPW - 1) / QR_DPW);
    if (k > p->kmax) p->kmax = k;
  }
}

// Node bookkeeping of the tree under construction (device resident).
struct QrNode {
  uint32_t begin, end;   // segment of positions in the order buffers
  int32_t buf;           // 0 = order A, 1 = order B, 2 = identity (root)
  int32_t hslot;         // histogram slot
  int32_t feature, thr_id;
  float threshold;
  int32_t left, right, parent;
  double sum, ss, deviance, value;
  uint64_t count;
  // best split of this node (merged over all features / ranks)
  double best_score;
  uint32_t best_f, best_t;
  uint64_t best_lc, best_rc;
  int32_t leaf_id;       // DFS leaf index, -1 for internal
  // batched growth (QR_BATCH): 1 = this node's split has been applied ahead of its
  // turn; its children wait under the provisional indices pre_l / pre_r
  int32_t pre;
  int32_t pre_l, pre_r;
  // batched growth keeps what applying the best split needs next to it
  int32_t best_lf;       // local index of best_f
  float best_thr;        // thresholds[best_f][best_t]
};

struct QrHeapItem {
  double key;
  int32_t val;
  int32_t pad;
};

#define QR_MAXNODES 1024

// Descriptor of the split being applied; written by the control kernel and
// consumed by the partition / histogram / scan kernels of the same step.
struct QrSplitDesc {
  int32_t active;          // 0 = nothing to do this step (tree finished)
  int32_t node, left, right;
  uint32_t begin, end;     // parent segment
  int32_t src_buf;         // buffer holding the parent segment
  int32_t dst_buf;
  uint32_t lcount, rcount;
  uint32_t feature, thr_id;  // global feature
  int32_t owner_local;     // local feature index on this rank, -1 if not owned
  int32_t small_is_left;
  // histogram work: direct build of `small`, sibling = parent - small
  int32_t small_node, big_node;
  int32_t parent_slot, small_slot, big_slot;
  uint32_t small_begin, small_n;
  int32_t pad;
};

// Document-sharded contexts: the part of the current split that lives on this
// rank (positions are local; QrSplitDesc::lcount / rcount / small_n are global).
struct QrLocalSplit {
  uint32_t lcount;               // local documents going left
  uint32_t small_begin, small_n; // local segment of the directly built child
  uint32_t pad;
};

// Level-wise (oblivious) growth splits every node of a level with the same
// (feature, slot): one descriptor per node, all served by single launches.
#define QR_MAXLEVEL 256
struct QrLevelNode {
  int32_t active;
  int32_t src_buf, dst_buf;
  int32_t small_is_left;
  uint32_t begin, end, lcount;     // the node's segment and its left count
  uint32_t small_begin, small_n;   // directly built child
  int32_t parent_slot, small_slot, big_slot;
  uint32_t q;                      // the level's plan quantum (same for all its nodes)
  uint32_t slot_base;              // first partial slot
  uint32_t part_first;             // first partition workgroup (global index)
  // batched leaf-wise growth: every node of the batch has its own split
  int32_t owner_local;             // local index of the split feature
  uint32_t thr_id;
  int32_t node, left, right;       // the split node and its children (final or provisional ids)
  int32_t spec;                    // 1 = applied ahead of its turn
  int32_t pad;
};

// Batched growth hands every workgroup of a step's launches its share ready-made
// (written by the control kernel), so a workgroup starts loading after ONE dependent
// read instead of map -> node descriptor -> plan.
struct QrHistWg {      // one histogram workgroup
  uint32_t begin;      // first position of its share in the list buffer
  uint32_t count;      // documents (0: nothing to do)
  uint32_t slot;       // first partial slot
  uint16_t block;      // feature block ...
  uint8_t buf;         // list buffer (0 / 1)
  uint8_t fw;          // ... its width ...
  uint32_t off256;     // ... and where its bins start (in units of 256 bytes)
  uint32_t pad;
};
struct QrPartWg {      // one partition workgroup
  uint32_t begin, n, lcount;   // the node's segment and left count (n == 0: nothing to do)
  uint32_t first, w;           // the node's first workgroup, this one's index within the node
  int32_t owner_local;
  uint32_t thr_id;
  uint8_t src_buf, dst_buf, small_is_left, pad;
};

struct QrScanWg {      // one (node of the batch, feature) workgroup of k_redscan
  uint32_t active;     // 0: nothing to do
  uint32_t slot0;      // first partial slot of the feature's block for this node
  uint32_t total;      // slots to sum (workgroups of the block x flushes per workgroup)
  uint32_t per, n;     // documents per workgroup / of the node (which slots were flushed)
  int32_t kmax;
  int32_t small_slot, big_slot, parent_slot, small_is_left;
  uint32_t col;        // column of the feature inside its block
  uint32_t part_first, part_nwg;  // the node's partition workgroups (their child sums: feature 0 adds them up)
  uint32_t pad;
};

// Leaf-wise growth applies up to QR_BATCH splits per step: the one the reference's
// loop needs next plus the most promising other candidates of the heap.
#define QR_BATCH 2

struct QrTreeState {
  int32_t nleaves_req;
  int32_t nnodes;
  int32_t taken;
  int32_t done;
  int32_t step;
  int32_t nsplits;
  uint64_t minls;
  int32_t heap_size;
  int32_t pad;
  QrHeapItem heap[QR_MAXNODES + 2];
  QrNode nodes[QR_MAXNODES];
  qr_split_t split_log[QR_MAXNODES];
  QrSplitDesc desc;
  QrLocalSplit loc;
  // oblivious (level-wise) growth: the split chosen for the current level
  uint32_t part_epoch;  // tag of the current partition's look-back granules
  int32_t obl_done, obl_level;
  uint32_t obl_f, obl_t;
  double obl_score;
  int32_t l_nodes;                 // nodes of the level being split
  int32_t l_owner_local;           // local index of the level's feature
  uint32_t l_hist_wgs, l_part_wgs; // workgroups the level's launches really use
  QrLevelNode lnode[QR_MAXLEVEL];
  int32_t next_prov, next_slot;    // batched growth: provisional node ids / histogram slots handed out
  int32_t spec_made, spec_used;    // statistics: splits applied ahead of their turn / later taken
  // The host enqueues a GUESSED number of steps (the previous tree's + 1, qr_k_tree_fit_batch):
  // steps that applied a batch, and whether the last control call of the enqueued sequence
  // still had a batch to apply -- the leaf / score kernels then leave and the host carries
  // the tree on (qr_k_tree_continue) when it fetches the records.
  int32_t real_steps, incomplete;
  // leaves in DFS order
  int32_t nleaves;
  int32_t leaf_nodes[QR_MAXNODES];
  uint32_t leaf_begin[QR_MAXNODES + 1];  // sorted by position
  int32_t leaf_at[QR_MAXNODES];          // node index per position-sorted leaf
  double leaf_value[QR_MAXNODES];        // per position-sorted leaf
  // pre-sorted lists (k_exact.hip), split search when the loop pops a node (k_xpop / k_xapply):
  // the node whose segments the step's scan launch walks (-1: none)
  int32_t xs_node, xs_buf;
  uint32_t xs_begin, xs_n;
  // ... the split k_xapply made exhausts the leaf budget: its children are leaves whatever they hold,
  // nobody will walk their segments -- k_xflag / k_xpart leave
  int32_t xs_last, xs_pad;
};

// Per-iteration scalars produced on the device.
struct QrScalars {
  unsigned long long maxabs_bits;  // max |pseudo-response| (bits of a double)
  int32_t scale_exp;               // q = rint(lambda * 2^scale_exp)
  int32_t pad;
  double scale, inv_scale;
  double root_ss;                  // sum of squares of the pseudo-responses
  double root_sum;                 // their plain f64 sum
  double metric_sum;               // sum of per-query metric
  double metric_gsum;              // document-sharded: the same over all ranks
  unsigned long long tag;          // pinned copy only: qr_scal_tag of the two metric sums and the sequence number
};
// (the pinned copy of the scalars validates itself like the tree records do: qr_metric_last)
#ifdef __HIPCC__
__host__ __device__
#endif
inline unsigned long long qr_scal_tag(const unsigned long long msum_bits, const unsigned long long gsum_bits, const int seq) {
  unsigned long long h = (unsigned long long)(unsigned)seq * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;
  h = (h ^ msum_bits) * 0x100000001B3ull;
  h ^= h >> 29;
  h = (h ^ gsum_bits) * 0x100000001B3ull;
  return h | 1ull;
}

// Host-pinned landing zone of the per-iteration read-backs (written by the kernels, each
// followed by a sequence number the host polls -- QrScalars::pad, QrNodesOut::pad[2]:
// the host never has to drain the stream to learn a metric or a tree).
struct QrNodesOut {
  int64_t nnodes;
  int64_t pad[7];   // [0] incomplete, [1] steps, [2] the sequence number the host polls, [3] the header's tag
  qr_node_t nodes[QR_MAXNODES];
};
// A record as it crosses to the host: qr_node_t with its four padding bytes (offset 20) carrying a
// TAG -- a hash of the record's other 44 bytes and of the tree's sequence number.  The host accepts
// a tree's records only when every tag fits (qr_tree_nodes re-reads until they do): a read-back
// that validates ITSELF, like the look-back words of the kernels, instead of trusting that the
// sequence number cannot reach host memory before a record written ahead of it (round 5's hunt met
// a leaf value that was not the tree's twice in ~170 processes; nothing on the device explains it).
struct QrNodeWire {
  int32_t feature, thr_id;
  float threshold;
  int32_t left, right;
  uint32_t tag;
  double value, deviance;
  uint64_t nsamples;
};
static_assert(sizeof(QrNodeWire) == sizeof(qr_node_t) && sizeof(qr_node_t) == 48, "the wire record is qr_node_t");
#ifdef __HIPCC__
__host__ __device__
#endif
inline uint32_t qr_node_tag(const QrNodeWire &r, const uint64_t seq) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(&r);
  uint32_t h = (uint32_t)seq * 0x9E3779B1u + (uint32_t)(seq >> 32) + 0x7F4A7C15u;
  for (int i = 0; i < 12; ++i) {
    if (i == 5) continue;  // (the tag's own place)
    h = (h ^ w[i]) * 0x01000193u;
    h ^= h >> 15;
  }
  return h | 1u;  // never 0: what a block of zeros holds
}
struct QrPinned {
  QrScalars scal;
  QrNodesOut tree;
  // batched growth: what the LAST control call of the enqueued sequence found, in ONE word
  // (early[0]: the call's sequence number << 16 | the steps the tree took so far << 1 | the tree
  // is incomplete, i.e. the guessed step count was too low).  The host settles a tree on this,
  // ~35 us before the leaf kernels and the score update behind it have run: the next iteration
  // is enqueued under them.
  int64_t early[4];
};

// k_lambda_u's roles (k_lambda.hip): workgroups [0, nC) take one query each, the eight waves
// together; then QR_LU_ROLES packed roles (a query per wave), largest capacity first
#define QR_LU_W 8
#define QR_LU_ROLES 3
struct QrLambdaPlanDev {
  uint32_t nC, nmaxC, kaccC, blocks;  // (blocks: the whole grid)
  struct {
    uint32_t blocks, per, count, first;  // workgroups, queries per workgroup, queries, their place in the list
    uint32_t nmax, kacc, slice;          // capacity (documents), top ranks kept, bytes of a working set
  } pk[QR_LU_ROLES];
};

struct qr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  int rank = 0, world = 1;
  int dmode = 0;                 // 0 = single GPU / feature-sharded, 1 = document-sharded
  uint64_t Nglobal = 0, Qglobal = 0;
  long long *d_xh = nullptr;     // doc-sharded histogram exchange: [cells] sums, [cells] counts, [2*world] f64 bits
  size_t xh_len = 0, xh_cells = 0;
  long long *d_xscal = nullptr;  // [world][4] f64 bits: maxabs, root_ss, root_sum, metric_sum
  long long *d_xleaf = nullptr;  // [world][2*nleaves] f64 bits: per-leaf (sum lambda, sum weight)
  long long *d_xlevel = nullptr;  // document-sharded level-wise growth: [node][feature][slot][sum, count] of a level
  size_t xlevel_cap = 0;
  // document-sharded batched growth: [QR_BATCH][feature][slot][sum, count] + [QR_BATCH][2 * world] f64 bits
  long long *d_xb = nullptr;
  size_t xb_len = 0;
  // k_decide_part (control step inside the partition launch) up to this many documents: every
  // workgroup of the launch runs the step on its own copy, which a launch of several rounds of
  // workgroups pays once per round (QR_FUSE_MAX_DOCS; measured, ms per iteration fused / not:
  // 1M 0.418 / 0.451, 2M 0.618 / 0.655, 4M 1.020 / 1.021, 8M 1.863 / 1.808)
  size_t fuse_max_docs = 4000000;
  QrTreeState *dtree_cur = nullptr;  // document-sharded batched growth: the copy of the tree state the last control step wrote
  bool dbatch_prepared = false;      // ... a control call of its own has made the next step's batch (its partition launch only applies it)
  bool dbatch_first = false;         // ... the next partition launch carries the ROOT's control step
  bool dbatch = false;            // the open tree grows by qr_tree_batch_* (document-sharded)
  // ... its last control call (qr_tree_batch_decide(last = 1)) has not been looked at yet; the tree
  // was ended like that (leaf kernels and score update enqueued behind a guess: they leave at once
  // if it was too low); a carried-on tree repeats the score update behind its leaf values
  bool dbatch_unsettled = false, dbatch_pending = false, dbatch_redo = false;
  size_t xleaf_cap = 0;
  int ncu = 256;
  // LDS the device really has (hipDeviceProp_t): a workgroup's opt-in maximum and a CU's total.  The
  // fast paths that were sized for gfx950's 160 KB are taken only when they fit (qr_api.hip,
  // k_wide.hip); everything else falls through to the general kernels.
  size_t lds_block = 160 * 1024, lds_cu = 160 * 1024;
  // training data
  size_t N = 0, F = 0, Q = 0, maxq = 0;
  float *d_raw = nullptr;        // [N][F] row-major f32
  float *d_labels = nullptr;
  uint32_t *d_qoff = nullptr;    // [Q+1]
  std::vector<float> h_labels;
  std::vector<uint64_t> h_qoff;
  // validation data
  size_t vN = 0, vQ = 0, vmaxq = 0;
  float *d_vraw = nullptr, *d_vlabels = nullptr;
  uint32_t *d_vqoff = nullptr;
  double *d_vscores = nullptr;
  std::vector<float> h_vlabels;
  std::vector<uint64_t> h_vqoff;
  // bins
  bool binned = false;
  int nblocks = 0;               // local blocks
  int flocal = 0;                // local real features
  std::vector<QrBlock> blocks;
  QrBlock *d_blocks = nullptr;
  int32_t *d_lf2gf = nullptr;    // local feature -> global feature
  int32_t *d_gf2lf = nullptr;    // global feature -> local feature or -1
  std::vector<int32_t> h_gf2lf, h_lf2gf;
  uint8_t *d_bins = nullptr;
  size_t bins_bytes = 0;
  uint8_t *d_bins_fm = nullptr;  // feature-major copy [flocal][N] for the partition
  // more than 255 thresholds per feature (k_wide.hip): ragged threshold rows (feature f
  // owns cells [woff[f], woff[f + 1]) of every per-node histogram), u32 bins
  bool wide = false;
  size_t wcells = 0;             // cells of one node histogram
  uint32_t wmax = 0;             // longest row
  uint32_t *d_woff = nullptr;    // [F + 1]
  // chunked scan of long rows (k_wide.hip): chunk table, per-chunk totals and bests
  uint32_t *d_wchunk = nullptr, *d_wchunk0 = nullptr, *d_wtot_c = nullptr;
  long long *d_wtot_s = nullptr, *d_wcbest = nullptr;
  size_t wchunks = 0;
  float *d_wthr = nullptr;       // [wcells]
  uint32_t *d_wbins = nullptr;   // [F][N] feature-major
  uint16_t *d_wbins16 = nullptr; // rows of up to 1152 slots: [group of 16 features][N][16] (k_whist16)
  uint64_t *d_wpart = nullptr;   // ... and its partial slots [document range][group][slots x 16] (k_wreduce16)
  size_t wpart_cap = 0;
  // k_exact.hip: rows too long for slot-indexed node histograms (--num-thresholds 0 on real-valued
  // columns) take the pre-sorted formulation -- per feature the documents sorted by slot, a node the
  // same segment of every feature's list
  bool xmode = false;
  unsigned long long *d_xroot = nullptr;     // [flocal][N] {slot << 32 | document}: the root's lists, never overwritten
  unsigned long long *d_xlist[2] = {nullptr, nullptr};  // the work lists the splits ping-pong between
  long long *d_xtot = nullptr;               // [2] fixed-point gradient totals of the node(s) being scanned
  uint8_t *d_xgoleft = nullptr;              // [N] go-left byte of the documents of the node being split
  unsigned long long *d_xpub = nullptr;      // the tiles' published words {epoch : 16, value : 48} (scan tiles, then partition tiles)
  long long *d_xnode_tot = nullptr, *d_xcs = nullptr;  // lazy split search: fixed-point gradient total per node; the winners' cumulative sums per feature
  long long *d_xgbest = nullptr;             // (= d_xtot + 2: best exact score seen per scan row)
  bool ens_depth_order = false;              // qr_ensemble_set_depth_order: trees walked and summed in ascending depth
  std::vector<uint32_t> ens_perm;            // ... position in the walk -> the model's tree (empty: the model's order)
  bool spec_exact = false;                   // the pending tree was enqueued by qr_k_exact_fit (tree_settle carries it on with qr_k_exact_continue)
  bool x_eager = false;                      // QR_X_EAGER: both children searched behind every split (the phase API's order)
  void *d_xtbest = nullptr;                  // [2][flocal][scan tiles] best candidate of every tile
  uint32_t xtiles_s = 0, xtiles_p = 0, xepoch = 0;
  std::vector<uint32_t> h_woff;
  std::vector<float> h_wthr;
  float *d_thr = nullptr;        // [F][256]
  uint32_t *d_thr_size = nullptr;
  std::vector<float> h_thr;
  std::vector<uint32_t> h_thr_size;
  // model state
  double *d_scores = nullptr, *d_lambda = nullptr, *d_weight = nullptr;
  // lambda / metric
  double *d_lg2 = nullptr;       // log2(r + 2), r < maxq(+valid)
  double *d_ilg2 = nullptr;      // 1.0 / log2(r + 2)
  size_t lg2_len = 0;
  double *d_idcg = nullptr, *d_vidcg = nullptr;
  int idcg_metric = -1, vidcg_metric = -1;
  size_t idcg_cutoff = (size_t)-1, vidcg_cutoff = (size_t)-1;
  double *d_qmetric = nullptr, *d_vqmetric = nullptr;
  uint32_t *d_ranks = nullptr, *d_vranks = nullptr;
  uint32_t *d_keys = nullptr;    // #strictly-greater scores per doc (k_rank)
  uint32_t *d_tied = nullptr;    // queue of queries with tied scores; [tied_cap] = count
  size_t tied_cap = 0, keys_cap = 0;
  // queries too long for the LDS-resident lambda kernel: flags / lists per set
  // (0 = training, 1 = validation) and the global scratch they run out of
  uint8_t *d_long_flag[2] = {nullptr, nullptr};
  uint32_t *d_long_list[2] = {nullptr, nullptr};
  std::vector<uint32_t> h_long_list[2];
  int long_tag[2] = {-1, -1};
  // ... and the others by size class (k_lambda keeps a query's working set in LDS, sized for
  // the longest query of ITS launch: one launch per class keeps the short queries -- the
  // many -- at full occupancy).  [which]: class lists concatenated on the device, per class
  // (first, count, longest query).
  uint32_t *d_qclass[2] = {nullptr, nullptr};
  struct QClass { uint32_t first, count, nmax; };
  std::vector<QClass> h_qclass[2];
  bool qclass_identity[2] = {false, false};  // one class holding every query in order
  // ... or, for a ragged set, ONE launch of three roles (k_lambda_u): the queries in dispatch order
  uint32_t *d_lu_list[2] = {nullptr, nullptr};
  QrLambdaPlanDev lu_plan[2];
  size_t lu_dyn[2] = {0, 0};
  bool lu_on[2] = {false, false};
  // Mart::update_modelscores left to the next lambda pass (k_tree.hip: qr_k_scores_update)
  int debug_lose_binning = 0;         // qr_debug_bins_clobber(which = 2): builds of the map that lose a few rows (test aid)
  int bins_rebuilt = 0;               // times qr_k_binning had to build the map again (verify-after-write; expected 0)
  bool no_lazy_scores = false;        // QR_LAZY_SCORES=0 at context creation: every score update is a launch of its own
  bool lazy_scores = false;
  double lazy_shrinkage = 0.0;
  size_t lu_kacc[2] = {0, 0};  // the top ranks the plan's slices were sized for
  size_t attr_lambda_u_lds = 64 * 1024;
  hipStream_t aux_stream[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t aux_fork = nullptr, aux_join[4] = {nullptr, nullptr, nullptr, nullptr};
  char *d_lscratch = nullptr;
  size_t lscratch_bytes = 0;
  double *d_ssq = nullptr;       // per-slice sum of squares partials
  double *d_qmax = nullptr;      // per-query / per-slice max |pseudo-response| (k_prep takes the maximum)
  size_t nqmax = 0;              // entries of d_qmax the last lambda / residual pass wrote
  QrScalars *d_scalars = nullptr;
  QrPinned *h_pin = nullptr;     // pinned host memory the kernels write the read-backs into
  QrPinned *d_pin = nullptr;     // the same memory through its device address
  // Read-backs: the kernel that finishes the scalars / the tree records writes them into the
  // pinned block and then a sequence number behind a system-scope fence; the host polls that
  // number (wait_seq_impl in qr_api.hip).  No event on the stream: an event record between two
  // kernels was ~6 us of idle GPU each, and waking the host from hipEventSynchronize ~25 us.
  double *d_prep_part = nullptr;  // k_prep: [16][4] workgroup partials + its ticket (word 64) + two slot sets (from word 72)
  // qr_prep.h: the lambda pass leaves the iteration's max |pseudo-response| in a slot set;
  // qr_lambda_compute DEFERS the launch that finishes the scalars: batched / level-wise growth
  // lets its workgroups ride in the tree's root scan launch (one launch less per iteration),
  // everything else that needs the scalars finishes them first (qr_k_prep_flush).
  int prep_parity = 0;            // slot set of the last lambda pass
  bool prep_deferred = false;     // the scalars of the last lambda pass are still to be finished
  bool no_defer = false;          // QR_NO_DEFER_PREP=1: always a launch of its own (A/B, debugging)
  QrHistWg *d_root_wg = nullptr;  // the root histogram launch's per-workgroup shares (k_tree.hip: root_shares)
  QrScanWg *d_root_scan = nullptr;  // ... and the root scan launch's per-feature shares
  int root_scan_n = 0;
  uint32_t root_wg_n = 0;
  int root_wg_g = 0, root_wg_buf = -1;
  bool root_wg_valid = false;      // the cached shares match (root_wg_n, root_wg_g, root_wg_buf, root_wg_gen)
  uint64_t blocks_gen = 0;         // counts the rebuilds of c->blocks (bins_finish): part of the cache key
  uint64_t root_wg_gen = 0;
  int obl_reset_nodes = 0;        // level-wise growth: node records the root scan launch's last workgroup resets (0: none)
  bool obl_own_launches = false;  // QR_OBL_OWN_LAUNCHES=1: the tree-state reset and k_finish as launches of their own (A/B, debugging)
  size_t prep_nss = 0;
  int prep_with_metric = 0, prep_publish = 0;
  int32_t scal_seq = 0;   // of the last launch that publishes the scalars
  int64_t nodes_seq = 0;  // of the last launch that publishes tree records
  unsigned long long readback_retries = 0;  // records that did not fit their sequence number at first sight
  int64_t early_seq = 0;  // of the last final control call (QrPinned::early)
  bool scal_pending = false, nodes_pending = false;
  uint64_t debug_root_short = 0;  // (test aid)
  uint64_t nodes_grown_on = 0;   // the root's document count the pending tree's records must show (qr_tree_nodes)
  size_t cur_maxnodes = 0;
  // tree
  uint32_t *d_order[2] = {nullptr, nullptr};
  uint64_t *d_partials = nullptr;
  long long *d_red_sum = nullptr;  // partials reduced over workgroups, [block][bin][fw]
  uint32_t *d_red_cnt = nullptr;
  size_t partial_slots = 0;
  long long *d_hsum = nullptr;   // [QR_MAXNODES..][flocal][256] cumulative fixed-point
  uint32_t *d_hcnt = nullptr;
  size_t hist_slots = 0;
  qr_split_t *d_featrec = nullptr;  // [2][flocal] per-feature best of the 2 nodes just scanned
  qr_split_t *d_recs_local = nullptr;  // [2]
  qr_split_t *d_recs_all = nullptr;    // [world][2]
  uint32_t *d_mask = nullptr;
  size_t mask_words = 0;
  float *d_featthr = nullptr;         // threshold value of every per-feature best record
  uint64_t cur_minls = 1;             // min leaf support of the tree being fitted (batched growth)
  double *d_lpart_ss = nullptr;       // batched growth: child sums per partition workgroup
  double *d_lpart_ss2 = nullptr;      // ... second copy (k_decide_part reads one, writes the other)
  double *d_jobsum = nullptr;         // [QR_BATCH][ss, sum] of the batch's directly built children (k_redscan adds the partials up)
  QrTreeState *d_tree2 = nullptr;     // ... second copy of the tree state (same reason)
  unsigned long long *d_bpart_state = nullptr;  // ... look-back granules of k_decide_part
  uint32_t bepoch = 0;                // ... their epoch, counted on the host
  QrHistWg *d_lhist_wg = nullptr;     // ... per-workgroup shares of the step's launches
  QrPartWg *d_lpart_wg = nullptr;
  QrScanWg *d_lscan_wg = nullptr;     // ... [QR_BATCH][flocal]
  QrPlan *d_lplan = nullptr;          // ... and the plan of every node of the batch
  bool no_batch = false;              // QR_NO_BATCH=1: one split per step (debugging aid)
  int exact_tail = 0;                 // QR_EXACT_TAIL=1: the std::sort emulation orders the ranks beyond the cutoff too
  // guessed step count of batched growth: steps to enqueue for the next tree (0 = the worst
  // case nleaves - 1), whether the tree just enqueued may turn out incomplete, and what the
  // continuation has to repeat (leaf kernels with `newton`, the score update with `shrinkage`)
  size_t steps_hint = 0;
  long steps_force = -1;              // QR_STEPS_HINT=k: always enqueue k steps (tests the continuation)
  bool spec_debug = false;            // QR_SPEC_DEBUG=1: diagnostic lines on stderr
  bool spec_pending = false, spec_scores_enqueued = false;
  int spec_newton = 0;
  double spec_shrinkage = 0.0;
  uint64_t spec_trees = 0, spec_misses = 0;
  uint32_t *d_red_cnt_loc = nullptr;  // document-sharded: the rank's own reduced counts ...
  uint32_t *d_hcnt_loc = nullptr;     // ... and their prefix per (node slot, feature, threshold slot)
  unsigned long long *d_part_state = nullptr;  // look-back granules {epoch, count}
  double *d_part_ss = nullptr;
  QrTreeState *d_tree = nullptr;
  // level-wise growth: workgroup -> (node, local index) maps, per-node reduced arrays
  uint32_t *d_lhist_map = nullptr, *d_lpart_map = nullptr;
  size_t lhist_cap = 0, lpart_cap = 0, lslots_cap = 0, lred_nodes = 0;
  uint64_t *d_lpartials = nullptr;
  double *d_lhistsum = nullptr;       // batched growth: [slot][ss, sum] of the child a histogram workgroup (block 0) read
  unsigned long long *d_lpart_state = nullptr;
  double *d_leafpart = nullptr;  // [slices][2] partial sums (k_leaf_sums), or [slices][16][2] (k_leaf_sums_doc)
  uint8_t *d_leafb = nullptr;    // [N] leaf of every document in the last small tree (k_leaf_sums_doc)
  size_t leaf_cap = 0;           // leaves the tree under construction can have (<= 16: the document-order leaf kernels)
  bool leafb_valid = false;      // d_leafb covers every document of the finished tree
  bool leaf_by_position = false; // QR_LEAF_BY_POSITION=1: always k_leaf_sums (tests compare the two)
  bool tree_valid = false;
  bool tree_open = false;
  int tree_step = 0;             // decides issued for the open tree
  // --subsample: documents per iteration (0 = all), the sampler's state and scratch
  size_t sub_k = 0;              // documents per sample (over ALL ranks' documents on a document-sharded context)
  size_t sub_n = 0;              // ... of which this context's own (== sub_k unless document-sharded)
  size_t sub_first = 0;          // document-sharded: global index of this rank's first document
  uint64_t sub_seed = 0, sub_iter = 0;
  uint8_t *d_present = nullptr;
  void *d_sample_work = nullptr;      // the select's histograms and state, the scatter's offsets (k_sample.hip)
  uint32_t sub_key_mask = 0xFFFFFFFFu; // (tests narrow the keys to meet equal ones: qr_debug_sample_key_mask)
  uint32_t *d_sample_count = nullptr;
  uint32_t mf_k = 0;             // --max-features: features a node's split search sees (0 = all)
  uint64_t mf_seed = 0, tree_counter = 0;
  size_t cur_nleaves = 0;
  size_t cur_depth = 0;          // of the open feature-sharded oblivious tree
  // ensemble
  qr_node_t *d_ens = nullptr;
  double *d_ens_w = nullptr;
  size_t ens_trees = 0, ens_maxnodes = 0;
  int ens_maxf = -1;             // largest feature index the ensemble tests
  // compact binned form of the ensemble (k_score_bin)
  bool sb_ready = false, sb_u8 = false;
  size_t sb_F = 0, sb_NI = 0, sb_NL = 0, sb_tmax = 0, sb_bins_bytes = 0;
  bool sb_self = false;          // the node array holds the leaves too (self-looping), k_score.hip
  // 4-byte node records (k_score_p4): u8 bins, trees of <= 255 nodes, row offsets < 65536;
  // NNP = 128 or 256 entries per tree
  bool p4_ready = false;
  size_t p4_NNP = 0;
  // per batch of 16 trees its LDS image: [16][NNP] records {row offset : 16, 255 - slot : 8, left
  // child : 8} in level order, then [16][NNP] f64 leaf value * tree weight at the leaf's position
  uint32_t *d_p4_batches = nullptr;
  uint8_t *d_p4_depth = nullptr;    // per group of 8 trees: levels of its deepest tree
  void *d_sb_nodes = nullptr, *d_sb_bins = nullptr;
  double *d_sb_leaves = nullptr;
  uint16_t *d_sb_root = nullptr;
  float *d_sb_thr = nullptr;
  uint32_t *d_sb_thr_cnt = nullptr;
  // oblivious ensemble (generate_oblivious.cc layout)
  uint32_t *d_obl_feat = nullptr, *d_obl_depths = nullptr;
  float *d_obl_thr = nullptr, *d_obl_w = nullptr;
  double *d_obl_leaves = nullptr;
  size_t obl_trees = 0, obl_depth = 0;
  uint32_t obl_maxf = 0;
  // binned form of the oblivious ensemble (k_obl_score_bin)
  bool ob_ready = false, ob_u8 = false;
  size_t ob_F = 0, ob_tmax = 0;
  uint32_t *d_ob_fk = nullptr;      // [trees][depth]: feature | threshold index << 16
  // k_obl_score_s (u8 bins, depth <= 8): per tree {row offset[8], slot[8]} for the scalar unit,
  // leaf * weight per tree; both padded to whole batches of obs_tb trees
  bool obs_ready = false;
  size_t obs_tb = 0, obs_tpad = 0, obs_nw = 0;
  uint32_t *d_obs_trees = nullptr;
  double *d_obs_leaves = nullptr;
  float *d_ob_thr = nullptr;        // [ob_F][ob_tmax] sorted distinct thresholds per feature
  uint32_t *d_ob_thr_cnt = nullptr;
  // largest dynamic-LDS size announced to the runtime per kernel family (a function
  // attribute is a per-device setting: one process may drive several devices)
  size_t attr_hist_lds = 0, attr_lambda_lds = 64 * 1024, attr_whist_lds = 0;
  // profiling
  bool prof_on = false;
  bool batch_root = false;        // the root launches of a batched tree (no reset launch ran: minls travels as an argument)
  bool finish_in_decide = false;  // the tree's last control call numbered its leaves (no k_finish launch)
  unsigned prof_stride = 1, prof_tick = 0;  // events on every prof_stride-th root launch
  bool prof_child = false;       // also time the child-histogram launches (qr_prof_enable(ctx, 2 | 1))
  bool prof_lambda = false;      // ... or the lambda pass's launch instead (qr_prof_enable(ctx, 4 | 1)): same slot
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events_child;
  uint64_t prof_launches = 0, prof_launches_child = 0;
  double prof_ms = 0.0, prof_ms_child = 0.0;
  double prof_bytes = 0.0;
};

#define QR_CHECK(ctx, expr)                                                  \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);        \
      return QR_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)

// A device-to-host copy into memory of the CALLER's frame (a stack word, a local vector) behind the
// context's stream: never returns while the copy may still be in flight -- an error between the
// request and the wait would otherwise leave a DMA writing a frame that is gone.
#define QR_D2H(ctx, dst, src, bytes)                                                         \
  do {                                                                                        \
    hipError_t _e = hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, (ctx)->stream); \
    const hipError_t _w = hipStreamSynchronize((ctx)->stream);                                \
    if (_e == hipSuccess) _e = _w;                                                            \
    if (_e != hipSuccess) {                                                                   \
      (void)hipDeviceSynchronize();                                                           \
      (ctx)->err = std::string("device-to-host copy of " #src ": ") + hipGetErrorString(_e);  \
      return QR_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

#define QR_FAIL(ctx, code, msg) \
  do {                          \
    (ctx)->err = (msg);         \
    return (code);              \
  } while (0)

// kernel launchers implemented in the .hip files -----------------------------
int qr_k_transpose(qr_ctx *c, const float *raw, float *col, size_t N, size_t F);
int qr_k_colstats(qr_ctx *c, const float *col, size_t N, size_t F, uint32_t limit,
                  uint32_t *d_vals, uint32_t *d_cnt, uint32_t *d_minmax);
int qr_k_binning(qr_ctx *c);
int qr_k_bins_verify(qr_ctx *c, unsigned long long *bad_rows, unsigned long long *bad_fm, unsigned long long *by_wg = nullptr);
int qr_k_wide_thresholds(qr_ctx *c, const float *d_col, size_t nthresholds);
int qr_k_wide_binning(qr_ctx *c, const float *d_col);
bool qr_k_wide_fast_rows(const qr_ctx *c, size_t max_slots);
struct QrTreeState;
bool qr_k_wide_batch_ok(const qr_ctx *c);
int qr_k_whist_scan_batch(qr_ctx *c, const QrTreeState *ts, const double *pss);
int qr_k_whist_scan(qr_ctx *c, int root_mode);
int qr_k_scores_flush(qr_ctx *c);
int qr_k_debug_check(qr_ctx *c);
int qr_k_exact_build(qr_ctx *c);
int qr_k_wide_stats(qr_ctx *c, const float *d_col, size_t limit, uint32_t *vals, uint32_t *cnt, uint32_t *mm);
int qr_k_wscan_doc(qr_ctx *c, int root_mode);   // document-sharded wide bins: all-reduced cells -> slot, scan
int qr_k_exact_scan(qr_ctx *c, int root_mode);
int qr_k_exact_fit(qr_ctx *c, size_t nleaves, uint64_t minls);        // lazy split search: the whole tree enqueued
int qr_k_exact_continue(qr_ctx *c, size_t steps);                      // ... carried on (tree_settle)
// the two control launches of a lazy step (k_tree.hip, next to the helpers they share with k_decide)
int qr_k_xpop(qr_ctx *c, int root_mode, size_t nleaves, uint64_t minls, int final_call);
int qr_k_xapply(qr_ctx *c, int root_mode);
int qr_k_xpartition(qr_ctx *c);
void qr_k_exact_free(qr_ctx *c);
#define QR_X_MIN_SLOTS 16384u  /* longest row from which a wide context takes the pre-sorted path (k_exact.hip) */
inline bool qr_exact_active(const qr_ctx *c) { return c->xmode && !c->sub_k; }
#define QR_WCHUNK 8192u  /* slots per workgroup of the chunked scan of long rows (k_wide.hip) */
int qr_k_wobl_fill(qr_ctx *c, int level);
int qr_k_wobl_hist(qr_ctx *c, int nodes);
int qr_k_lambda(qr_ctx *c, int which, int metric, size_t cutoff, int mode);
int qr_k_residual(qr_ctx *c);
int qr_k_prep(qr_ctx *c, size_t nslices, int with_metric, int publish = 0);
struct QrPrepJob;
void qr_k_prep_job(qr_ctx *c, size_t nss, int with_metric, int publish, QrPrepJob *j);
int qr_k_prep_flush(qr_ctx *c);
#define QR_PREP_WORDS (72 + 2 * 64)  /* d_prep_part: partials, ticket, two slot sets of 64 words */
inline unsigned long long *qr_prep_slots(qr_ctx *c, int parity) {
  return reinterpret_cast<unsigned long long *>(c->d_prep_part + 72 + 64 * (parity & 1));
}
inline int qr_next_scal_seq(qr_ctx *c) {  // never 0: the value the block starts with
  c->scal_seq = c->scal_seq == 0x7fffffff ? 1 : c->scal_seq + 1;
  return c->scal_seq;
}
int qr_k_prep_pack(qr_ctx *c);
int qr_k_prep_global(qr_ctx *c);
int qr_k_tree_leaves_global(qr_ctx *c, int newton);
size_t qr_k_sample_work_bytes(size_t Nloc);
int qr_k_sample_draw(qr_ctx *c);
int qr_k_sample_sums(qr_ctx *c);
int qr_k_metric_reduce(qr_ctx *c, int which);
int qr_k_tree_begin(qr_ctx *c, size_t nleaves, uint64_t minls);
int qr_k_tree_decide(qr_ctx *c);
int qr_k_tree_apply(qr_ctx *c);
int qr_k_tree_finish(qr_ctx *c, int newton);
int qr_k_scores_update(qr_ctx *c, double shrinkage, bool repeat = false);
int qr_k_oblivious_fit(qr_ctx *c, size_t depth, uint64_t minls);
int qr_k_obl_begin(qr_ctx *c, size_t depth, uint64_t minls);
int qr_k_obl_propose(qr_ctx *c, int level);
int qr_k_obl_mark(qr_ctx *c, int level);
int qr_k_obl_apply(qr_ctx *c, int level, int last);
int qr_k_tree_fit_batch(qr_ctx *c, size_t nleaves, uint64_t minls);
int qr_k_tree_continue(qr_ctx *c, size_t nleaves, uint64_t minls, size_t steps_done, size_t max_steps = 0);
// (document-sharded ranks: the same growth phase by phase, the host's all-reduces in between)
int qr_k_dbatch_root_hist(qr_ctx *c, size_t nleaves, uint64_t minls);
int qr_k_dbatch_root_decide(qr_ctx *c, size_t nleaves, uint64_t minls);
int qr_k_dbatch_apply(qr_ctx *c, size_t nleaves);
int qr_k_dbatch_decide(qr_ctx *c, size_t nleaves, uint64_t minls, int final_call);
int qr_k_ensemble_score(qr_ctx *c, const float *d_x, size_t N, size_t F,
                        double *d_out, double *d_partial = nullptr, int ignore_weights = 0);
int qr_k_obl_score(qr_ctx *c, const float *d_x, size_t N, size_t F, double *d_out);
int qr_k_ensemble_score_fast(qr_ctx *c, const float *d_x, size_t N, size_t F, double *d_out);
int qr_k_obl_score_fast(qr_ctx *c, const float *d_x, size_t N, size_t F, double *d_out);
