// mart_multi.cc -- Mart::learn on the GPUs of one node: `quicklearn --gpus N`.
//
// One process, one host thread per GPU; every thread drives its own device context
// (include/qr_hip.h) on its own HIP stream and calls RCCL (ncclAllReduce /
// ncclAllGather over xGMI) on that same stream between the phase calls, so no
// collective ever waits on the host.  Two layouts:
//
//   documents  rank r holds the queries [r Q/N, (r+1) Q/N) and every feature of them.
//              What is exchanged is the node histogram itself -- exact fixed-point
//              integers, so ONE int64 sum all-reduce per histogram gives every rank the
//              bits one GPU would have accumulated over all documents; scan, gains, heap
//              and partition then run redundantly / locally and agree without another
//              exchange.  Thresholds come from the whole set (column statistics gathered
//              over the ranks, mart.cc:147-169 over their union).  The default.
//   features   north_star's layout: every rank holds all documents and the feature
//              columns [r F/N, (r+1) F/N) of the bin matrix; per split an all-gather of
//              the ranks' best-split records and a sum all-reduce of the go-left bit mask
//              (only the owner of the winning feature contributes non-zero words).
//
// Oblivious trees (one all-gather of the ranks' level records + one all-reduce of the
// go-left bits by document per level) and --subsample (every rank draws the same sample)
// run in the feature layout, where every rank holds every document.
// The boosting loop is Mart::learn's (mart.cc:307-395): validation, early stop, rollback
// and --partial saves included.  Rank 0 keeps the ensemble; every other rank checks that
// it built the same tree and the run stops if one did not.
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <limits>
#include <mutex>
#include <thread>

#include "mart.h"

namespace quickrank {
namespace learning {
namespace forests {

namespace {

class Barrier {  // reusable; std::barrier is C++20
 public:
  explicit Barrier(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m_);
    const size_t gen = gen_;
    if (++count_ == n_) {
      count_ = 0;
      ++gen_;
      cv_.notify_all();
    } else {
      cv_.wait(lk, [&] { return gen != gen_; });
    }
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_, count_ = 0;
  size_t gen_ = 0;
};

// A fatal error on one rank's thread while its siblings may be blocked inside an RCCL
// collective or at the barrier: exit() would run the static and HIP / RCCL destructors next
// to those in-flight collectives and can hang instead of terminating, so the process leaves
// at once (streams flushed by hand, no destructors).
[[noreturn]] void fatal_exit() {
  std::cout.flush();
  std::cerr.flush();
  std::_Exit(EXIT_FAILURE);
}
[[noreturn]] void die(qr_ctx *c, const char *what) {
  std::cerr << "!!! " << what << ": " << qr_last_error(c) << std::endl;
  fatal_exit();
}
#define QRM(c, call)                           \
  do {                                         \
    if ((call) != QR_OK) die(c, #call);        \
  } while (0)
#define NCCL(call)                                                                     \
  do {                                                                                 \
    const ncclResult_t r_ = (call);                                                    \
    if (r_ != ncclSuccess) {                                                           \
      std::cerr << "!!! " #call ": " << ncclGetErrorString(r_) << std::endl;           \
      fatal_exit();                                                                    \
    }                                                                                  \
  } while (0)

struct Shared {
  int world;
  Barrier bar;
  // thresholds of the whole set (document layout): per-rank column statistics
  std::vector<uint32_t> vals, cnt, mm;
  // per-rank (sum of per-query metrics, queries) of the training / validation set
  std::vector<double> msum[2], mq[2];
  // rank 0's tree of the iteration, for the cross-rank check
  std::vector<qr_node_t> nodes0;
  size_t nn0 = 0;
  // document layout, --num-thresholds 0 (or above 255) on columns whose slots a document-sharded
  // node histogram cannot hold: every rank finds the same numbers, leaves before any training, and
  // the run starts over in the feature layout (learn_multi)
  bool relayout = false;
  explicit Shared(int w) : world(w), bar(w) {
    for (int k = 0; k < 2; ++k) {
      msum[k].assign(w, 0.0);
      mq[k].assign(w, 0.0);
    }
  }
};

int metric_code_of(const std::string &m) {
  if (m == "NDCG") return QR_METRIC_NDCG;
  if (m == "DCG") return QR_METRIC_DCG;
  std::cerr << " !! Train Metric was not set properly" << std::endl;  // driver.cc:114-117
  exit(EXIT_FAILURE);
}

// whole queries [q0, q1) of rank r: balanced boundaries Q r / w, so that no rank is left
// without queries while w <= Q (ceil(Q / w)-sized slices would give rank 3 of 4 nothing at Q = 9)
void query_slice(size_t Q, int r, int w, size_t *q0, size_t *q1) {
  *q0 = Q * (size_t)r / (size_t)w;
  *q1 = Q * (size_t)(r + 1) / (size_t)w;
}

}  // namespace

// (for the CPU tests: the slices of a world of w ranks, without a GPU)
void multi_query_slice(size_t Q, int r, int w, size_t *q0, size_t *q1) { query_slice(Q, r, w, q0, q1); }

void Mart::learn_multi(std::shared_ptr<data::Dataset> training, std::shared_ptr<data::Dataset> validation,
                       const std::string &metric, size_t cutoff, size_t partial_save,
                       const std::string &output_basename, int ngpus, bool feature_sharded) {
  const bool obliv = algo_ >= OBVMART;
  // More than 255 thresholds per feature (--num-thresholds 0 = every distinct value, the
  // reference's default, or any value above 255): feature-block sharding grows leaf-wise trees on
  // them (every rank holds the whole rows of its own features; round 3).  A document-sharded
  // histogram of every distinct value would be an all-reduce of 10^7 - 10^8 cells per node, and the
  // level-wise phase calls use u8 bins: both refused.
  // (round 4: document shards take them too -- the thresholds of the whole set from every rank's
  // column statistics, qr_bins_build_wide_with -- while a node histogram stays within 4M cells)
  const bool many = nthresholds_ > 255 || nthresholds_ == 0;
  if ((max_features_ != 1.0f && obliv) || (many && obliv)) {
    if (many)
      std::cerr << "!!! --gpus > 1 with --num-thresholds 0 or above 255 needs MART / LAMBDAMART (sharded oblivious "
                   "trees take --num-thresholds in [1, 255])."
                << std::endl;
    else
      std::cerr << "!!! --max-features applies to MART / LAMBDAMART." << std::endl;
    exit(EXIT_FAILURE);
  }
  const int W = ngpus;
  // (the feature layout holds every feature range only while W - 1 ranges leave one for the last rank)
  const bool no_feature_layout =
      ((training->num_features() + (size_t)W - 1) / (size_t)W) * (size_t)(W - 1) >= training->num_features();
  const int mcode = metric_code_of(metric);
  const bool lambda = algo_ == LAMBDAMART || algo_ == OBVLAMBDAMART;
  const size_t N = training->num_instances(), F = training->num_features(), Q = training->num_queries();
  // (feature ranges are ceil(F / W) wide, qr_api.hip: the last rank must still own a feature)
  if ((size_t)W > Q || (feature_sharded && ((F + (size_t)W - 1) / (size_t)W) * (size_t)(W - 1) >= F)) {
    std::cerr << "!!! more GPUs than " << (feature_sharded ? "feature ranges" : "queries") << std::endl;
    exit(EXIT_FAILURE);
  }
  std::cout << "# Initialization";
  std::cout.flush();
  auto t_init0 = std::chrono::high_resolution_clock::now();
  std::vector<ncclComm_t> comms(W);
  std::vector<int> devs(W);
  for (int r = 0; r < W; ++r) devs[r] = r;
  NCCL(ncclCommInitAll(comms.data(), W, devs.data()));
  Shared sh(W);
  // (qr_bins_stats: nthresholds + 1 distinct values per column, 256 of them for --num-thresholds 0)
  const uint32_t limit = nthresholds_ ? (uint32_t)std::min<size_t>(nthresholds_, 255) + 1 : 256u;
  if (!feature_sharded) {
    sh.vals.assign((size_t)W * F * (limit + 1), 0);
    sh.cnt.assign((size_t)W * F, 0);
    sh.mm.assign((size_t)W * 2 * F, 0);
  }
  best_metric_on_validation_ = std::numeric_limits<double>::lowest();
  best_metric_on_training_ = std::numeric_limits<double>::lowest();
  best_model_ = 0;
  ensemble_model_.set_capacity(ntrees_);
  const size_t maxnodes = obliv ? ((size_t)1 << (treedepth_ + 1)) : 2 * nleaves_ + 1;
  sh.nodes0.resize(maxnodes);
  unsigned long long sample_seed = sampling_seed_;
  if ((subsample_ != 1.0f || max_features_ != 1.0f) && sample_seed == 0)  // the reference seeds from the clock at every draw
    sample_seed = (unsigned long long)std::chrono::system_clock::now().time_since_epoch().count();
  std::chrono::high_resolution_clock::time_point t_train0;

  // restart from a previously saved model (mart.cc:237-253): its scores of the training and
  // validation documents, computed once; every rank starts from its own part of them
  const size_t first = ensemble_model_.get_size();
  std::vector<Score> rs_train, rs_valid;
  if (first) {
    rs_train.resize(N);
    score_dataset(*training, rs_train.data());
    if (validation) {
      rs_valid.resize(validation->num_instances());
      score_dataset(*validation, rs_valid.data());
    }
  }

  auto worker = [&](const int r) {
    qr_ctx *c = nullptr;
    if (qr_ctx_create(r, &c) != QR_OK) die(nullptr, "qr_ctx_create");
    void *sv = nullptr;
    QRM(c, qr_ctx_stream(c, &sv));
    hipStream_t stream = (hipStream_t)sv;
    ncclComm_t comm = comms[r];
    int nranks = 0;
    NCCL(ncclCommCount(comm, &nranks));
    // ---- data
    size_t tq0 = 0, tq1 = Q, vq0 = 0, vq1 = validation ? validation->num_queries() : 0;
    if (feature_sharded) {
      QRM(c, qr_ctx_set_shard(c, r, W));
    } else {
      query_slice(Q, r, W, &tq0, &tq1);
      if (validation) query_slice(validation->num_queries(), r, W, &vq0, &vq1);
      QRM(c, qr_ctx_set_doc_shard(c, r, W, N, Q));
    }
    auto upload = [&](const data::Dataset &ds, size_t q0, size_t q1, bool valid) {
      const size_t d0 = ds.offset(q0), d1 = ds.offset(q1);
      std::vector<uint64_t> qo(q1 - q0 + 1);
      for (size_t q = q0; q <= q1; ++q) qo[q - q0] = ds.offset(q) - d0;
      if (d1 == d0) return;  // (a rank without validation queries: nothing to upload)
      if (valid)
        QRM(c, qr_valid_upload(c, ds.at(d0, 0), d1 - d0, ds.num_features(), ds.labels() + d0, qo.data(),
                               q1 - q0));
      else
        QRM(c, qr_dataset_upload(c, ds.at(d0, 0), d1 - d0, ds.num_features(), ds.labels() + d0, qo.data(),
                                 q1 - q0));
    };
    upload(*training, tq0, tq1, false);
    const bool has_valid = validation && vq1 > vq0;
    if (has_valid) upload(*validation, vq0, vq1, true);
    // ---- Mart::init: thresholds + bin map
    if (feature_sharded) {
      // (u8 bins when every feature of the rank has at most 255 thresholds, else the wide path:
      // every rank decides for its own features -- the kernels read a flag, the protocol is the same)
      const int brc = qr_bins_build(c, nthresholds_, nullptr, nullptr);
      if (brc == QR_ERR_UNSUPPORTED)
        QRM(c, qr_bins_build_wide(c, nthresholds_, nullptr, nullptr));
      else if (brc != QR_OK)
        die(c, "qr_bins_build");
    } else {
      // up to 255 thresholds per feature: u8 bins; more (or --num-thresholds 0 on a column with more
      // distinct values): ragged rows.  Every rank computes the same thresholds from everybody's
      // statistics, so every rank takes the same branch.
      bool wide = nthresholds_ > 255;
      if (!wide) {
        QRM(c, qr_bins_stats(c, nthresholds_, &sh.vals[(size_t)r * F * (limit + 1)], &sh.cnt[(size_t)r * F],
                             &sh.mm[(size_t)r * 2 * F]));
        sh.bar.wait();
        std::vector<float> thr(F * QR_MAX_BINS);
        std::vector<uint32_t> ts(F);
        const int trc = qr_thresholds_from_stats(F, nthresholds_, W, sh.vals.data(), sh.cnt.data(), sh.mm.data(),
                                                 thr.data(), ts.data());
        if (trc == QR_OK)
          QRM(c, qr_bins_build_with(c, thr.data(), ts.data()));
        else if (trc == QR_ERR_UNSUPPORTED && nthresholds_ == 0)
          wide = true;   // a column with more than 255 distinct values
        else
          die(c, "qr_thresholds_from_stats");
        sh.bar.wait();   // (everybody has read the statistics before the wide ones overwrite them)
      }
      if (wide) {
        const size_t wl = nthresholds_ ? nthresholds_ + 1 : (size_t)65536;
        if (r == 0) {
          sh.vals.assign((size_t)W * F * wl, 0);
          sh.cnt.assign((size_t)W * F, 0);
          sh.mm.assign((size_t)W * 2 * F, 0);
        }
        sh.bar.wait();
        QRM(c, qr_bins_stats_wide(c, wl, &sh.vals[(size_t)r * F * wl], &sh.cnt[(size_t)r * F],
                                  &sh.mm[(size_t)r * 2 * F]));
        sh.bar.wait();
        std::vector<uint32_t> ts(F);
        size_t cells = 0;
        const int wrc = qr_thresholds_from_stats_wide(F, nthresholds_, W, wl, sh.vals.data(), sh.cnt.data(),
                                                      sh.mm.data(), nullptr, 0, ts.data(), &cells);
        // A document-sharded node histogram is all-reduced cell by cell: its rows hold at most 4M slots
        // in all (qr_bins_build_wide_with refuses more), and the statistics carry at most 65536 distinct
        // values per column.  Beyond that -- the reference's DEFAULT flag on real-valued columns -- the
        // best split of a node is a function of the prefix sums over ALL documents in slot order, which
        // shards by FEATURE, not by document: the run starts over in the feature layout, where every
        // rank grows on the pre-sorted lists of its own columns (k_exact.hip) and the exchange per
        // split is 32-byte records and a bit mask.  Same model either way; every rank sees the same
        // numbers here, so every rank takes this exit.
        if ((wrc == QR_ERR_UNSUPPORTED && nthresholds_ == 0) || (wrc == QR_OK && cells > ((size_t)4 << 20))) {
          if (no_feature_layout) {
            if (r == 0)
              std::cerr << "!!! --shard docs: more threshold slots than a document-sharded node histogram can "
                           "hold (use fewer --num-thresholds or --shard features)." << std::endl;
            fatal_exit();
          }
          sh.bar.wait();
          if (r == 0) sh.relayout = true;
          qr_ctx_destroy(c);
          return;
        } else if (wrc != QR_OK)
          die(c, "qr_thresholds_from_stats_wide");
        std::vector<float> thr(cells);
        if (qr_thresholds_from_stats_wide(F, nthresholds_, W, wl, sh.vals.data(), sh.cnt.data(), sh.mm.data(),
                                          thr.data(), cells, ts.data(), &cells) != QR_OK)
          die(c, "qr_thresholds_from_stats_wide");
        QRM(c, qr_bins_build_wide_with(c, thr.data(), ts.data(), nullptr, nullptr));
        sh.bar.wait();   // (everybody has merged the statistics)
        if (r == 0) {    // W * F * 65536 u32 of column statistics: not kept for the rest of the training
          std::vector<uint32_t>().swap(sh.vals);
          std::vector<uint32_t>().swap(sh.cnt);
          std::vector<uint32_t>().swap(sh.mm);
        }
      }
    }
    QRM(c, qr_scores_reset(c));
    if (first) {
      QRM(c, qr_scores_set(c, rs_train.data() + (feature_sharded ? 0 : training->offset(tq0))));
      if (has_valid) QRM(c, qr_valid_scores_set(c, rs_valid.data() + (feature_sharded ? 0 : validation->offset(vq0))));
    }
    // --max-features (rt.cc:222-243): a node's subset is a function of (seed, tree, node, feature):
    // the same on every rank
    if (max_features_ != 1.0f) QRM(c, qr_tree_set_max_features(c, max_features_, sample_seed));
    // the same draw on every rank: feature-sharded ranks hold every document; document-sharded
    // ones draw from the keys of all ranks' documents and keep their own part
    if (subsample_ != 1.0f) {
      if (feature_sharded)
        QRM(c, qr_subsample_set(c, subsample_, sample_seed));
      else
        QRM(c, qr_subsample_set_doc(c, subsample_, sample_seed, training->offset(tq0)));
    }
    void *x_hist = nullptr, *x_scal = nullptr, *x_leaf = nullptr, *recs_local = nullptr, *recs_all = nullptr,
         *mask = nullptr;
    size_t n_hist = 0, n_scal = 0, n_leaf = 0, rec_bytes = 0, mask_bytes = 0;
    if (feature_sharded && obliv)
      QRM(c, qr_obl_exchange_buffers(c, &recs_local, &recs_all, &rec_bytes, &mask, &mask_bytes));
    else if (feature_sharded)
      QRM(c, qr_exchange_buffers(c, &recs_local, &recs_all, &rec_bytes, &mask, &mask_bytes));
    else
      QRM(c, qr_doc_exchange_buffers(c, &x_hist, &n_hist, &x_scal, &n_scal, nullptr, nullptr));
    // what this rank hands to RCCL while it grows trees: calls and payload bytes (an all-gather counted by what
    // every rank receives), reported per tree behind the training table
    size_t n_coll = 0, coll_bytes = 0;
    auto sum64 = [&](void *p, size_t n) {
      ++n_coll;
      coll_bytes += n * 8;
      NCCL(ncclAllReduce(p, p, n, ncclInt64, ncclSum, comm, stream));
    };
    auto gather_recs = [&]() {
      ++n_coll;
      coll_bytes += rec_bytes * (size_t)nranks;
      NCCL(ncclAllGather(recs_local, recs_all, rec_bytes, ncclInt8, comm, stream));
    };
    auto sum_mask = [&]() {
      ++n_coll;
      coll_bytes += mask_bytes;
      NCCL(ncclAllReduce(mask, mask, mask_bytes / 4, ncclInt32, ncclSum, comm, stream));
    };
    // (QR_DOC_BATCH=0: one split per exchange, the protocol of rounds 1-3)
    const bool doc_batch = !(getenv("QR_DOC_BATCH") && atoi(getenv("QR_DOC_BATCH")) == 0);
    sh.bar.wait();
    if (r == 0) {
      auto t_init1 = std::chrono::high_resolution_clock::now();
      std::cout << ": " << std::setprecision(2) << std::chrono::duration<double>(t_init1 - t_init0).count()
                << " s." << std::endl;
      std::cout << "# " << W << " GPUs (RCCL communicator of " << nranks << " ranks), "
                << (feature_sharded ? "feature-block" : "document") << " sharding" << std::endl;
      std::cout << std::fixed << std::setprecision(4);
      std::cout << "# Training:" << std::endl << "# -------------------------" << std::endl
                << "# iter. training validation" << std::endl << "# -------------------------" << std::endl;
      t_train0 = std::chrono::high_resolution_clock::now();
    }
    // the metric of set `which` over all ranks' queries (metric.h:77-106), the same bits on
    // every rank: per-rank sums added in rank order
    auto metric_all = [&](int which) -> MetricScore {
      if (feature_sharded) {  // replicated documents: every rank evaluates everything
        MetricScore v = 0;
        QRM(c, qr_metric_eval(c, which, mcode, cutoff, &v));
        return v;
      }
      const size_t nq = which ? vq1 - vq0 : tq1 - tq0;
      MetricScore v = 0;
      if (nq) QRM(c, qr_metric_eval(c, which, mcode, cutoff, &v));
      sh.msum[which][r] = v * (double)nq;
      sh.mq[which][r] = (double)nq;
      sh.bar.wait();
      double s = 0.0, n = 0.0;
      for (int k = 0; k < W; ++k) {
        s += sh.msum[which][k];
        n += sh.mq[which][k];
      }
      sh.bar.wait();  // (everybody has read before the next call overwrites)
      return n > 0 ? s / n : 0.0;
    };
    auto report = [&](size_t iter, MetricScore on_training, const MetricScore *on_validation) {
      // best-model bookkeeping on every rank (same values), one table line from rank 0
      bool star = false;
      if (on_validation) {
        if (*on_validation > best_valid_r_[r]) {
          best_train_r_[r] = on_training;
          best_valid_r_[r] = *on_validation;
          best_model_r_[r] = iter - 1;
          star = true;
        }
      } else if (on_training > best_train_r_[r]) {
        best_train_r_[r] = on_training;
        best_model_r_[r] = iter - 1;
        star = true;
      }
      if (r == 0) {
        std::cout << std::setw(7) << iter << std::setw(9) << on_training;
        if (on_validation) std::cout << std::setw(9) << *on_validation;
        if (star) std::cout << " *";
        std::cout << std::endl;
      }
    };
    std::vector<qr_node_t> nodes(maxnodes);
    if (first) {  // the loaded model is the best one so far (mart.cc:244-253)
      const MetricScore mt = metric_all(0);
      MetricScore mv = 0;
      if (validation) mv = metric_all(1);
      best_train_r_[r] = mt;
      if (validation) best_valid_r_[r] = mv;
      best_model_r_[r] = first - 1;
      if (r == 0) {
        std::cout << std::setw(7) << first << std::setw(9) << mt;
        if (validation) std::cout << std::setw(9) << mv;
        std::cout << " *" << std::endl;
      }
    }
    // as Mart::learn: the metric rides with the next lambda pass (a sampled ranking is not the metric's)
    const bool lagged = lambda && !validation && subsample_ == 1.0f;
    size_t built = 0;
    for (size_t m = first; m < ntrees_; ++m) {
      if (validation && (valid_iterations_ && m > best_model_r_[r] + valid_iterations_)) break;
      // ---- pseudo-responses
      if (lambda)
        QRM(c, qr_lambda_compute(c, mcode, cutoff));  // lambdamart.cc:62-152
      else
        QRM(c, qr_residual_compute(c));               // mart.cc:418-431
      if (!feature_sharded) {
        sum64(x_scal, n_scal);
        QRM(c, qr_lambda_finish(c));
      }
      // ---- tree
      size_t nn = 0;
      bool scores_enqueued = false;
      if (obliv && !feature_sharded) {  // ot.cc:32-201 over document shards: one exchange per level
        QRM(c, qr_obl_begin(c, treedepth_, minleafsupport_));
        sum64(x_hist, n_hist);
        for (size_t level = 0; level < treedepth_; ++level) {
          QRM(c, qr_obl_propose(c, level));
          QRM(c, qr_obl_apply(c, level));
          if (level + 1 < treedepth_) {  // ot.cc:127: no histograms for the leaves
            void *x_level = nullptr;
            size_t n_level = 0;
            QRM(c, qr_obl_level_exchange(c, level, &x_level, &n_level));
            sum64(x_level, n_level);
          }
        }
        QRM(c, qr_tree_end(c, lambda, nullptr, nullptr));
        QRM(c, qr_doc_exchange_buffers(c, nullptr, nullptr, nullptr, nullptr, &x_leaf, &n_leaf));
        sum64(x_leaf, n_leaf);
        QRM(c, qr_tree_leaves_finish(c, lambda, nodes.data(), &nn));
      } else if (obliv) {  // level by level, feature-sharded
        QRM(c, qr_obl_begin(c, treedepth_, minleafsupport_));
        for (size_t level = 0; level < treedepth_; ++level) {
          QRM(c, qr_obl_propose(c, level));
          gather_recs();
          QRM(c, qr_obl_mark(c, level));
          sum_mask();
          QRM(c, qr_obl_apply(c, level));
        }
        QRM(c, qr_tree_end(c, lambda, nodes.data(), &nn));
      } else if (feature_sharded) {
        QRM(c, qr_tree_begin(c, nleaves_, minleafsupport_));
        gather_recs();
        for (size_t s = 0; s + 1 < nleaves_; ++s) {
          QRM(c, qr_tree_decide(c));
          sum_mask();
          QRM(c, qr_tree_apply(c));
          gather_recs();
        }
        QRM(c, qr_tree_decide(c));
        QRM(c, qr_tree_end(c, lambda, nodes.data(), &nn));
      } else if (doc_batch && qr_tree_batch_supported(c, nleaves_)) {
        // rt.cc:58-90 with up to two splits per exchange (include/qr_hip.h, qr_tree_batch_*): the
        // root's histogram, then one buffer of batch cells per step.  The number of steps is a
        // guess (the last tree's); the last control step says whether it sufficed -- the same
        // on every rank, which grow the same trees.
        size_t steps = 0;
        QRM(c, qr_tree_batch_begin(c, nleaves_, minleafsupport_, &steps));
        sum64(x_hist, n_hist);
        QRM(c, qr_tree_batch_root(c));
        void *x_batch = nullptr;
        size_t n_batch = 0;
        QRM(c, qr_tree_batch_exchange(c, &x_batch, &n_batch));
        auto run = [&](size_t k) {
          for (size_t s = 0; s < k; ++s) {
            QRM(c, qr_tree_batch_apply(c));
            sum64(x_batch, n_batch);
            QRM(c, qr_tree_batch_decide(c, s + 1 == k ? 1 : 0));
          }
        };
        run(steps);
        // The tree is ENDED behind the guess -- leaf kernels, leaf exchange, leaf values, score
        // update all enqueued (they leave at once on a tree whose steps did not suffice) -- and only
        // then does the host look at the last control step: the GPU works on the tree's tail
        // while the host waits, instead of idling until the host has seen the word and enqueued it.
        auto end_tree = [&]() {
          QRM(c, qr_tree_end(c, lambda, nullptr, nullptr));
          QRM(c, qr_doc_exchange_buffers(c, nullptr, nullptr, nullptr, nullptr, &x_leaf, &n_leaf));
          sum64(x_leaf, n_leaf);
          QRM(c, qr_tree_leaves_finish(c, lambda, nullptr, nullptr));  // (a carried-on tree: repeats the score update)
        };
        end_tree();
        QRM(c, qr_scores_update(c, shrinkage_));  // mart.cc:345, :356
        scores_enqueued = true;
        int incomplete = 0;
        QRM(c, qr_tree_batch_settle(c, &incomplete, nullptr));
        if (incomplete) {  // (re-opened by the settle call)
          for (size_t done = steps, piece = 1; incomplete; piece *= 2) {
            const size_t left = nleaves_ - 1 > done ? nleaves_ - 1 - done : 1;
            const size_t k = std::min(piece, left);
            run(k);
            done += k;
            QRM(c, qr_tree_batch_settle(c, &incomplete, nullptr));
          }
          end_tree();
        }
        QRM(c, qr_tree_nodes(c, nodes.data(), &nn));
      } else {
        QRM(c, qr_tree_begin(c, nleaves_, minleafsupport_));
        sum64(x_hist, n_hist);
        for (size_t s = 0; s + 1 < nleaves_; ++s) {
          QRM(c, qr_tree_decide(c));
          QRM(c, qr_tree_apply(c));
          sum64(x_hist, n_hist);
        }
        QRM(c, qr_tree_decide(c));
        QRM(c, qr_tree_end(c, lambda, nullptr, nullptr));
        QRM(c, qr_doc_exchange_buffers(c, nullptr, nullptr, nullptr, nullptr, &x_leaf, &n_leaf));
        sum64(x_leaf, n_leaf);
        QRM(c, qr_tree_leaves_finish(c, lambda, nodes.data(), &nn));
      }
      if (!scores_enqueued) QRM(c, qr_scores_update(c, shrinkage_));  // mart.cc:345, :356
      // ---- every rank must have built rank 0's tree
      if (r == 0) {
        memcpy(sh.nodes0.data(), nodes.data(), nn * sizeof(qr_node_t));
        sh.nn0 = nn;
      }
      sh.bar.wait();
      if (r != 0) {
        bool same = nn == sh.nn0;
        for (size_t i = 0; same && i < nn; ++i)
          same = nodes[i].feature == sh.nodes0[i].feature && nodes[i].thr_id == sh.nodes0[i].thr_id &&
                 nodes[i].left == sh.nodes0[i].left && nodes[i].right == sh.nodes0[i].right &&
                 nodes[i].nsamples == sh.nodes0[i].nsamples;
        if (!same) {
          std::cerr << "!!! rank " << r << " built a different tree than rank 0 in iteration " << m + 1
                    << std::endl;
          fatal_exit();
        }
      }
      sh.bar.wait();
      if (r == 0) ensemble_model_.push(tree_from_records(nodes.data(), 0), shrinkage_);  // mart.cc:342
      ++built;
      // ---- metrics, best model, early stop (mart.cc:347-376)
      if (lagged) {
        if (m > first) {
          MetricScore prev = 0;
          QRM(c, qr_metric_last(c, &prev));  // of the scores this iteration's lambda pass ranked
          report(m, prev, nullptr);
        }
      } else {
        const MetricScore mt = metric_all(0);
        MetricScore mv = 0;
        if (validation) mv = metric_all(1);
        report(m + 1, mt, validation ? &mv : nullptr);
      }
      if (r == 0 && partial_save != 0 && !output_basename.empty() && (m + 1) % partial_save == 0)
        save(output_basename, (int)(m + 1));
    }
    if (lagged && built) {
      const MetricScore last = metric_all(0);
      report(first + built, last, nullptr);
    }
    QRM(c, qr_synchronize(c));
    if (r == 0 && built)
      std::cout << "# collectives per tree: " << std::setprecision(1) << (double)n_coll / (double)built << ", "
                << (double)coll_bytes / (double)built / 1024.0 << " KB handed over per rank" << std::setprecision(4)
                << std::endl;
    sh.bar.wait();
    qr_ctx_destroy(c);
  };

  best_train_r_.assign(W, std::numeric_limits<double>::lowest());
  best_valid_r_.assign(W, std::numeric_limits<double>::lowest());
  best_model_r_.assign(W, 0);
  std::vector<std::thread> threads;
  for (int r = 0; r < W; ++r) threads.emplace_back(worker, r);
  for (auto &t : threads) t.join();
  for (int r = 0; r < W; ++r) {
    const ncclResult_t dr = ncclCommDestroy(comms[r]);
    if (dr != ncclSuccess)
      std::cerr << "!!! ncclCommDestroy (rank " << r << "): " << ncclGetErrorString(dr) << std::endl;
  }
  if (sh.relayout) {
    std::cout << std::endl
              << "# --shard docs: every distinct value a threshold -- the lists shard by feature: feature layout"
              << std::endl;
    learn_multi(training, validation, metric, cutoff, partial_save, output_basename, ngpus, true);
    return;
  }
  best_metric_on_training_ = best_train_r_[0];
  best_metric_on_validation_ = best_valid_r_[0];
  best_model_ = best_model_r_[0];
  // rollback to the best model observed on the validation data (mart.cc:390-395)
  if (validation)
    while (ensemble_model_.is_notempty() && ensemble_model_.get_size() > best_model_ + 1) ensemble_model_.pop();
  auto t_train1 = std::chrono::high_resolution_clock::now();
  std::cout << std::endl;
  std::cout << metric << "@" << cutoff << " on training data = " << best_metric_on_training_ << std::endl;
  if (validation)
    std::cout << metric << "@" << cutoff << " on validation data = " << best_metric_on_validation_
              << std::endl;
  std::cout << std::endl
            << "#\t Training Time: " << std::setprecision(2)
            << std::chrono::duration<double>(t_train1 - t_train0).count() << " s." << std::endl;
}

}  // namespace forests
}  // namespace learning
}  // namespace quickrank
