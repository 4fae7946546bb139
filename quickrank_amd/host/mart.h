// mart.h -- C++ host mirror of Mart / LambdaMart / ObliviousMart /
// ObliviousLambdaMart (include/learning/forests/{mart,lambdamart,obliviousmart,
// obliviouslambdamart}.h of the reference) on top of the C-ABI device layer
// (include/qr_hip.h).  Same constructor parameters, same learn() phases and
// console output, same XML model; the tree kernels run on the GPU.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/qr_hip.h"
#include "dataset.h"
#include "xml.h"

namespace quickrank {
namespace learning {
namespace forests {

// rtnode.h:37-132
struct RTNode {
  float threshold = 0.0f;
  double avglabel = 0.0;
  int featureidx = -1;     // -1 == uint_max: leaf
  unsigned featureid = 0;  // 1-based id written to the model file
  std::unique_ptr<RTNode> left, right;
  bool is_leaf() const { return featureidx < 0; }
  double score_instance(const Feature *d) const {
    return is_leaf() ? avglabel
                     : (d[featureidx] <= threshold ? left->score_instance(d)
                                                   : right->score_instance(d));
  }
  void append_xml_model(xml::Node *parent, const std::string &pos = "") const;  // rtnode.cc:48-77
  static std::unique_ptr<RTNode> parse_xml(const xml::Node &split);             // rtnode.cc:79-117
};

// ensemble.h:33-105
class Ensemble {
 public:
  void set_capacity(size_t n) { capacity_ = n; }
  void push(std::unique_ptr<RTNode> root, double weight);  // exits when full (ensemble.cc:97-103)
  void pop() { roots_.pop_back(); weights_.pop_back(); }
  size_t get_size() const { return roots_.size(); }
  bool is_notempty() const { return !roots_.empty(); }
  const RTNode *getTree(size_t i) const { return roots_[i].get(); }
  double getWeight(size_t i) const { return weights_[i]; }
  void append_xml_model(xml::Node *parent) const;  // ensemble.cc:133-147
  // flat records for qr_ensemble_upload; returns max_nodes
  size_t flatten(std::vector<qr_node_t> *nodes, std::vector<double> *weights) const;

 private:
  std::vector<std::unique_ptr<RTNode>> roots_;
  std::vector<double> weights_;
  size_t capacity_ = 0;
};

class Mart {
 public:
  enum Algo { MART = 0, LAMBDAMART = 1, OBVMART = 2, OBVLAMBDAMART = 3 };
  static const char *algo_name(Algo a);
  static bool algo_from_name(const std::string &s, Algo *out);

  Mart(Algo algo, size_t ntrees, double shrinkage, size_t nthresholds, size_t ntreeleaves,
       size_t minleafsupport, size_t valid_iterations, size_t treedepth = 3);
  explicit Mart(const xml::Node &model);  // mart.cc:37-89, obliviousmart.cc:35-40
  ~Mart();

  std::string name() const { return algo_name(algo_); }
  // --subsample / --max-features (mart.cc:287-329, rt.cc:222-243); seed 0 = from the
  // clock, as the reference always does
  void set_sampling(float subsample, float max_features, unsigned long long seed) {
    subsample_ = subsample;
    max_features_ = max_features;
    sampling_seed_ = seed;
  }
  void print(std::ostream &os) const;  // mart.cc:95-115 / obliviousmart.cc:42-57

  // mart.cc:208-416.  metric: "NDCG" | "DCG".
  void learn(std::shared_ptr<data::Dataset> training, std::shared_ptr<data::Dataset> validation,
             const std::string &metric, size_t cutoff, size_t partial_save,
             const std::string &output_basename);
  // The same on `ngpus` GPUs of one node (host/mart_multi.cc): a host thread per GPU,
  // RCCL on the contexts' streams; documents (default) or feature blocks sharded.
  void learn_multi(std::shared_ptr<data::Dataset> training, std::shared_ptr<data::Dataset> validation,
                   const std::string &metric, size_t cutoff, size_t partial_save,
                   const std::string &output_basename, int ngpus, bool feature_sharded);
  // ltr_algorithm.cc:44-52 on the device
  void score_dataset(const data::Dataset &dataset, Score *scores, float *kernel_ms = nullptr);
  // Driver::extract_partial_scores (driver.cc:411-445) over Ensemble::partial_scores_instance
  // (ensemble.cc:120-131): a dataset of ntrees "features" = the per-tree scores cast to
  // Feature (f32), labels and query boundaries of `dataset`
  std::shared_ptr<data::Dataset> partial_scores(const data::Dataset &dataset, bool ignore_weights = false);
  // metric.h:77-106 on the host scores of a loaded dataset (device evaluation)
  MetricScore evaluate(const data::Dataset &dataset, const Score *scores, const std::string &metric,
                       size_t cutoff);
  std::unique_ptr<xml::Node> get_xml_model() const;  // mart.cc:470-491, obliviousmart.cc:66-84
  void save(const std::string &output_basename, int iteration = -1) const;  // ltr_algorithm.cc:54-66
  static std::shared_ptr<Mart> load_model_from_file(const std::string &model_filename);
  bool import_model_state(Mart &other);  // mart.cc:493-517
  const Ensemble &ensemble() const { return ensemble_model_; }
  // the <info> block as numbers: trees, thresholds, leaves, min leaf support, early-stop rounds, depth
  void info(int *algo, size_t out[6], double *shrinkage) const {
    *algo = (int)algo_;
    out[0] = ntrees_, out[1] = nthresholds_, out[2] = nleaves_, out[3] = minleafsupport_;
    out[4] = valid_iterations_, out[5] = treedepth_;
    *shrinkage = shrinkage_;
  }

 private:
  void ensure_ctx();
  std::unique_ptr<RTNode> tree_from_records(const qr_node_t *nodes, int i) const;

  Algo algo_;
  size_t ntrees_, nthresholds_, nleaves_, minleafsupport_, valid_iterations_, treedepth_;
  double shrinkage_;
  float subsample_ = 1.0f, max_features_ = 1.0f, collapse_leaves_factor_ = 0.0f;
  unsigned long long sampling_seed_ = 0;  // of the document / feature sampling streams
  Ensemble ensemble_model_;
  qr_ctx *ctx_ = nullptr;
  MetricScore best_metric_on_training_ = 0, best_metric_on_validation_ = 0;
  size_t best_model_ = 0;
  // learn_multi: every rank's thread keeps the (identical) best-model bookkeeping
  std::vector<MetricScore> best_train_r_, best_valid_r_;
  std::vector<size_t> best_model_r_;
};

}  // namespace forests
}  // namespace learning
}  // namespace quickrank
