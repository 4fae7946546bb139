// dataset.h -- host-side Dataset mirror (include/data/dataset.h:37-142 of the
// reference): row-major f32 features, f32 labels, query offsets built from runs
// of equal consecutive qids (dataset.cc:63-87).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace quickrank {
typedef float Label;
typedef double Score;
typedef float Feature;
typedef unsigned int QueryID;
typedef double MetricScore;

namespace data {

class Dataset {
 public:
  Dataset(size_t n_instances, size_t n_features)
      : max_instances_(n_instances), num_features_(n_features),
        data_((Feature *)calloc(std::max<size_t>(n_instances * n_features, 1), sizeof(Feature))),
        labels_(n_instances, 0.0f) {
    // (calloc: the zero pages of a large matrix are first touched by whoever fills the rows --
    // the parallel SVMLight reader -- instead of being written once here by one thread)
    if (!data_) {
      std::cerr << "!!! Impossible to allocate the dataset." << std::endl;
      exit(EXIT_FAILURE);
    }
    offsets_.push_back(0);
  }
  ~Dataset() { free(data_); }
  Dataset(const Dataset &) = delete;
  Dataset &operator=(const Dataset &) = delete;

  // dataset.cc:63-87
  void addInstance(QueryID q_id, Label i_label, const std::vector<Feature> &i_features) {
    if (i_features.size() > num_features_ || num_instances_ == max_instances_) {
      std::cerr << "!!! Impossible to add a new instance to the dataset." << std::endl;
      exit(EXIT_FAILURE);
    }
    labels_[num_instances_] = i_label;
    Feature *row = data_ + num_instances_ * num_features_;
    for (size_t i = 0; i < i_features.size(); i++) row[i] = i_features[i];
    if (num_instances_ == 0 || last_instance_id_ != q_id) {
      num_queries_++;
      offsets_.push_back(0);
      last_instance_id_ = q_id;
    }
    num_instances_++;
    offsets_.back() = num_instances_;
  }

  // bulk construction (parallel readers): rows are written in place through at()
  // and set_label(); close_rows() then derives the queries from the qid column
  // with the same rule as addInstance (a new query at every change of qid)
  void set_label(size_t doc, Label l) { labels_[doc] = l; }
  void close_rows(const std::vector<QueryID> &qids) {
    for (size_t i = 0; i < qids.size() && i < max_instances_; ++i) {
      if (i == 0 || last_instance_id_ != qids[i]) {
        num_queries_++;
        offsets_.push_back(0);
        last_instance_id_ = qids[i];
      }
      offsets_.back() = i + 1;
    }
    num_instances_ = std::min(qids.size(), max_instances_);
  }

  Feature *at(size_t doc, size_t f) { return data_ + doc * num_features_ + f; }
  const Feature *at(size_t doc, size_t f) const { return data_ + doc * num_features_ + f; }
  Label getLabel(size_t doc) const { return labels_[doc]; }
  const Label *labels() const { return labels_.data(); }
  size_t offset(size_t q) const { return offsets_[q]; }
  const std::vector<uint64_t> &offsets() const { return offsets_; }
  size_t num_features() const { return num_features_; }
  size_t num_queries() const { return num_queries_; }
  size_t num_instances() const { return num_instances_; }

 private:
  size_t max_instances_, num_features_;
  size_t num_queries_ = 0, num_instances_ = 0;
  QueryID last_instance_id_ = 0;
  Feature *data_;
  std::vector<Label> labels_;
  std::vector<uint64_t> offsets_;
};

}  // namespace data
}  // namespace quickrank
