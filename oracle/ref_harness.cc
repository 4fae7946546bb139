// ref_harness.cc -- C-callable shim over the REFERENCE's own translation units.
//
// TEST INFRASTRUCTURE ONLY.  Buildable only where /root/reference exists (this
// container), and LOADED only here: the CPU tests that pin the restatement
// (tests/test_oracle_vs_ref.py) and tests/golden/make_golden.py.  Nothing that runs on
// the GPU box -- the `-m gpu` tests, __graft_entry__.smoke(), bench.py -- loads
// oracle/_ref/libqr_ref.so; they check against liboracle.so (the C restatement) and the
// committed golden vectors.
// It links the reference's unmodified sources
//   src/data/{dataset,vertical_dataset,queryresults,rankedresults}.cc
//   src/metric/ir/{dcg,ndcg}.cc   src/learning/tree/rtnode_histogram.cc
//   src/utils/radix.cc   src/io/svml.cc   src/utils/strutils.cc
// and instantiates the header-only include/utils/maxheap.h (the growth-order heap of
// RegressionTree::fit) and include/utils/symmatrix.h (the jacobian's packed index)
// (the subset of the hot path that compiles from its own files; everything that
// includes learning/tree/rtnode.h needs the un-vendored pugixml submodule and is
// treated as unbuildable -- no stand-in headers are written).
// No reference source text is copied here: this file only #includes reference
// headers by path and calls their public API.
#include <algorithm>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "data/dataset.h"
#include "data/queryresults.h"
#include "data/rankedresults.h"
#include "data/vertical_dataset.h"
#include "learning/tree/rtnode_histogram.h"
#include "metric/ir/dcg.h"
#include "metric/ir/ndcg.h"
#include "io/svml.h"
#include "utils/radix.h"
#include "utils/maxheap.h"
#include "utils/symmatrix.h"

using namespace quickrank;

namespace {
struct ScoreDesc {
  const double *v;
  bool operator()(int i, int j) const { return v[i] > v[j]; }
};
std::shared_ptr<metric::ir::Metric> make_metric(int m, size_t cutoff) {
  if (m) return std::make_shared<metric::ir::Ndcg>(cutoff);
  return std::make_shared<metric::ir::Dcg>(cutoff);
}
std::shared_ptr<data::Dataset> make_dataset(const float *rowmajor,
                                            const float *labels,
                                            const uint64_t *qoff, size_t nq,
                                            size_t F) {
  const size_t N = qoff[nq];
  auto ds = std::make_shared<data::Dataset>(N, F);
  for (size_t q = 0; q < nq; ++q)
    for (size_t i = qoff[q]; i < qoff[q + 1]; ++i) {
      std::vector<Feature> x(F, 0.0f);
      if (rowmajor) x.assign(rowmajor + i * F, rowmajor + (i + 1) * F);
      ds->addInstance((QueryID)(q + 1), labels[i], x);
    }
  return ds;
}
}  // namespace

extern "C" {

// QueryResults::indexing_of_sorted_labels (queryresults.cc:47-53)
void ref_rank_by_score(const double *scores, size_t n, uint64_t *idx) {
  data::QueryResults qr(n, nullptr, nullptr);
  static_assert(sizeof(size_t) == sizeof(uint64_t), "size_t");
  qr.indexing_of_sorted_labels(scores, (size_t *)idx);
}

// libstdc++ std::partial_sort(first,last,last): the introsort depth-limit
// fallback, exercised directly (system library, not reference code).
void ref_heapsort_by_score(const double *scores, size_t n, uint64_t *idx) {
  std::vector<size_t> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = i;
  ScoreDesc c{scores};
  std::partial_sort(v.begin(), v.end(), v.end(), c);
  for (size_t i = 0; i < n; ++i) idx[i] = v[i];
}

// the label sort of Ndcg::compute_idcg (ndcg.cc:40-41), same libstdc++ call
void ref_sort_labels_desc_int(float *labels, size_t n) {
  std::sort(labels, labels + n, std::greater<int>());
}

double ref_eval_query(int m, const float *labels, const double *scores,
                      size_t n, size_t cutoff) {
  auto metric = make_metric(m, cutoff);
  data::QueryResults qr(n, const_cast<float *>(labels), nullptr);
  return metric->evaluate_result_list(&qr, scores);
}

double ref_eval_dataset(int m, const float *labels, const double *scores,
                        const uint64_t *qoff, size_t nq, size_t cutoff,
                        int vertical) {
  auto metric = make_metric(m, cutoff);
  auto ds = make_dataset(nullptr, labels, qoff, nq, 1);
  if (vertical) {
    auto v = std::make_shared<data::VerticalDataset>(ds);
    return metric->evaluate_dataset(v, scores);
  }
  return metric->evaluate_dataset(ds, scores);
}

// RankedResults ctor (rankedresults.cc:27-41) + Metric::jacobian
// out: packed upper triangle n(n+1)/2; sorted_labels_out[n]; unmap_out[n]
void ref_jacobian(int m, const float *labels, const double *scores, size_t n,
                  size_t cutoff, double *out, float *sorted_labels_out,
                  uint64_t *unmap_out) {
  auto metric = make_metric(m, cutoff);
  auto qr = std::make_shared<data::QueryResults>(
      n, const_cast<float *>(labels), nullptr);
  auto ranked = std::make_shared<data::RankedResults>(
      qr, const_cast<double *>(scores));
  auto jac = metric->jacobian(ranked);
  const size_t tri = n * (n + 1) / 2;
  for (size_t i = 0; i < tri; ++i) out[i] = jac->at(i);
  for (size_t i = 0; i < n; ++i) {
    sorted_labels_out[i] = ranked->sorted_labels()[i];
    unmap_out[i] = ranked->pos_of_rank(i);
  }
}

// idx_radixsort (radix.cc:35-73)
void ref_argsort_f32(const float *v, size_t n, uint64_t *idx) {
  auto r = idx_radixsort(v, n);
  for (size_t i = 0; i < n; ++i) idx[i] = r[i];
}

// RTRootHistogram ctor + update + child ctor + sibling-by-subtraction
// (rtnode_histogram.cc:41-87, 172-253).  thr is [F][cap]; outputs [F][cap].
// left_ids/nleft: the left child's sample ids.  transform != 0 uses
// transform_intorightchild on a copy path (non-root case), else the
// (parent,left) constructor (root case).
void ref_histograms(const float *rowmajor, size_t N, size_t F, const float *thr,
                    const uint64_t *thr_size, size_t cap, const double *labels,
                    const uint64_t *left_ids, size_t nleft, int transform,
                    uint32_t *stmap_out, uint64_t *count0_out, double *root_sum,
                    uint64_t *root_count, double *root_ss, double *left_sum,
                    uint64_t *left_count, double *left_ss, double *right_sum,
                    uint64_t *right_count, double *right_ss) {
  std::vector<uint64_t> qoff = {0, N};
  std::vector<float> lab(N, 0.0f);
  auto ds = make_dataset(rowmajor, lab.data(), qoff.data(), 1, F);
  data::VerticalDataset vds(ds);
  std::vector<size_t *> sorted(F);
  std::vector<float *> thrp(F);
  std::vector<size_t> ts(F);
  for (size_t f = 0; f < F; ++f) {
    sorted[f] = idx_radixsort(vds.at(0, f), N).release();
    thrp[f] = const_cast<float *>(thr + f * cap);
    ts[f] = thr_size[f];
  }
  RTRootHistogram *root =
      new RTRootHistogram(&vds, sorted.data(), N, thrp.data(), ts.data());
  for (size_t f = 0; f < F; ++f) {
    for (size_t i = 0; i < N; ++i) stmap_out[f * N + i] = (uint32_t)root->stmap[f][i];
    for (size_t t = 0; t < ts[f]; ++t) count0_out[f * cap + t] = root->count[f][t];
  }
  std::vector<size_t> ids(N);
  for (size_t i = 0; i < N; ++i) ids[i] = i;
  root->update(const_cast<double *>(labels), N, ids.data());
  auto dump = [&](RTNodeHistogram *h, double *s, uint64_t *c, double *ss) {
    for (size_t f = 0; f < F; ++f)
      for (size_t t = 0; t < ts[f]; ++t) {
        s[f * cap + t] = h->sumlbl[f][t];
        c[f * cap + t] = h->count[f][t];
      }
    *ss = h->squares_sum_;
  };
  dump(root, root_sum, root_count, root_ss);
  std::vector<size_t> lids(left_ids, left_ids + nleft);
  RTNodeHistogram *left = new RTNodeHistogram(root, lids.data(), nleft, labels);
  dump(left, left_sum, left_count, left_ss);
  if (transform) {
    // non-root path: the parent histogram is turned into the right child
    RTNodeHistogram *p = new RTNodeHistogram(root, ids.data(), N, labels);
    p->transform_intorightchild(left);
    dump(p, right_sum, right_count, right_ss);
    delete p;
  } else {
    RTNodeHistogram *r = new RTNodeHistogram(root, left);
    dump(r, right_sum, right_count, right_ss);
    delete r;
  }
  delete left;
  delete root;
  for (size_t f = 0; f < F; ++f) delete[] sorted[f];
}

// Svml::read_horizontal (svml.cc:38-161).  Call with NULL buffers to size them.
int ref_svml_read(const char *path, size_t *N, size_t *F, size_t *Q, float *x, float *labels,
                  uint64_t *qoff) {
  io::Svml reader;
  auto ds = reader.read_horizontal(path);
  *N = ds->num_instances();
  *F = ds->num_features();
  *Q = ds->num_queries();
  if (x) memcpy(x, ds->at(0, 0), *N * *F * sizeof(float));
  if (labels)
    for (size_t i = 0; i < *N; ++i) labels[i] = ds->getLabel(i);
  if (qoff)
    for (size_t q = 0; q <= *Q; ++q) qoff[q] = ds->offset(q);
  return 0;
}

// Svml::write (svml.cc:163-188)
int ref_svml_write(const char *path, const float *x, const float *labels, const uint64_t *qoff,
                   size_t Q, size_t F) {
  auto ds = make_dataset(x, labels, qoff, Q, F);
  io::Svml().write(ds, path);
  return 0;
}

// MaxHeap<int> (maxheap.h:58-88), driven by a sequence of operations: ops[i] >= 0 =
// push(keys[i], ops[i]); ops[i] < 0 = pop.  top_out[i] = top() after operation i (-1
// when the heap is empty), size_out[i] = get_size().  The constructor's initial size
// (RegressionTree::fit passes the leaf count, rt.cc:57) only decides when it reallocs.
void ref_heap_trace(const double *keys, const int32_t *ops, size_t n, size_t initsize,
                    int32_t *top_out, uint64_t *size_out) {
  MaxHeap<int> h(initsize);
  for (size_t i = 0; i < n; ++i) {
    if (ops[i] >= 0)
      h.push(keys[i], ops[i]);
    else if (h.is_notempty())
      h.pop();
    top_out[i] = h.is_notempty() ? h.top() : -1;
    size_out[i] = h.get_size();
  }
}

// SymMatrix<double>::at(i, j) (symmatrix.h:29-89): position of (i, j) in the packed
// array, for every pair; out is [size][size]
void ref_sym_index(size_t size, uint64_t *out) {
  SymMatrix<double> m(size);
  for (size_t i = 0; i < size; ++i)
    for (size_t j = 0; j < size; ++j) out[i * size + j] = (uint64_t)(m.vectat(i, j) - m.vectat(0, 0));
}

// QueryResults::sorted_labels (queryresults.cc:55-62): the labels of the first `cutoff` ranks
void ref_sorted_labels(const float *labels, const double *scores, size_t n, size_t cutoff, float *dest) {
  data::QueryResults qr(n, const_cast<float *>(labels), nullptr);
  qr.sorted_labels(scores, dest, cutoff);
}

// Dataset::addInstance over a qid COLUMN (dataset.cc:63-87: a new query at every change of
// qid, repeated and non-monotone ids included) and the VerticalDataset made of it
// (vertical_dataset.cc: feature-major transposition).  Returns the number of queries;
// offsets_out[nq + 1], vertical_out [F][N], vertical_offsets_out[nq + 1] (when not null).
size_t ref_dataset_layout(const float *rowmajor, const float *labels, const uint32_t *qids, size_t N, size_t F,
                          uint64_t *offsets_out, float *vertical_out, uint64_t *vertical_offsets_out) {
  auto ds = std::make_shared<data::Dataset>(N, F);
  for (size_t i = 0; i < N; ++i) {
    std::vector<Feature> x(rowmajor + i * F, rowmajor + (i + 1) * F);
    ds->addInstance((QueryID)qids[i], labels[i], x);
  }
  const size_t nq = ds->num_queries();
  if (offsets_out)
    for (size_t q = 0; q <= nq; ++q) offsets_out[q] = ds->offset(q);
  if (vertical_out) {
    data::VerticalDataset v(ds);
    for (size_t f = 0; f < F; ++f)
      for (size_t i = 0; i < N; ++i) vertical_out[f * N + i] = *v.at(i, f);
    if (vertical_offsets_out)
      for (size_t q = 0; q <= v.num_queries(); ++q) vertical_offsets_out[q] = v.offset(q);
  }
  return nq;
}

}  // extern "C"
