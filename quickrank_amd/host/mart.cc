#include "mart.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>

namespace quickrank {
namespace learning {
namespace forests {

// ---- RTNode ------------------------------------------------------------------
void RTNode::append_xml_model(xml::Node *parent, const std::string &pos) const {
  xml::Node *split = parent->append_child("split");
  if (!pos.empty()) split->append_attribute("pos", pos);
  // (rtnode.cc:48-77 streams with setprecision(max_digits10): "%.17g" / "%.9g")
  if (is_leaf()) {
    split->append_child("output")->text = xml::fmt_double(avglabel);
  } else {
    split->append_child("feature")->text = std::to_string(featureid);
    split->append_child("threshold")->text = xml::fmt_float(threshold);
    left->append_xml_model(split, "left");
    right->append_xml_model(split, "right");
  }
}

std::unique_ptr<RTNode> RTNode::parse_xml(const xml::Node &split_xml) {
  std::unique_ptr<RTNode> left_child, right_child;
  bool is_leaf = false;
  unsigned feature_id = 0;
  float threshold = 0.0f;
  double prediction = 0.0;
  for (const auto &c : split_xml.children) {
    if (c->name == "output") {
      prediction = strtod(c->text.c_str(), nullptr);
      is_leaf = true;
      break;
    } else if (c->name == "feature") {
      feature_id = (unsigned)strtoul(c->text.c_str(), nullptr, 10);
    } else if (c->name == "threshold") {
      threshold = strtof(c->text.c_str(), nullptr);
    } else if (c->name == "split") {
      if (c->attribute("pos") == "left")
        left_child = parse_xml(*c);
      else
        right_child = parse_xml(*c);
    }
  }
  std::unique_ptr<RTNode> n(new RTNode());
  if (is_leaf) {
    n->avglabel = prediction;
  } else {
    n->threshold = threshold;
    n->featureidx = (int)feature_id - 1;  // rtnode.cc:113
    n->featureid = feature_id;
    n->left = std::move(left_child);
    n->right = std::move(right_child);
    if (!n->left || !n->right) return nullptr;
  }
  return n;
}

// ---- Ensemble ----------------------------------------------------------------
void Ensemble::push(std::unique_ptr<RTNode> root, double weight) {
  if (roots_.size() >= capacity_) {
    std::cerr << "Error adding a new tree into the ensemble, capacity reached!";
    exit(1);
  }
  roots_.push_back(std::move(root));
  weights_.push_back(weight);
}

void Ensemble::append_xml_model(xml::Node *parent) const {
  xml::Node *ensemble = parent->append_child("ensemble");
  for (size_t i = 0; i < roots_.size(); ++i) {
    xml::Node *tree = ensemble->append_child("tree");
    tree->append_attribute("id", std::to_string(i + 1));
    tree->append_attribute("weight", xml::fmt_double(weights_[i]));
  }
  const long nt = (long)roots_.size();   // the trees' own elements: independent, on all host threads
#pragma omp parallel for schedule(static)
  for (long i = 0; i < nt; ++i)
    if (roots_[(size_t)i]) roots_[(size_t)i]->append_xml_model(ensemble->children[(size_t)i].get());
}

static size_t count_nodes(const RTNode *n) {
  return n->is_leaf() ? 1 : 1 + count_nodes(n->left.get()) + count_nodes(n->right.get());
}
static int flatten_rec(const RTNode *n, qr_node_t *out, int *next) {
  const int me = (*next)++;
  qr_node_t &r = out[me];
  memset(&r, 0, sizeof r);
  r.feature = n->featureidx;
  r.thr_id = -1;
  r.threshold = n->threshold;
  r.left = r.right = -1;
  r.value = n->avglabel;
  if (!n->is_leaf()) {
    const int l = flatten_rec(n->left.get(), out, next);
    const int rr = flatten_rec(n->right.get(), out, next);
    out[me].left = l;
    out[me].right = rr;
  }
  return me;
}
size_t Ensemble::flatten(std::vector<qr_node_t> *nodes, std::vector<double> *weights) const {
  size_t max_nodes = 1;
  for (auto &r : roots_) max_nodes = std::max(max_nodes, count_nodes(r.get()));
  nodes->assign(roots_.size() * max_nodes, qr_node_t());
  for (size_t t = 0; t < roots_.size(); ++t) {
    for (size_t i = 0; i < max_nodes; ++i) {
      qr_node_t &pad = (*nodes)[t * max_nodes + i];
      pad.feature = -1;
      pad.left = pad.right = -1;
      pad.thr_id = -1;
    }
    int next = 0;
    flatten_rec(roots_[t].get(), nodes->data() + t * max_nodes, &next);
  }
  *weights = weights_;
  return max_nodes;
}

// ---- Mart --------------------------------------------------------------------
const char *Mart::algo_name(Algo a) {
  switch (a) {
    case MART: return "MART";                    // mart.cc:35
    case LAMBDAMART: return "LAMBDAMART";        // lambdamart.cc:30
    case OBVMART: return "OBVMART";              // obliviousmart.cc:33
    default: return "OBVLAMBDAMART";             // obliviouslambdamart.cc:33
  }
}
bool Mart::algo_from_name(const std::string &s, Algo *out) {
  for (int a = 0; a < 4; ++a)
    if (s == algo_name((Algo)a)) {
      *out = (Algo)a;
      return true;
    }
  return false;
}

static void qr_die(qr_ctx *c, const char *what) {  // dataset.cc:37-43 convention
  std::cerr << "!!! " << what << ": " << qr_last_error(c) << std::endl;
  exit(EXIT_FAILURE);
}
#define QR(call)                          \
  do {                                    \
    if ((call) != QR_OK) qr_die(ctx_, #call); \
  } while (0)

Mart::Mart(Algo algo, size_t ntrees, double shrinkage, size_t nthresholds, size_t ntreeleaves,
           size_t minleafsupport, size_t valid_iterations, size_t treedepth)
    : algo_(algo), ntrees_(ntrees), nthresholds_(nthresholds),
      nleaves_(algo >= OBVMART ? ((size_t)1 << treedepth) : ntreeleaves),  // obliviousmart.h ctor
      minleafsupport_(minleafsupport), valid_iterations_(valid_iterations), treedepth_(treedepth),
      shrinkage_(shrinkage) {}

Mart::Mart(const xml::Node &model) : algo_(MART), treedepth_(3) {
  const xml::Node *info = model.child("info");
  const xml::Node *ens = model.child("ensemble");
  if (!info || !ens) {
    std::cerr << "!!! Unable to parse tree from XML model." << std::endl;
    exit(EXIT_FAILURE);
  }
  algo_from_name(info->child_text("type"), &algo_);
  ntrees_ = strtoul(info->child_text("trees", "0").c_str(), nullptr, 10);
  nleaves_ = strtoul(info->child_text("leaves", "0").c_str(), nullptr, 10);
  minleafsupport_ = strtoul(info->child_text("leafsupport", "0").c_str(), nullptr, 10);
  nthresholds_ = strtoul(info->child_text("discretization", "0").c_str(), nullptr, 10);
  valid_iterations_ = strtoul(info->child_text("estop", "0").c_str(), nullptr, 10);
  shrinkage_ = strtod(info->child_text("shrinkage", "0").c_str(), nullptr);
  if (info->child("subsample")) subsample_ = strtof(info->child_text("subsample").c_str(), nullptr);
  if (info->child("max_features")) max_features_ = strtof(info->child_text("max_features").c_str(), nullptr);
  if (info->child("collapse_leaves_factor"))
    collapse_leaves_factor_ = strtof(info->child_text("collapse_leaves_factor").c_str(), nullptr);
  if (info->child("depth")) treedepth_ = strtoul(info->child_text("depth").c_str(), nullptr, 10);
  ensemble_model_.set_capacity(ntrees_);
  // (the trees are independent: built on all host threads, pushed in file order)
  const long nt = (long)ens->children.size();
  std::vector<std::unique_ptr<RTNode>> roots((size_t)nt);
  std::vector<double> weights((size_t)nt);
#pragma omp parallel for schedule(static)
  for (long t = 0; t < nt; ++t) {
    const xml::Node &tree = *ens->children[(size_t)t];
    weights[(size_t)t] = strtod(tree.attribute("weight").c_str(), nullptr);
    if (const xml::Node *rs = tree.child("split")) roots[(size_t)t] = RTNode::parse_xml(*rs);
  }
  for (long t = 0; t < nt; ++t) {
    if (!roots[(size_t)t]) {
      std::cerr << "!!! Unable to parse tree from XML model." << std::endl;
      exit(EXIT_FAILURE);
    }
    ensemble_model_.push(std::move(roots[(size_t)t]), weights[(size_t)t]);
  }
}

Mart::~Mart() {
  if (ctx_) qr_ctx_destroy(ctx_);
}

void Mart::print(std::ostream &os) const {
  os << "# Ranker: " << name() << std::endl << "# max no. of trees = " << ntrees_ << std::endl;
  if (algo_ >= OBVMART)
    os << "# max tree depth = " << treedepth_ << std::endl;
  else
    os << "# no. of tree leaves = " << nleaves_ << std::endl;
  os << "# shrinkage = " << shrinkage_ << std::endl
     << "# min leaf support = " << minleafsupport_ << std::endl;
  if (algo_ < OBVMART) {
    os << "# subsample = " << subsample_ << std::endl << "# max_features = " << max_features_ << std::endl;
  }
  if (nthresholds_)
    os << "# no. of thresholds = " << nthresholds_ << std::endl;
  else
    os << "# no. of thresholds = unlimited" << std::endl;
  if (valid_iterations_)
    os << "# no. of no gain rounds before early stop = " << valid_iterations_ << std::endl;
}

void Mart::ensure_ctx() {
  if (ctx_) return;
  if (qr_ctx_create(0, &ctx_) != QR_OK) {
    std::cerr << "!!! " << qr_last_error(nullptr) << std::endl;
    exit(EXIT_FAILURE);
  }
}

std::unique_ptr<RTNode> Mart::tree_from_records(const qr_node_t *nodes, int i) const {
  std::unique_ptr<RTNode> n(new RTNode());
  const qr_node_t &r = nodes[i];
  n->avglabel = r.value;
  if (r.feature >= 0) {
    n->featureidx = r.feature;
    n->featureid = (unsigned)r.feature + 1;  // rt.cc:350-352
    n->threshold = r.threshold;
    n->left = tree_from_records(nodes, r.left);
    n->right = tree_from_records(nodes, r.right);
  }
  return n;
}

static int metric_code(const std::string &m) {
  if (m == "NDCG") return QR_METRIC_NDCG;
  if (m == "DCG") return QR_METRIC_DCG;
  std::cerr << " !! Train Metric was not set properly" << std::endl;  // driver.cc:114-117
  exit(EXIT_FAILURE);
}

void Mart::learn(std::shared_ptr<data::Dataset> training, std::shared_ptr<data::Dataset> validation,
                 const std::string &metric, size_t cutoff, size_t partial_save,
                 const std::string &output_basename) {
  std::cout << "# Initialization";
  std::cout.flush();
  auto t_init0 = std::chrono::high_resolution_clock::now();
  ensure_ctx();
  const int mcode = metric_code(metric);
  const bool lambda = algo_ == LAMBDAMART || algo_ == OBVLAMBDAMART;
  const bool obliv = algo_ >= OBVMART;
  best_metric_on_validation_ = std::numeric_limits<double>::lowest();
  best_metric_on_training_ = std::numeric_limits<double>::lowest();
  best_model_ = 0;
  ensemble_model_.set_capacity(ntrees_);
  // init(): thresholds + bin map on the device (mart.cc:117-176)
  QR(qr_dataset_upload(ctx_, training->at(0, 0), training->num_instances(), training->num_features(),
                       training->labels(), training->offsets().data(), training->num_queries()));
  if (validation)
    QR(qr_valid_upload(ctx_, validation->at(0, 0), validation->num_instances(), validation->num_features(),
                       validation->labels(),
                       validation->offsets().data(), validation->num_queries()));
  // Up to 255 thresholds per feature: u8 bins.  More -- --num-thresholds above 255, or the
  // default 0 ("every distinct value", quicklearn.cc:103) on a column with more than 255 of
  // them -- the wide path (u32 bins, ragged threshold rows): same trees, slower kernels.
  int brc = nthresholds_ <= 255 ? qr_bins_build(ctx_, nthresholds_, nullptr, nullptr) : QR_ERR_UNSUPPORTED;
  if (brc == QR_ERR_UNSUPPORTED && (nthresholds_ == 0 || nthresholds_ > 255)) {
    size_t cells = 0, cap = 0;
    brc = qr_bins_build_wide(ctx_, nthresholds_, &cells, &cap);
    if (brc == QR_OK)
      std::cout << "# " << cells << " threshold slots over " << training->num_features()
                << " features (up to " << cap << " per feature): 32-bit bins" << std::endl;
  }
  QR(brc);
  QR(qr_scores_reset(ctx_));
  const bool sampling = subsample_ != 1.0f || max_features_ != 1.0f;
  if (sampling) {
    if (obliv && max_features_ != 1.0f) {  // (ObliviousRT::fit, ot.cc:32-201, has no feature sampling)
      std::cerr << "!!! --max-features applies to MART / LAMBDAMART." << std::endl;
      exit(EXIT_FAILURE);
    }
    unsigned long long seed = sampling_seed_;
    if (seed == 0)  // the reference seeds from the clock at every draw
      seed = (unsigned long long)std::chrono::system_clock::now().time_since_epoch().count();
    if (subsample_ != 1.0f) QR(qr_subsample_set(ctx_, subsample_, seed));
    if (max_features_ != 1.0f) QR(qr_tree_set_max_features(ctx_, max_features_, seed));
  }
  // restart from a previously saved model (mart.cc:237-253)
  if (ensemble_model_.is_notempty()) {
    best_model_ = ensemble_model_.get_size() - 1;
    std::vector<Score> s(training->num_instances());
    score_dataset(*training, s.data());
    QR(qr_scores_set(ctx_, s.data()));
    QR(qr_metric_eval(ctx_, 0, mcode, cutoff, &best_metric_on_training_));
    if (validation) {
      std::vector<Score> v(validation->num_instances());
      score_dataset(*validation, v.data());
      QR(qr_valid_scores_set(ctx_, v.data()));
      QR(qr_metric_eval(ctx_, 1, mcode, cutoff, &best_metric_on_validation_));
    }
  }
  auto t_init1 = std::chrono::high_resolution_clock::now();
  std::cout << ": " << std::setprecision(2) << std::chrono::duration<double>(t_init1 - t_init0).count()
            << " s." << std::endl;

  std::cout << std::fixed << std::setprecision(4);
  std::cout << "# Training:" << std::endl;
  std::cout << "# -------------------------" << std::endl;
  std::cout << "# iter. training validation" << std::endl;
  std::cout << "# -------------------------" << std::endl;
  if (ensemble_model_.is_notempty()) {
    std::cout << std::setw(7) << ensemble_model_.get_size() << std::setw(9) << best_metric_on_training_;
    if (validation) std::cout << std::setw(9) << best_metric_on_validation_;
    std::cout << " *" << std::endl;
  }
  auto t_train0 = std::chrono::high_resolution_clock::now();
  std::vector<qr_node_t> nodes(obliv ? ((size_t)1 << (treedepth_ + 1)) : 2 * nleaves_ + 1);
  // One table line (mart.cc:351-376): the best-model bookkeeping on the way.
  auto report = [&](size_t iter, MetricScore on_training, const MetricScore *on_validation) {
    std::cout << std::setw(7) << iter << std::setw(9) << on_training;
    if (on_validation) {
      std::cout << std::setw(9) << *on_validation;
      if (*on_validation > best_metric_on_validation_) {
        best_metric_on_training_ = on_training;
        best_metric_on_validation_ = *on_validation;
        best_model_ = iter - 1;
        std::cout << " *";
      }
    } else if (on_training > best_metric_on_training_) {
      best_metric_on_training_ = on_training;
      best_model_ = iter - 1;
      std::cout << " *";
    }
    std::cout << std::endl;
  };
  // Without a validation set nothing decided inside the loop depends on the
  // training metric, and the lambda pass of iteration m+1 ranks exactly the scores
  // whose metric iteration m reports (mart.cc:347 vs lambdamart.cc:104): the value
  // is then a by-product of that pass (qr_metric_last) and the line of iteration m
  // is printed one iteration late -- same numbers, one ranking pass per iteration
  // instead of two, and no stream drain in between.
  const bool lagged = lambda && !validation && subsample_ == 1.0f;  // sampled rankings are not the metric's
  const size_t first = ensemble_model_.get_size();
  // The records of tree m are fetched after iteration m + 1's first pass is enqueued: the
  // library settles a tree on its last control call, before its leaf kernels and score update
  // have run, so that pass lines up under them and the device never waits for the host
  // (nothing of tree m is overwritten before the next fit).
  bool tree_pending = false;
  auto take_tree = [&]() {
    if (!tree_pending) return;
    size_t nn = 0;
    QR(qr_tree_nodes(ctx_, nodes.data(), &nn));                            // waits for the tree only
    ensemble_model_.push(tree_from_records(nodes.data(), 0), shrinkage_);  // mart.cc:342
    tree_pending = false;
  };
  for (size_t m = first; m < ntrees_; ++m) {
    if (validation && (valid_iterations_ && m > best_model_ + valid_iterations_)) break;
    if (lambda)
      QR(qr_lambda_compute(ctx_, mcode, cutoff));  // lambdamart.cc:62-152
    else
      QR(qr_residual_compute(ctx_));               // mart.cc:418-431
    take_tree();
    if (obliv)
      QR(qr_oblivious_fit(ctx_, treedepth_, minleafsupport_, lambda, nullptr, nullptr));
    else
      QR(qr_tree_fit(ctx_, nleaves_, minleafsupport_, lambda, nullptr, nullptr));
    QR(qr_scores_update(ctx_, shrinkage_));                                // mart.cc:345, :356
    if (lagged && m > first) {
      MetricScore prev = 0;
      QR(qr_metric_last(ctx_, &prev));  // waits for this iteration's lambda pass only
      report(m, prev, nullptr);
    }
    tree_pending = true;
    if (!lagged) {
      take_tree();
      MetricScore metric_on_training = 0, metric_on_validation = 0;
      QR(qr_metric_eval(ctx_, 0, mcode, cutoff, &metric_on_training));     // mart.cc:347
      if (validation)
        QR(qr_metric_eval(ctx_, 1, mcode, cutoff, &metric_on_validation)); // mart.cc:359
      report(m + 1, metric_on_training, validation ? &metric_on_validation : nullptr);
    }
    if (partial_save != 0 && !output_basename.empty() && (m + 1) % partial_save == 0) {
      take_tree();
      save(output_basename, (int)(m + 1));
    }
  }
  take_tree();
  if (lagged && ensemble_model_.get_size() > first) {
    MetricScore last = 0;
    QR(qr_metric_eval(ctx_, 0, mcode, cutoff, &last));
    report(ensemble_model_.get_size(), last, nullptr);
  }
  // rollback to the best model observed on the validation data (mart.cc:390-395)
  if (validation)
    while (ensemble_model_.is_notempty() && ensemble_model_.get_size() > best_model_ + 1)
      ensemble_model_.pop();
  auto t_train1 = std::chrono::high_resolution_clock::now();
  std::cout << std::endl;
  std::cout << metric << "@" << cutoff << " on training data = " << best_metric_on_training_ << std::endl;
  if (validation)
    std::cout << metric << "@" << cutoff << " on validation data = " << best_metric_on_validation_
              << std::endl;
  std::cout << std::endl;
  std::cout << "#\t Training Time: " << std::setprecision(2)
            << std::chrono::duration<double>(t_train1 - t_train0).count() << " s." << std::endl;
}

void Mart::score_dataset(const data::Dataset &dataset, Score *scores, float *kernel_ms) {
  ensure_ctx();
  std::vector<qr_node_t> nodes;
  std::vector<double> weights;
  const size_t max_nodes = ensemble_model_.flatten(&nodes, &weights);
  if (weights.empty()) {
    for (size_t i = 0; i < dataset.num_instances(); ++i) scores[i] = 0.0;
    return;
  }
  QR(qr_ensemble_upload(ctx_, nodes.data(), weights.size(), max_nodes, weights.data()));
  QR(qr_ensemble_score(ctx_, dataset.at(0, 0), dataset.num_instances(), dataset.num_features(), scores,
                       kernel_ms));
}

std::shared_ptr<data::Dataset> Mart::partial_scores(const data::Dataset &dataset, bool ignore_weights) {
  ensure_ctx();
  std::vector<qr_node_t> nodes;
  std::vector<double> weights;
  const size_t max_nodes = ensemble_model_.flatten(&nodes, &weights);
  const size_t T = weights.size(), N = dataset.num_instances();
  if (T == 0) {
    std::cerr << "# ## ERROR!! Only Ensemble methods support the export of detailed score tree by tree"
              << std::endl;  // driver.cc:426-430
    exit(EXIT_FAILURE);
  }
  QR(qr_ensemble_upload(ctx_, nodes.data(), T, max_nodes, weights.data()));
  std::vector<double> part(N * T);
  QR(qr_ensemble_partial_scores(ctx_, dataset.at(0, 0), N, dataset.num_features(), ignore_weights ? 1 : 0,
                                part.data()));
  auto out = std::make_shared<data::Dataset>(N, T);
  std::vector<Feature> row(T);
  for (size_t q = 0; q < dataset.num_queries(); ++q)
    for (size_t i = dataset.offset(q); i < dataset.offset(q + 1); ++i) {
      for (size_t t = 0; t < T; ++t) row[t] = (Feature)part[i * T + t];  // Score -> Feature (driver.cc:436-438)
      out->addInstance((QueryID)q, dataset.getLabel(i), row);
    }
  return out;
}

MetricScore Mart::evaluate(const data::Dataset &dataset, const Score *scores, const std::string &metric,
                           size_t cutoff) {
  // a scratch context: the test set is not the training set of ctx_
  qr_ctx *c = nullptr;
  if (qr_ctx_create(0, &c) != QR_OK) {
    std::cerr << "!!! " << qr_last_error(nullptr) << std::endl;
    exit(EXIT_FAILURE);
  }
  MetricScore out = 0;
  if (qr_dataset_upload(c, dataset.at(0, 0), dataset.num_instances(), dataset.num_features(),
                        dataset.labels(), dataset.offsets().data(), dataset.num_queries()) ||
      qr_scores_set(c, scores) || qr_metric_eval(c, 0, metric_code(metric), cutoff, &out))
    qr_die(c, "evaluate");
  qr_ctx_destroy(c);
  return out;
}

std::unique_ptr<xml::Node> Mart::get_xml_model() const {
  std::unique_ptr<xml::Node> root(new xml::Node());
  root->name = "ranker";
  xml::Node *info = root->append_child("info");
  info->append_child("type")->text = name();
  info->append_child("trees")->text = std::to_string(ntrees_);
  info->append_child("leaves")->text = std::to_string(nleaves_);
  if (algo_ >= OBVMART) info->append_child("depth")->text = std::to_string(treedepth_);
  info->append_child("shrinkage")->text = xml::fmt_double(shrinkage_);
  info->append_child("leafsupport")->text = std::to_string(minleafsupport_);
  info->append_child("discretization")->text = std::to_string(nthresholds_);
  if (algo_ >= OBVMART) {
    // obliviousmart.cc:81 writes nthresholds_ into <estop>; reproduced as is
    info->append_child("estop")->text = std::to_string(nthresholds_);
  } else {
    info->append_child("estop")->text = std::to_string(valid_iterations_);
    info->append_child("subsample")->text = xml::fmt_float(subsample_);
    info->append_child("max_features")->text = xml::fmt_float(max_features_);
    info->append_child("collapse_leaves_factor")->text = xml::fmt_float(collapse_leaves_factor_);
  }
  ensemble_model_.append_xml_model(root.get());
  return root;
}

void Mart::save(const std::string &output_basename, int iteration) const {
  if (output_basename.empty()) return;
  std::string filename(output_basename);
  if (iteration != -1) filename += ".T" + std::to_string(iteration) + ".xml";
  auto doc = get_xml_model();
  xml::save_file(*doc, filename);
  xml::release(std::move(doc));
}

std::shared_ptr<Mart> Mart::load_model_from_file(const std::string &model_filename) {
  if (model_filename.empty()) {
    std::cerr << "!!! Model filename is empty." << std::endl;
    exit(EXIT_FAILURE);
  }
  auto doc = xml::load_file(model_filename);
  if (!doc || doc->name != "ranker") {
    std::cerr << "!!! Model " + model_filename + " is not parsed correctly." << std::endl;
    exit(EXIT_FAILURE);
  }
  Algo a;
  const xml::Node *info = doc->child("info");
  if (!info || !algo_from_name(info->child_text("type"), &a)) return nullptr;  // ltr_algorithm.cc:123
  std::shared_ptr<Mart> m(new Mart(*doc));
  xml::release(std::move(doc));
  return m;
}

bool Mart::import_model_state(Mart &other) {
  if (std::abs(shrinkage_ - other.shrinkage_) > 0.000001 || nthresholds_ != other.nthresholds_ ||
      nleaves_ != other.nleaves_ || minleafsupport_ != other.minleafsupport_ ||
      valid_iterations_ != other.valid_iterations_)
    return false;
  if (algo_ >= OBVMART && treedepth_ != other.treedepth_) return false;
  ensemble_model_ = std::move(other.ensemble_model_);
  ensemble_model_.set_capacity(ntrees_);
  return true;
}

}  // namespace forests
}  // namespace learning
}  // namespace quickrank
