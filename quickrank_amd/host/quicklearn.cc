// quicklearn -- command-line front end with QuickRank's flag surface for the
// algorithms this build accelerates (src/quicklearn.cc:89-507 and
// src/driver/driver.cc:45-226 of the reference): same option names, defaults
// and phase order (load -> train -> save -> test -> scores file).  Options of
// out-of-scope subsystems (DART, CLEAVER, linear rankers) are recognised and
// rejected with a message.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <set>

#include "codegen.h"
#include "mart.h"
#include "svml.h"

using namespace quickrank;
using learning::forests::Mart;

namespace {
struct Opt {
  std::string help, dflt;
  bool has_arg;
};
const std::vector<std::pair<std::string, Opt>> kOptions = {
    {"algo", {"LtR algorithm: [MART|LAMBDAMART|OBVMART|OBVLAMBDAMART].", "LAMBDAMART", true}},
    {"train-metric", {"set train metric: [DCG|NDCG].", "NDCG", true}},
    {"train-cutoff", {"set train metric cutoff.", "10", true}},
    {"partial", {"set partial file save frequency.", "100", true}},
    {"train", {"set training file.", "", true}},
    {"valid", {"set validation file.", "", true}},
    {"features", {"set features file.", "", true}},
    {"model-in", {"set input model file (for testing, re-training or optimization)", "", true}},
    {"model-out", {"set output model file", "", true}},
    {"skip-train", {"skip training phase.", "", false}},
    {"restart-train", {"restart training phase from a previous trained model.", "", false}},
    {"num-trees", {"set number of trees.", "1000", true}},
    {"shrinkage", {"set shrinkage.", "0.1", true}},
    {"num-thresholds", {"set number of thresholds.", "0", true}},
    {"min-leaf-support", {"set minimum number of leaf support.", "1", true}},
    {"end-after-rounds",
     {"set num. rounds with no gain in validation before ending (if 0 disabled).", "100", true}},
    {"num-leaves", {"set number of leaves [applies only to MART/LambdaMART].", "10", true}},
    {"tree-depth", {"set tree depth [applies only to ObliviousMART/ObliviousLambdaMART].", "3", true}},
    {"subsample",
     {"set documents per iteration: fraction if <= 1, count if > 1 [MART/LambdaMART].", "1", true}},
    {"max-features",
     {"set features per split search: fraction if <= 1, count if > 1 [MART/LambdaMART].", "1", true}},
    {"gpus", {"train on this many GPUs of the node (RCCL over xGMI) [MART/LambdaMART].", "1", true}},
    {"shard", {"what the GPUs split: [docs|features] (documents, or feature blocks of the bin matrix).", "docs", true}},
    {"seed", {"seed of the document / feature sampling (0: from the clock, like the reference).", "0", true}},
    {"test-metric", {"set test metric: [DCG|NDCG].", "NDCG", true}},
    {"test-cutoff", {"set test metric cutoff.", "10", true}},
    {"test", {"set testing file.", "", true}},
    {"scores", {"set output scores file.", "", true}},
    {"detailed", {"enable detailed testing [applies only to ensemble models].", "", false}},
    {"model-file", {"set XML model file path [code generation].", "", true}},
    {"code-file", {"set C code file path [code generation].", "", true}},
    {"generator", {"set C code generation strategy: [condop|oblivious|vpred].", "condop", true}},
};
const std::set<std::string> kOutOfScope = {
    "meta-algo", "final-num-trees", "opt-last-only", "meta-end-after-rounds", "meta-verbose",
    "sample-type", "normalize-type", "adaptive-type", "rate-drop", "skip-drop", "keep-drop",
    "best-on-train", "random-keep", "drop-on-best", "num-samples", "window-size", "reduction-factor",
    "max-iterations", "max-failed-valid", "adaptive", "train-partial", "valid-partial", "opt-algo",
    "opt-method", "opt-model", "opt-algo-model", "pruning-rate", "with-line-search",
    "line-search-model", "collapse-leaves-factor"};

void help() {
  std::cout << "quicklearn (MI355X build): LambdaMART / MART / oblivious variants on the GPU\n\n";
  for (auto &o : kOptions) {
    std::string left = "  --" + o.first + (o.second.has_arg ? " <arg>" : "");
    if (!o.second.dflt.empty()) left += " (" + o.second.dflt + ")";
    std::cout << std::left << std::setw(40) << left << o.second.help << "\n";
  }
  std::cout << "  -h,--help                             print help message.\n";
}

std::shared_ptr<data::Dataset> load_dataset(const std::string &file, const std::string &label) {
  io::Svml reader;
  std::cout << "# Reading " + label + " dataset: " << file << std::endl;
  std::shared_ptr<data::Dataset> ds = reader.read_horizontal(file);  // driver.cc:387-407
  std::cout << std::setprecision(2) << "#\t Reading time: " << reader.reading_time() << " s. @ "
            << reader.file_size() / 1024 / 1024 / std::max(reader.reading_time(), 1e-9) << " MB/s "
            << " (post-proc.: " << reader.processing_time() << " s.)" << std::endl;
  std::cout << "#\t Dataset size: " << ds->num_instances() << " x " << ds->num_features()
            << " (instances x features)" << std::endl
            << "#\t Num queries: " << ds->num_queries() << " | Avg. len: " << std::setprecision(3)
            << ds->num_instances() / (float)ds->num_queries() << std::endl;
  return ds;
}
}  // namespace

int main(int argc, char *argv[]) {
  std::cout << std::fixed;
  std::map<std::string, std::string> v;
  std::set<std::string> isset;
  for (auto &o : kOptions) v[o.first] = o.second.dflt;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "-h" || a == "--help") {
      help();
      return EXIT_SUCCESS;
    }
    if (a.rfind("--", 0) != 0) {
      std::cerr << "!!! unexpected argument " << a << std::endl;
      return EXIT_FAILURE;
    }
    a = a.substr(2);
    if (kOutOfScope.count(a)) {
      std::cerr << "!!! option --" << a << " belongs to a subsystem outside this build's scope "
                << "(see DESIGN.md section 9)." << std::endl;
      return EXIT_FAILURE;
    }
    bool found = false;
    for (auto &o : kOptions)
      if (o.first == a) {
        found = true;
        isset.insert(a);
        if (o.second.has_arg) {
          if (i + 1 >= argc) {
            std::cerr << "!!! option --" << a << " needs an argument" << std::endl;
            return EXIT_FAILURE;
          }
          v[a] = argv[++i];
        }
      }
    if (!found) {
      std::cerr << "!!! unknown option --" << a << std::endl;
      return EXIT_FAILURE;
    }
  }
  if (!isset.count("train") && !isset.count("test") && !isset.count("model-file")) {  // driver.cc:47-51
    help();
    return EXIT_FAILURE;
  }
  // code generation follows the training / test phases when both are asked for
  // (driver.cc:197-224)
  auto codegen = [&]() {
    if (isset.count("model-file") && isset.count("code-file"))
      return io::generate(v["generator"], v["model-file"], v["code-file"]);
    return (int)EXIT_SUCCESS;
  };
  if (!isset.count("train") && !isset.count("test")) return codegen();
  // ltr_algorithm_factory.cc:41-261 for the in-scope names
  std::shared_ptr<Mart> algo;
  if (isset.count("model-in") && !isset.count("restart-train") &&
      (isset.count("skip-train") || !isset.count("train"))) {
    std::cout << "# Loading model from file " << v["model-in"] << std::endl;
    algo = Mart::load_model_from_file(v["model-in"]);
  } else {
    Mart::Algo a;
    if (!Mart::algo_from_name(v["algo"], &a)) {
      std::cerr << " !! LTR Algorithm was not set properly" << std::endl;  // driver.cc:58-61
      return EXIT_FAILURE;
    }
    algo = std::make_shared<Mart>(a, std::stoul(v["num-trees"]), std::stod(v["shrinkage"]),
                                  std::stoul(v["num-thresholds"]), std::stoul(v["num-leaves"]),
                                  std::stoul(v["min-leaf-support"]), std::stoul(v["end-after-rounds"]),
                                  std::stoul(v["tree-depth"]));
    algo->set_sampling(std::stof(v["subsample"]), std::stof(v["max-features"]), std::stoull(v["seed"]));
    if (isset.count("model-in") && isset.count("restart-train")) {
      auto loaded = Mart::load_model_from_file(v["model-in"]);
      if (!loaded || !algo->import_model_state(*loaded)) {  // ltr_algorithm_factory.cc:249-258
        std::cerr << " !! Mismatch between the loaded model and the parameters of the model to train"
                  << std::endl;
        return EXIT_FAILURE;
      }
    }
  }
  if (!algo) {
    std::cerr << " !! LTR Algorithm was not set properly" << std::endl;
    return EXIT_FAILURE;
  }
  std::cout << std::endl;
  algo->print(std::cout);
  std::cout << std::endl;

  if (isset.count("train") && !isset.count("skip-train")) {
    auto training = load_dataset(v["train"], "training");
    std::shared_ptr<data::Dataset> validation;
    if (!v["valid"].empty()) validation = load_dataset(v["valid"], "validation");
    if (!v["features"].empty())  // driver.cc:108-110: the reference reads the flag and does nothing with it
      std::cout << "# --features " << v["features"] << ": accepted and not used, as in the reference "
                << "(driver.cc:108-110 is a TODO)" << std::endl;
    int gpus = 0;   // (checked parse: std::stoi throws on "--gpus two")
    {
      const std::string &g = v["gpus"];
      char *endp = nullptr;
      const long gl = std::strtol(g.c_str(), &endp, 10);
      if (!g.empty() && endp && *endp == '\0' && gl >= 1 && gl <= 1024) gpus = (int)gl;
    }
    if (gpus < 1 || (v["shard"] != "docs" && v["shard"] != "features")) {
      std::cerr << "!!! --gpus needs a positive count, --shard docs or features" << std::endl;
      return EXIT_FAILURE;
    }
    if (gpus > 1 || isset.count("shard"))  // (--shard with one GPU runs the sharded protocol on one rank)
      algo->learn_multi(training, validation, v["train-metric"], std::stoul(v["train-cutoff"]),
                        std::stoul(v["partial"]), v["model-out"], gpus, v["shard"] == "features");
    else
      algo->learn(training, validation, v["train-metric"], std::stoul(v["train-cutoff"]),
                  std::stoul(v["partial"]), v["model-out"]);  // driver.cc:228-246
    if (!v["model-out"].empty()) {
      std::cout << std::endl << "# Writing model to file: " << v["model-out"] << std::endl << std::endl;
      algo->save(v["model-out"]);
    }
  }
  if (isset.count("test")) {  // driver.cc:326-385
    auto test = load_dataset(v["test"], "test");
    std::vector<Score> scores(test->num_instances(), 0.0);
    const size_t k = std::stoul(v["test-cutoff"]);
    if (isset.count("detailed")) {  // driver.cc:335-358: tree-by-tree scores as an SVMLight file
      auto part = algo->partial_scores(*test);
      const size_t T = part->num_features();
      for (size_t i = 0; i < part->num_instances(); ++i) {
        const Feature *row = part->at(i, 0);
        for (size_t t = 0; t < T; ++t) scores[i] += row[t];  // f32 per-tree scores summed in f64 (driver.cc:345-347)
      }
      MetricScore s = algo->evaluate(*test, scores.data(), v["test-metric"], k);
      std::cout << v["test-metric"] << "@" << k << " on test data = " << std::setprecision(4) << s << std::endl
                << std::endl;
      io::Svml().write(*part, v["scores"]);
      std::cout << "# Partial Scores written to file: " << v["scores"] << std::endl;
      return codegen();
    }
    algo->score_dataset(*test, scores.data());
    MetricScore s = algo->evaluate(*test, scores.data(), v["test-metric"], k);
    std::cout << std::endl << v["test-metric"] << "@" << k << " on test data = " << std::setprecision(4)
              << s << std::endl << std::endl;
    if (!v["scores"].empty()) {
      if (!io::write_scores(v["scores"], scores.data(), test->num_instances())) {
        std::cerr << "!!! Error while opening file " << v["scores"] << "." << std::endl;
        exit(EXIT_FAILURE);
      }
      std::cout << "# Scores written to file: " << v["scores"] << std::endl;
    }
  }
  return codegen();
}
