// quickscore -- times ensemble scoring like src/quickscore.cc:64-134 of the
// reference, with one difference forced by the device: the reference links a
// ranker() generated from the XML model and recompiled (documentation/
// quickscore.md:14-23); here the XML model is loaded at run time (-m) and scored
// by the GPU kernel.  Flags -d/-r/-s keep their meaning.
#include <chrono>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>

#include "mart.h"
#include "svml.h"

using namespace quickrank;

int main(int argc, char *argv[]) {
  std::string dataset_file, model_file, scores_file;
  unsigned rounds = 10;  // quickscore.cc:70
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> std::string {
      if (i + 1 >= argc) {
        std::cerr << "!!! option " << a << " needs an argument" << std::endl;
        exit(EXIT_FAILURE);
      }
      return argv[++i];
    };
    if (a == "-d" || a == "--dataset") dataset_file = next();
    else if (a == "-m" || a == "--model") model_file = next();
    else if (a == "-r" || a == "--rounds") rounds = (unsigned)std::stoul(next());
    else if (a == "-s" || a == "--scores") scores_file = next();
    else {
      std::cout << "quickscore -d <dataset> -m <model.xml> [-r <rounds> (10)] [-s <scores file>]\n";
      return a == "-h" || a == "--help" ? EXIT_SUCCESS : EXIT_FAILURE;
    }
  }
  if (dataset_file.empty() || model_file.empty()) {
    std::cout << "quickscore -d <dataset> -m <model.xml> [-r <rounds> (10)] [-s <scores file>]\n";
    return EXIT_FAILURE;
  }
  auto model = learning::forests::Mart::load_model_from_file(model_file);
  if (!model) {
    std::cerr << "!!! unsupported model type in " << model_file << std::endl;
    return EXIT_FAILURE;
  }
  io::Svml reader;
  std::cout << "# Reading test dataset: " << dataset_file << std::endl;
  std::shared_ptr<data::Dataset> ds = reader.read_horizontal(dataset_file);
  std::cout << "#\t Dataset size: " << ds->num_instances() << " x " << ds->num_features()
            << " (instances x features)" << std::endl
            << "#\t Num queries: " << ds->num_queries() << std::endl;
  std::vector<Score> scores(ds->num_instances(), 0.0);
  double kernel_total = 0.0;
  auto t0 = std::chrono::high_resolution_clock::now();
  for (unsigned r = 0; r < rounds; ++r) {
    float ms = 0;
    model->score_dataset(*ds, scores.data(), &ms);
    kernel_total += ms * 1e-3;
  }
  auto t1 = std::chrono::high_resolution_clock::now();
  const double total = std::chrono::duration<double>(t1 - t0).count();
  // quickscore.cc:112-120
  std::cout << "       Total scoring time: " << total << " s." << std::endl
            << "Avg. Dataset scoring time: " << total / rounds << " s." << std::endl
            << "Avg.    Doc. scoring time: " << total / ds->num_instances() / rounds << " s." << std::endl
            << "   of which device kernel: " << kernel_total / rounds << " s. per round (features resident)"
            << std::endl;
  if (!scores_file.empty()) {
    if (!io::write_scores(scores_file, scores.data(), ds->num_instances())) {
      std::cerr << "!!! Error while opening file " << scores_file << "." << std::endl;
      return EXIT_FAILURE;
    }
  }
  return EXIT_SUCCESS;
}
