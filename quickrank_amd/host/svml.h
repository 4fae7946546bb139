// svml.h -- SVMLight / LETOR text format: parallel reader, writer.
// Same grammar, exit statuses and public method names as the reference's io::Svml
// (src/io/svml.cc:38-188); see svml.cc for what a line means.
#pragma once
#include <memory>
#include <string>

#include "dataset.h"

namespace quickrank {
namespace io {

// strtof's result and end pointer, by a shorter way where that is certain to agree (svml.cc)
float parse_float(const char *s, char **after);

// One score per line, "%.17g" -- what `os << setprecision(max_digits10) << score << endl` prints
// (driver.cc:376-383, quickscore.cc:122-130) -- formatted on all host threads, written once.
bool write_scores(const std::string &path, const double *scores, size_t n);

class Svml {
  // seconds spent parsing the text / filling the dense matrix, bytes of the file
  struct Stats {
    double parse_s = 0.0, fill_s = 0.0;
    long bytes = 0;
  } stats_;

 public:
  // Malformed input ends the program like the reference does: status 1 (a second
  // token that is not "qid:..."), 2 (no label: blank lines included), 3 (negative
  // qid), 4 (a feature token that is not "id:value").
  std::unique_ptr<data::Dataset> read_horizontal(const std::string &path);

  // one line per document: "<label> qid:<q> 1:<v1> 2:<v2> ..." (every feature, %.9f)
  void write(const data::Dataset &dataset, const std::string &path);

  double reading_time() const { return stats_.parse_s; }
  double processing_time() const { return stats_.fill_s; }
  long file_size() const { return stats_.bytes; }
};

}  // namespace io
}  // namespace quickrank
