// svml.h -- SVMLight / LETOR reader and writer with the reference's grammar
// (src/io/svml.cc:38-188, src/utils/strutils.cc:36-74).
#pragma once
#include <memory>
#include <string>

#include "dataset.h"

namespace quickrank {
namespace io {

class Svml {
 public:
  // Exits with the reference's codes on malformed input: 1 (missing "qid:"),
  // 2 (missing label), 3 (negative qid), 4 (malformed feature token).
  std::unique_ptr<data::Dataset> read_horizontal(const std::string &filename);
  void write(const data::Dataset &dataset, const std::string &file);
  double reading_time() const { return reading_time_; }
  double processing_time() const { return processing_time_; }
  long file_size() const { return file_size_; }

 private:
  double reading_time_ = 0, processing_time_ = 0;
  long file_size_ = 0;
};

}  // namespace io
}  // namespace quickrank
