#include "svml.h"

#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>
#include <list>

namespace quickrank {
namespace io {

namespace {
inline bool isspc(char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\v' || ch == '\f' || ch == '\r'; }

// strutils.cc:36-47: skip spaces, return the token closed at the next space; if
// `exitch` is the first non-space the function returns immediately.
char *read_token(char *&str, const char exitch = '\0') {
  while (isspc(*str) && *str != '\0') ++str;
  if (*str == exitch) return str;
  char *token = str;
  while (!isspc(*str) && *str != '\0' && *str != exitch) ++str;
  if (*str != '\0') *str++ = '\0';
  return token;
}

// strutils.cc:63-74
unsigned int atou(char *str, const char *sep) {
  while (isspc(*str) && *str != '\0') ++str;
  for (size_t i = 0; sep[i] != '\0' && *str != '\0'; ++i, ++str)
    if (*str != sep[i]) exit(1);
  int x = atoi(str);
  if (x < 0) exit(3);
  return (unsigned int)x;
}
}  // namespace

std::unique_ptr<data::Dataset> Svml::read_horizontal(const std::string &filename) {
  FILE *f = fopen(filename.c_str(), "r");
  if (!f) {
    std::cerr << "!!! Error while opening file " << filename << "." << std::endl;
    exit(EXIT_FAILURE);
  }
  struct stat filestatus;
  stat(filename.c_str(), &filestatus);
  file_size_ = filestatus.st_size;
  auto t0 = std::chrono::high_resolution_clock::now();

  size_t maxfid = 0;
  std::list<size_t> data_qids;
  std::list<Label> data_labels;
  std::list<std::vector<Feature>> data_instances;

  char *line = NULL;
  size_t linelength = 0;
  while (!feof(f)) {
    ssize_t nread = getline(&line, &linelength, f);
    if (nread <= 0) continue;
    char *token = NULL, *pch = line;
    while (isspc(*pch) && *pch != '\0') ++pch;
    if (*pch == '#') continue;  // comment line
    if (*(token = read_token(pch)) == '\0') exit(2);  // label is mandatory (ISEMPTY, strutils.h:44)
    Label relevance = atof(token);
    size_t qid = atou(read_token(pch), "qid:");
    std::vector<Feature> curr_instance(maxfid);
    while (*(token = read_token(pch, '#')) != '\0') {
      if (*token == '#') {
        *pch = '\0';  // trailing description
      } else {
        size_t fid = 0;
        float fval = 0.0f;
        if (sscanf(token, "%zu:%f", &fid, &fval) != 2) exit(4);
        if (fid > maxfid) {
          maxfid = fid;
          curr_instance.resize(maxfid);
        }
        curr_instance[fid - 1] = fval;
      }
    }
    data_qids.push_back(qid);
    data_labels.push_back(relevance);
    data_instances.push_back(std::move(curr_instance));
  }
  free(line);
  fclose(f);
  auto t1 = std::chrono::high_resolution_clock::now();

  auto dataset = std::unique_ptr<data::Dataset>(new data::Dataset(data_qids.size(), maxfid));
  auto i_q = data_qids.begin();
  auto i_l = data_labels.begin();
  auto i_x = data_instances.begin();
  for (; i_q != data_qids.end(); ++i_q, ++i_l, ++i_x)
    dataset->addInstance((QueryID)*i_q, *i_l, *i_x);
  auto t2 = std::chrono::high_resolution_clock::now();
  reading_time_ = std::chrono::duration<double>(t1 - t0).count();
  processing_time_ = std::chrono::duration<double>(t2 - t1).count();
  return dataset;
}

// svml.cc:163-188
void Svml::write(const data::Dataset &dataset, const std::string &file) {
  std::ofstream out(file, std::ofstream::out | std::ofstream::trunc);
  for (size_t q = 0; q < dataset.num_queries(); q++) {
    for (size_t r = dataset.offset(q); r < dataset.offset(q + 1); r++) {
      out << std::setprecision(0) << dataset.getLabel(r) << " qid:" << q + 1;
      const Feature *x = dataset.at(r, 0);
      for (size_t f = 0; f < dataset.num_features(); f++)
        out << " " << f + 1 << ":" << std::fixed
            << std::setprecision(std::numeric_limits<Feature>::max_digits10) << x[f];
      out << std::endl;
    }
  }
}

}  // namespace io
}  // namespace quickrank
