// codegen.cc -- see codegen.h.  The generators work on a flattened copy of the
// model (one record per <split> element, numbers kept as the text the file holds)
// instead of walking the XML document while printing.
#include "codegen.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <iostream>
#include <numeric>
#include <sstream>
#include <vector>

#include "xml.h"

namespace quickrank {
namespace io {

namespace {

// One <split> element.  Field rules of the reference's readers: children are
// visited in document order; an <output> makes the element a leaf and ends the
// visit; of repeated <feature> / <threshold> / <split pos=...> the last one counts.
struct Split {
  bool leaf = false;
  std::string feature, threshold, output;
  int left = -1, right = -1;        // indices into Model::splits
  std::vector<int> kids;            // every <split> child in document order, with ...
  std::vector<bool> kid_is_left;    // ... whether its pos attribute says "left"
};

struct Model {
  std::vector<Split> splits;
  struct Tree {
    std::string weight;  // attribute text
    int root = -1;       // -1: <tree> without a <split>
  };
  std::vector<Tree> trees;
  std::string shrinkage, depth;
  bool parsed = false;
};

int flatten(const xml::Node &el, Model &m) {
  const int me = (int)m.splits.size();
  m.splits.emplace_back();
  for (const auto &c : el.children) {
    if (c->name == "output") {
      m.splits[me].output = c->text;
      m.splits[me].leaf = true;
      break;
    }
    if (c->name == "feature") m.splits[me].feature = c->text;
    if (c->name == "threshold") m.splits[me].threshold = c->text;
    if (c->name == "split") {
      const int k = flatten(*c, m);
      const std::string pos = c->attribute("pos");
      m.splits[me].kids.push_back(k);
      m.splits[me].kid_is_left.push_back(pos == "left");
      if (pos == "left") m.splits[me].left = k;
      if (pos == "right") m.splits[me].right = k;
    }
  }
  return me;
}

Model read_model(const std::string &doc) {
  Model m;
  auto root = xml::parse(doc);
  if (!root) return m;
  m.parsed = true;
  if (root->name != "ranker") return m;
  if (const xml::Node *info = root->child("info")) {
    m.shrinkage = info->child_text("shrinkage");
    m.depth = info->child_text("depth");
  }
  const xml::Node *ens = root->child("ensemble");
  if (!ens) return m;
  for (const auto &t : ens->children) {
    if (t->name != "tree") continue;
    Model::Tree tr;
    tr.weight = t->attribute("weight");
    if (const xml::Node *s = t->child("split")) tr.root = flatten(*s, m);
    m.trees.push_back(tr);
  }
  return m;
}

unsigned to_uint(const std::string &s) { return (unsigned)strtoul(s.c_str(), nullptr, 10); }
float to_float(const std::string &s) { return (float)strtod(s.c_str(), nullptr); }

std::string fmt(const char *f, double v) {
  char b[400];
  snprintf(b, sizeof b, f, v);
  return b;
}

// ---------------------------------------------------------------- condop
void condop_expr(const Model &m, int at, std::string &out) {
  if (at < 0) return;  // a missing child prints nothing, like an empty XML node
  const Split &s = m.splits[at];
  if (s.leaf) {
    out += s.output;
    return;
  }
  // v[] is indexed from 0, the model's feature ids from 1; an integer-looking
  // threshold gets ".0" so that the "f" suffix makes a valid float literal
  std::string thr = s.threshold;
  if (thr.find('.') == std::string::npos) thr += ".0";
  out += "( v[" + std::to_string(to_uint(s.feature) - 1u) + "] <= " + thr + "f ? ";
  condop_expr(m, s.left, out);
  out += " : ";
  condop_expr(m, s.right, out);
  out += " )";
}

// ---------------------------------------------------------------- oblivious
// Oblivious trees are symmetric: the leftmost path holds every level's test.
void leftmost_tests(const Model &m, int at, std::vector<unsigned> &fids,
                    std::vector<std::string> &thrs) {
  while (at >= 0 && !m.splits[at].leaf) {
    const Split &s = m.splits[at];
    // (a level whose <feature>/<threshold> is missing contributes nothing)
    if (!s.feature.empty()) fids.push_back(to_uint(s.feature) - 1u);
    if (!s.threshold.empty()) thrs.push_back(s.threshold);
    at = s.left;
  }
}

bool leaves_dfs(const Model &m, int at, std::vector<std::string> &out) {
  if (at < 0) return false;  // malformed: the reference recurses forever here
  const Split &s = m.splits[at];
  if (s.leaf) {
    out.push_back(s.output);
    return true;
  }
  return leaves_dfs(m, s.left, out) && leaves_dfs(m, s.right, out);
}

// Levels counted the way generate_oblivious.cc:170-182 counts them: one for the
// root <split>, one more for every step down the FIRST <split> child whose own
// first <split> child still has <split> children.
int counted_depth(const Model &m, int root) {
  if (root < 0) return 0;
  int d = 0, at = root;
  for (;;) {
    ++d;
    const Split &s = m.splits[at];
    if (s.kids.empty()) break;
    const Split &first = m.splits[s.kids[0]];
    if (first.kids.empty()) break;
    at = s.kids[0];
  }
  return d;
}

template <class T, class P>
std::string table(const std::vector<std::vector<T>> &rows, const std::vector<size_t> &order, P print) {
  std::string o = "\t";
  for (size_t i = 0; i < rows.size(); ++i) {
    if (i) o += ",\n\t";
    o += "\t{ ";
    const auto &r = rows[order[i]];
    for (size_t j = 0; j < r.size(); ++j) {
      if (j) o += ", ";
      o += print(r[j]);
    }
    o += " }";
  }
  return o + "\n};\n\n";
}

}  // namespace

std::string condop_code(const std::string &model_xml) {
  const Model m = read_model(model_xml);
  std::string o = "double ranker(float* v) {\n\treturn 0.0 ";
  for (const auto &t : m.trees) {
    if (t.root < 0) continue;
    // the weight goes through a float and is printed with 3 decimals
    o += "\n\t\t + " + fmt("%.3f", (double)to_float(t.weight)) + "f * ";
    condop_expr(m, t.root, o);
  }
  return o + ";\n}\n";
}

std::string oblivious_code(const std::string &model_xml, bool *ok) {
  const Model m = read_model(model_xml);
  *ok = false;
  const size_t n = m.trees.size();
  if (n == 0) return "!!! The model holds no trees.";
  const unsigned depth = to_uint(m.depth);
  std::vector<int> depths(n);
  std::vector<std::vector<std::string>> outputs(n), thrs(n);
  std::vector<std::vector<unsigned>> fids(n);
  for (size_t i = 0; i < n; ++i) {
    depths[i] = counted_depth(m, m.trees[i].root);
    if (!leaves_dfs(m, m.trees[i].root, outputs[i])) return "!!! Tree without both children.";
    leftmost_tests(m, m.trees[i].root, fids[i], thrs[i]);
  }
  // shallow trees first (the scorer below runs one loop per depth)
  std::vector<size_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&depths](int a, int b) { return depths[a] < depths[b]; });
  const int max_depth = depths[order.back()];
  std::vector<size_t> per_depth;
  {
    int d = 1;
    size_t from = 0;
    for (size_t i = 0; i < n; ++i) {
      while (depths[order[i]] > d) {
        per_depth.push_back(i - from);
        ++d;
        from = i;
      }
      if (d == max_depth) break;
    }
    per_depth.push_back(n - from);
  }
  std::string o;
  o += "#define N " + std::to_string(n) + " // no. of trees\n";
  o += "#define M " + std::to_string(depth) + " // max tree depth\n";
  o += "#define F " + std::to_string(1u << depth) + " // max number of leaves\n\n";
  o += "const float tree_weights[N] = { ";
  for (size_t i = 0; i < n; ++i) {
    if (i) o += ", ";
    o += fmt("%.9f", (double)to_float(m.trees[order[i]].weight));
  }
  o += " };\n\n";
  auto text = [](const std::string &s) { return s; };
  o += "const double leaf_outputs[N][F] = { \n" + table(outputs, order, text);
  o += "const unsigned int features_ids[N][M] = { \n" +
       table(fids, order, [](unsigned v) { return std::to_string(v); });
  o += "const float thresholds[N][M] = { \n" + table(thrs, order, text);
  o += "#define SHL(n,p) ((n)<<(p))\n\n";
  o += "unsigned int leaf_id(float *v, unsigned int const *fids, float const *thresh, const unsigned int m) {\n"
       "  unsigned int leafidx=0;\n"
       "  for (unsigned int i=0; i<m; ++i)\n"
       "    leafidx |= SHL( v[fids[i]]>thresh[i], m-1-i);\n"
       "  return leafidx;\n"
       "}\n\n";
  o += "double ranker(float *v) {\n  double score = 0.0;\n  int i = 0;\n";
  for (int d = 0; d < max_depth; ++d) {
    o += "  for (int j = 0; j < " + std::to_string(per_depth[d]) + "; ++j) {\n";
    o += "    score += tree_weights[i] * leaf_outputs[i][leaf_id(v, features_ids[i], thresholds[i], " +
         std::to_string(d + 1) + ")];\n";
    o += "    i++;\n  }\n";
  }
  o += "  return score;\n}\n";
  *ok = true;
  return o;
}

namespace {
// longest root-to-leaf path counted in elements, a leaf being 1
// (generate_vpred.cc:48-64)
unsigned vpred_depth(const Model &m, int at) {
  const Split &s = m.splits[at];
  if (s.leaf) return 1;
  unsigned l = 0, r = 0;
  for (size_t k = 0; k < s.kids.size(); ++k) {
    const unsigned d = 1 + vpred_depth(m, s.kids[k]);
    if (s.kid_is_left[k]) l = d; else r = d;
  }
  return std::max(l, r);
}
}  // namespace

std::string vpred_text(const std::string &model_xml, bool *ok) {
  const Model m = read_model(model_xml);
  *ok = false;
  if (!m.parsed) return "!!! Model filename is not parsed correctly.";
  const double rate = strtod(m.shrinkage.c_str(), nullptr);
  std::string o = std::to_string(m.trees.size()) + "\n";
  struct Item {
    int at;
    uint32_t id, pid;
    bool left;
    std::string parent_feature;
  };
  for (const auto &t : m.trees) {
    if (t.root < 0) return "!!! Tree without a <split> element.";
    const uint32_t depth = vpred_depth(m, t.root) - 1;
    o += std::to_string(depth) + "\n";
    const uint32_t inner = (uint32_t)(std::pow(2, depth) - 1);  // ids below this are inner slots
    uint32_t next_id = 0;
    std::deque<Item> q;
    q.push_back({t.root, next_id++, (uint32_t)-1, false, ""});
    for (; !q.empty(); q.pop_front()) {
      const Item it = q.front();
      const Split &s = m.splits[it.at];
      const std::string head = std::to_string(it.id) + " " + std::to_string(it.pid) + " ";
      if (s.leaf) {
        const std::string value = fmt("%g", rate * strtod(s.output.c_str(), nullptr));
        if (it.id >= inner)
          o += "leaf " + head + (it.left ? "1 " : "0 ") + value + "\n";
        else  // a leaf above the last level takes an inner slot, with its parent's feature
          o += "node " + head + std::to_string(atoi(it.parent_feature.c_str()) - 1) + " " +
               (it.left ? "1 " : "0 ") + value + "\n";
        continue;
      }
      const std::string f0 = std::to_string(atoi(s.feature.c_str()) - 1);
      if (it.id == 0)
        o += "root 0 " + f0 + " " + s.threshold + "\n";
      else
        o += "node " + head + f0 + " " + (it.left ? "1 " : "0 ") + s.threshold + "\n";
      for (size_t k = 0; k < s.kids.size(); ++k)
        q.push_back({s.kids[k], next_id++, it.id, (bool)s.kid_is_left[k], s.feature});
    }
    o += "end\n";
  }
  *ok = true;
  return o;
}

int generate(const std::string &generator, const std::string &model_file,
             const std::string &code_file) {
  if (generator != "condop" && generator != "oblivious" && generator != "vpred")
    return EXIT_SUCCESS;  // driver.cc:204-223: an unknown strategy is silently skipped
  if (generator == "condop")
    std::cout << "applying conditional operators strategy for C code generation to: ";
  else if (generator == "oblivious")
    std::cout << "applying oblivious strategy for C code generation to: ";
  else
    std::cout << "generating VPred input file from: ";
  std::cout << model_file << std::endl;
  if (model_file.empty()) {
    std::cerr << "!!! Model filename is empty." << std::endl;
    return EXIT_FAILURE;
  }
  std::string doc;
  {
    std::ifstream f(model_file);
    std::stringstream ss;
    if (f) ss << f.rdbuf();
    doc = ss.str();  // unreadable file: an empty document, as for the reference
  }
  bool ok = true;
  std::string text;
  if (generator == "condop")
    text = condop_code(doc);
  else if (generator == "oblivious")
    text = oblivious_code(doc, &ok);
  else
    text = vpred_text(doc, &ok);
  if (!ok) {
    std::cerr << text << std::endl;
    return EXIT_FAILURE;
  }
  std::ofstream out(code_file, std::ofstream::out);
  out << text;
  return EXIT_SUCCESS;
}

}  // namespace io
}  // namespace quickrank
