// xml.h -- the small XML subset QuickRank's model files use (elements,
// attributes, text), written the way pugixml's save_file(path, "\t",
// format_default | format_no_declaration) writes it (ltr_algorithm.cc:54-66):
// tab indentation, one element per line, text-only elements on one line, no XML
// declaration.  pugixml itself is an un-vendored submodule of the reference
// (parity of its number formatting is unpinned: we use "%.17g" for doubles and
// "%.9g" for floats, what current pugixml releases do).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace quickrank {
namespace xml {

struct Node {
  std::string name;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::string text;
  std::vector<std::unique_ptr<Node>> children;

  Node *append_child(const std::string &n) {
    children.emplace_back(new Node());
    children.back()->name = n;
    return children.back().get();
  }
  void append_attribute(const std::string &k, const std::string &v) { attrs.emplace_back(k, v); }
  const Node *child(const std::string &n) const {
    for (auto &c : children)
      if (c->name == n) return c.get();
    return nullptr;
  }
  std::string attribute(const std::string &k) const {
    for (auto &a : attrs)
      if (a.first == k) return a.second;
    return "";
  }
  std::string child_text(const std::string &n, const std::string &dflt = "") const {
    const Node *c = child(n);
    return c ? c->text : dflt;
  }
};

std::string to_string(const Node &root);                 // serialised document
bool save_file(const Node &root, const std::string &path);
std::unique_ptr<Node> parse(const std::string &doc);      // nullptr on syntax error
std::unique_ptr<Node> load_file(const std::string &path);
void release(std::unique_ptr<Node> doc);                  // frees a large document on all host threads

std::string fmt_double(double v);   // "%.17g"
std::string fmt_float(float v);     // "%.9g"

}  // namespace xml
}  // namespace quickrank
