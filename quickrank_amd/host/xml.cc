#include "xml.h"

#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace quickrank {
namespace xml {

std::string fmt_double(double v) {
  char buf[64];
  snprintf(buf, sizeof buf, "%.17g", v);
  return buf;
}
std::string fmt_float(float v) {
  char buf[64];
  snprintf(buf, sizeof buf, "%.9g", (double)v);
  return buf;
}

static std::string escape(const std::string &s, bool attr) {
  std::string o;
  for (char c : s) {
    if (c == '&') o += "&amp;";
    else if (c == '<') o += "&lt;";
    else if (c == '>') o += "&gt;";
    else if (c == '"' && attr) o += "&quot;";
    else o += c;
  }
  return o;
}

static void write_node(const Node &n, int depth, std::string &out) {
  out.append((size_t)depth, '\t');
  out += '<';
  out += n.name;
  for (auto &a : n.attrs) {
    out += ' ';
    out += a.first;
    out += "=\"";
    out += escape(a.second, true);
    out += '"';
  }
  if (n.children.empty() && n.text.empty()) {
    out += " />\n";
    return;
  }
  out += ">";
  if (n.children.empty()) {  // text-only element stays on one line
    out += escape(n.text, false);
    out += "</";
    out += n.name;
    out += ">\n";
    return;
  }
  out += "\n";
  if (n.children.size() >= 64) {  // a model's <ensemble>: the children side by side, joined in order
    const int nt = std::max(1, std::min<int>(omp_get_max_threads(), (int)(n.children.size() / 16)));
    std::vector<std::string> part((size_t)nt);
#pragma omp parallel for num_threads(nt) schedule(static, 1)
    for (int t = 0; t < nt; ++t) {
      const size_t i0 = n.children.size() * (size_t)t / (size_t)nt, i1 = n.children.size() * (size_t)(t + 1) / (size_t)nt;
      for (size_t i = i0; i < i1; ++i) write_node(*n.children[i], depth + 1, part[(size_t)t]);
    }
    size_t total = out.size();
    for (auto &p : part) total += p.size();
    out.reserve(total + (size_t)depth + n.name.size() + 8);
    for (auto &p : part) out += p;
  } else {
    for (auto &c : n.children) write_node(*c, depth + 1, out);
  }
  out.append((size_t)depth, '\t');
  out += "</";
  out += n.name;
  out += ">\n";
}

std::string to_string(const Node &root) {
  std::string out;
  write_node(root, 0, out);
  return out;
}

bool save_file(const Node &root, const std::string &path) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const std::string doc = to_string(root);
  const bool ok = fwrite(doc.data(), 1, doc.size(), f) == doc.size();
  return (fclose(f) == 0) && ok;
}

namespace {
struct Parser {
  const std::string &s;
  size_t p = 0;
  explicit Parser(const std::string &str) : s(str) {}
  void skip_ws() { while (p < s.size() && isspace((unsigned char)s[p])) ++p; }
  bool starts(const char *t) const { return s.compare(p, strlen(t), t) == 0; }
  static std::string unescape(const std::string &t) {
    std::string o;
    for (size_t i = 0; i < t.size(); ++i) {
      if (t[i] == '&') {
        if (t.compare(i, 5, "&amp;") == 0) { o += '&'; i += 4; continue; }
        if (t.compare(i, 4, "&lt;") == 0) { o += '<'; i += 3; continue; }
        if (t.compare(i, 4, "&gt;") == 0) { o += '>'; i += 3; continue; }
        if (t.compare(i, 6, "&quot;") == 0) { o += '"'; i += 5; continue; }
        if (t.compare(i, 6, "&apos;") == 0) { o += '\''; i += 5; continue; }
      }
      o += t[i];
    }
    return o;
  }
  void skip_misc() {  // declaration, comments, doctype
    for (;;) {
      skip_ws();
      if (starts("<?")) { size_t e = s.find("?>", p); if (e == std::string::npos) { p = s.size(); return; } p = e + 2; }
      else if (starts("<!--")) { size_t e = s.find("-->", p); if (e == std::string::npos) { p = s.size(); return; } p = e + 3; }
      else if (starts("<!")) { size_t e = s.find('>', p); if (e == std::string::npos) { p = s.size(); return; } p = e + 1; }
      else return;
    }
  }
  void skip_ws_comments() {  // what element() itself skips between an element's children
    for (;;) {
      skip_ws();
      if (!starts("<!--")) return;
      size_t e = s.find("-->", p);
      if (e == std::string::npos) { p = s.size(); return; }
      p = e + 3;
    }
  }
  std::string name() {
    size_t b = p;
    while (p < s.size() && !isspace((unsigned char)s[p]) && s[p] != '>' && s[p] != '/' && s[p] != '=') ++p;
    return s.substr(b, p - b);
  }
  std::unique_ptr<Node> element() {
    if (p >= s.size() || s[p] != '<') return nullptr;
    ++p;
    std::unique_ptr<Node> n(new Node());
    n->name = name();
    if (n->name.empty()) return nullptr;
    for (;;) {
      skip_ws();
      if (p >= s.size()) return nullptr;
      if (s[p] == '/') {
        if (p + 1 < s.size() && s[p + 1] == '>') { p += 2; return n; }
        return nullptr;
      }
      if (s[p] == '>') { ++p; break; }
      std::string k = name();
      skip_ws();
      if (p >= s.size() || s[p] != '=') return nullptr;
      ++p;
      skip_ws();
      if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) return nullptr;
      const char q = s[p++];
      size_t e = s.find(q, p);
      if (e == std::string::npos) return nullptr;
      n->attrs.emplace_back(k, unescape(s.substr(p, e - p)));
      p = e + 1;
    }
    std::string text;
    for (;;) {
      if (p >= s.size()) return nullptr;
      if (s[p] == '<') {
        if (starts("</")) {
          p += 2;
          std::string cn = name();
          skip_ws();
          if (cn != n->name || p >= s.size() || s[p] != '>') return nullptr;
          ++p;
          break;
        }
        if (starts("<!--")) { size_t e = s.find("-->", p); if (e == std::string::npos) return nullptr; p = e + 3; continue; }
        auto c = element();
        if (!c) return nullptr;
        n->children.push_back(std::move(c));
      } else {
        size_t e = s.find('<', p);
        if (e == std::string::npos) return nullptr;
        text += s.substr(p, e - p);
        p = e;
      }
    }
    // text of an element = its character data with surrounding white space removed
    size_t b = 0, e = text.size();
    while (b < e && isspace((unsigned char)text[b])) ++b;
    while (e > b && isspace((unsigned char)text[e - 1])) --e;
    n->text = unescape(text.substr(b, e - b));
    return n;
  }
};
}  // namespace

// A model of thousands of trees is one flat run of <tree> elements: cut the run at the element
// starts, parse the pieces on all host threads with the same Parser, hang them where a placeholder
// stood in the (serially parsed) rest.  Every assumption is checked -- each piece must parse into
// <tree> elements and end exactly where the next begins -- and anything unexpected (a "<tree" in a
// comment, character data between trees, fewer than 64 of them) returns nullptr: the caller then
// parses the whole document serially.
static std::unique_ptr<Node> parse_tree_run_parallel(const std::string &doc) {
  static const char kOpen[] = "<tree", kClose[] = "</tree", kHole[] = "qr-tree-run";
  std::vector<size_t> starts;
  for (size_t at = doc.find(kOpen); at != std::string::npos; at = doc.find(kOpen, at + 5)) {
    const char c = at + 5 < doc.size() ? doc[at + 5] : '\0';
    if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '>' || c == '/') starts.push_back(at);
  }
  if (starts.size() < 64) return nullptr;
  size_t last = doc.rfind(kClose);
  if (last == std::string::npos || last < starts.back()) return nullptr;
  last = doc.find('>', last);
  if (last == std::string::npos) return nullptr;
  const size_t A = starts[0], B = last + 1;
  // the document around the run
  std::string rest;
  rest.reserve(A + 32 + (doc.size() - B));
  rest.append(doc, 0, A).append("<").append(kHole).append(" />").append(doc, B, std::string::npos);
  Parser outer(rest);
  outer.skip_misc();
  std::unique_ptr<Node> root = outer.element();
  if (!root) return nullptr;
  Node *parent = nullptr;
  size_t slot = 0;
  std::vector<Node *> stack{root.get()};
  while (!stack.empty() && !parent) {
    Node *n = stack.back();
    stack.pop_back();
    for (size_t i = 0; i < n->children.size(); ++i) {
      if (n->children[i]->name == kHole) {
        parent = n;
        slot = i;
        break;
      }
      stack.push_back(n->children[i].get());
    }
  }
  if (!parent) return nullptr;
  // the run itself
  const int nt = std::max(1, std::min<int>(omp_get_max_threads(), (int)(starts.size() / 16)));
  std::vector<std::vector<std::unique_ptr<Node>>> got((size_t)nt);
  std::vector<char> ok((size_t)nt, 1);
#pragma omp parallel for num_threads(nt) schedule(static, 1)
  for (int t = 0; t < nt; ++t) {
    const size_t i0 = starts.size() * (size_t)t / (size_t)nt, i1 = starts.size() * (size_t)(t + 1) / (size_t)nt;
    const size_t lo = starts[i0], hi = i1 < starts.size() ? starts[i1] : B;
    Parser ps(doc);
    ps.p = lo;
    got[(size_t)t].reserve(i1 - i0);
    while (ps.p < hi) {
      std::unique_ptr<Node> e = ps.element();
      if (!e || e->name != "tree") {
        ok[(size_t)t] = 0;
        break;
      }
      got[(size_t)t].push_back(std::move(e));
      ps.skip_ws_comments();  // between trees: white space and comments, nothing element() would refuse
    }
    if (ps.p != hi && !(hi == B && ps.p >= B)) ok[(size_t)t] = 0;
    if (got[(size_t)t].size() != i1 - i0) ok[(size_t)t] = 0;
  }
  for (char k : ok)
    if (!k) return nullptr;
  std::vector<std::unique_ptr<Node>> kids;
  kids.reserve(parent->children.size() + starts.size());
  for (size_t i = 0; i < slot; ++i) kids.push_back(std::move(parent->children[i]));
  for (auto &v : got)
    for (auto &e : v) kids.push_back(std::move(e));
  for (size_t i = slot + 1; i < parent->children.size(); ++i) kids.push_back(std::move(parent->children[i]));
  parent->children = std::move(kids);
  return root;
}

std::unique_ptr<Node> parse(const std::string &doc) {
  if (doc.size() >= (1u << 20)) {
    std::unique_ptr<Node> n = parse_tree_run_parallel(doc);
    if (n) return n;
  }
  Parser ps(doc);
  ps.skip_misc();
  auto n = ps.element();
  return n;
}

void release(std::unique_ptr<Node> doc) {
  if (!doc) return;
  // the long child list (a model's <ensemble>) goes first, its elements side by side
  std::vector<Node *> stack{doc.get()};
  while (!stack.empty()) {
    Node *n = stack.back();
    stack.pop_back();
    if (n->children.size() >= 64) {
      const long k = (long)n->children.size();
#pragma omp parallel for schedule(static)
      for (long i = 0; i < k; ++i) n->children[(size_t)i].reset();
      n->children.clear();
    } else {
      for (auto &c : n->children) stack.push_back(c.get());
    }
  }
}

std::unique_ptr<Node> load_file(const std::string &path) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return nullptr;
  std::string doc;
  if (fseek(f, 0, SEEK_END) == 0) {
    const long n = ftell(f);
    rewind(f);
    if (n > 0) {
      doc.resize((size_t)n);
      doc.resize(fread(&doc[0], 1, (size_t)n, f));
    }
  } else {  // not seekable: by pieces
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) doc.append(buf, n);
  }
  fclose(f);
  return parse(doc);
}

}  // namespace xml
}  // namespace quickrank
