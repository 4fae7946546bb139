#include "xml.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

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
  out += "<" + n.name;
  for (auto &a : n.attrs) out += " " + a.first + "=\"" + escape(a.second, true) + "\"";
  if (n.children.empty() && n.text.empty()) {
    out += " />\n";
    return;
  }
  out += ">";
  if (n.children.empty()) {  // text-only element stays on one line
    out += escape(n.text, false) + "</" + n.name + ">\n";
    return;
  }
  out += "\n";
  for (auto &c : n.children) write_node(*c, depth + 1, out);
  out.append((size_t)depth, '\t');
  out += "</" + n.name + ">\n";
}

std::string to_string(const Node &root) {
  std::string out;
  write_node(root, 0, out);
  return out;
}

bool save_file(const Node &root, const std::string &path) {
  std::ofstream f(path, std::ofstream::out | std::ofstream::trunc);
  if (!f) return false;
  f << to_string(root);
  return (bool)f;
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

std::unique_ptr<Node> parse(const std::string &doc) {
  Parser ps(doc);
  ps.skip_misc();
  auto n = ps.element();
  return n;
}

std::unique_ptr<Node> load_file(const std::string &path) {
  std::ifstream f(path);
  if (!f) return nullptr;
  std::stringstream ss;
  ss << f.rdbuf();
  return parse(ss.str());
}

}  // namespace xml
}  // namespace quickrank
