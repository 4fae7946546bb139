// host_capi.cc -- C shims over the host classes for the (CPU-side) tests of the
// SVMLight reader and the XML model round trip; no device calls here.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "codegen.h"
#include "mart.h"
#include "svml.h"

using namespace quickrank;

namespace quickrank { namespace learning { namespace forests {
void multi_query_slice(size_t Q, int r, int w, size_t *q0, size_t *q1);  // mart_multi.cc
}}}

extern "C" {

// whole queries [q0, q1) that rank r of w holds under `quicklearn --gpus w --shard docs`
void qrh_query_slice(size_t Q, int r, int w, size_t *q0, size_t *q1) {
  learning::forests::multi_query_slice(Q, r, w, q0, q1);
}

// reads `path`; returns N, F, Q through pointers, copies into caller buffers when
// they are non-NULL (call once with NULLs to size them)
int qrh_svml_read(const char *path, size_t *N, size_t *F, size_t *Q, float *x, float *labels,
                  uint64_t *qoff) {
  io::Svml reader;
  auto ds = reader.read_horizontal(path);
  *N = ds->num_instances();
  *F = ds->num_features();
  *Q = ds->num_queries();
  if (x) memcpy(x, ds->at(0, 0), *N * *F * sizeof(float));
  if (labels) memcpy(labels, ds->labels(), *N * sizeof(float));
  if (qoff) memcpy(qoff, ds->offsets().data(), (*Q + 1) * sizeof(uint64_t));
  return 0;
}

// Test hooks for io::parse_float (the reader's number conversion) against strtof itself:
// `texts` holds n NUL-terminated strings back to back; returns how many differ in value bits or
// in the end pointer, the index of the first in *first_bad.
size_t qrh_parse_float_check(const char *texts, size_t n, size_t *first_bad) {
  size_t bad = 0;
  const char *p = texts;
  for (size_t i = 0; i < n; ++i) {
    char *a1, *a2;
    const float u = io::parse_float(p, &a1), v = strtof(p, &a2);
    if (memcmp(&u, &v, 4) != 0 || a1 != a2) {
      if (!bad++ && first_bad) *first_bad = i;
    }
    p += strlen(p) + 1;
  }
  return bad;
}
// ... and on texts made here: every "0.dddddd" (the usual look of a LETOR value), then `n` random
// floats printed as %.9f (the writer's format), %.9g, %e, %.17g of a nearby double, and with 1..25
// random digits and exponents.  Returns the number of disagreements.
size_t qrh_parse_float_selftest(uint64_t seed, size_t n, char *first_bad, size_t first_bad_len) {
  size_t bad = 0;
  char buf[128];
  auto check = [&](const char *t) {
    char *a1, *a2;
    const float u = io::parse_float(t, &a1), v = strtof(t, &a2);
    if (memcmp(&u, &v, 4) != 0 || a1 != a2) {
      if (!bad++ && first_bad && first_bad_len) {
        strncpy(first_bad, t, first_bad_len - 1);
        first_bad[first_bad_len - 1] = 0;
      }
    }
  };
  for (unsigned k = 0; k < 1000000; ++k) {
    snprintf(buf, sizeof buf, "0.%06u", k);
    check(buf);
  }
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
  auto next = [&]() {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
  };
  for (size_t i = 0; i < n; ++i) {
    uint32_t bits = (uint32_t)next();
    if (((bits >> 23) & 0xFF) == 0xFF) bits &= 0x7F7FFFFFu;  // finite
    float f;
    memcpy(&f, &bits, 4);
    snprintf(buf, sizeof buf, "%.9f", (double)f);
    if (strlen(buf) < 60) check(buf);
    snprintf(buf, sizeof buf, "%.9g", (double)f);
    check(buf);
    snprintf(buf, sizeof buf, "%e", (double)f);
    check(buf);
    // a double next to a float midpoint, printed exactly enough to land on either side
    float g;
    uint32_t b2 = bits + 1;
    memcpy(&g, &b2, 4);
    const double mid = ((double)f + (double)g) * 0.5;
    snprintf(buf, sizeof buf, "%.17g", mid);
    check(buf);
    snprintf(buf, sizeof buf, "%.17g", nextafter(mid, 1e300));
    check(buf);
    snprintf(buf, sizeof buf, "%.16g", nextafter(mid, -1e300));
    check(buf);
    // random digit strings: 1..25 digits, a point somewhere, sometimes an exponent, sometimes junk behind
    int nd = 1 + (int)(next() % 25), pt = (int)(next() % (unsigned)(nd + 2)) - 1, o = 0;
    if (next() % 4 == 0) buf[o++] = next() % 2 ? '-' : '+';
    for (int d = 0; d < nd; ++d) {
      if (d == pt) buf[o++] = '.';
      buf[o++] = (char)('0' + next() % 10);
    }
    if (pt == nd) buf[o++] = '.';
    switch (next() % 6) {
      case 0: o += snprintf(buf + o, 16, "e%d", (int)(next() % 80) - 40); break;
      case 1: o += snprintf(buf + o, 16, "E+%d", (int)(next() % 12)); break;
      case 2: buf[o++] = 'e'; break;
      case 3: buf[o++] = 'x'; buf[o++] = '1'; break;
      default: break;
    }
    buf[o] = 0;
    check(buf);
  }
  return bad;
}

// The same in one pass over the file: open (parse), ask the sizes, copy out, close.
void *qrh_svml_open(const char *path, size_t *N, size_t *F, size_t *Q) {
  io::Svml reader;
  data::Dataset *ds = reader.read_horizontal(path).release();
  *N = ds->num_instances();
  *F = ds->num_features();
  *Q = ds->num_queries();
  return ds;
}
void qrh_svml_copy(const void *handle, float *x, float *labels, uint64_t *qoff) {
  const data::Dataset *ds = (const data::Dataset *)handle;
  const size_t N = ds->num_instances(), F = ds->num_features(), Q = ds->num_queries();
  if (x) memcpy(x, ds->at(0, 0), N * F * sizeof(float));
  if (labels) memcpy(labels, ds->labels(), N * sizeof(float));
  if (qoff) memcpy(qoff, ds->offsets().data(), (Q + 1) * sizeof(uint64_t));
}
void qrh_svml_close(void *handle) { delete (data::Dataset *)handle; }

int qrh_write_scores(const char *path, const double *scores, size_t n) {
  return io::write_scores(path, scores, n) ? 0 : 1;
}

int qrh_svml_write(const char *path, const float *x, const float *labels, const uint64_t *qoff,
                   size_t Q, size_t F) {
  const size_t N = qoff[Q];
  data::Dataset ds(N, F);
  for (size_t q = 0; q < Q; ++q)
    for (size_t i = qoff[q]; i < qoff[q + 1]; ++i)
      ds.addInstance((QueryID)(q + 1), labels[i], std::vector<Feature>(x + i * F, x + (i + 1) * F));
  io::Svml().write(ds, path);
  return 0;
}

// load a model file and write it back (XML round trip)
int qrh_model_roundtrip(const char *in_path, const char *out_path) {
  auto m = learning::forests::Mart::load_model_from_file(in_path);
  if (!m) return 1;
  m->save(out_path);
  return 0;
}

// ... the same with the two halves timed (scripts/xml_bench.py)
int qrh_model_roundtrip_timed(const char *in_path, const char *out_path, double *load_s, double *save_s) {
  const auto t0 = std::chrono::high_resolution_clock::now();
  auto m = learning::forests::Mart::load_model_from_file(in_path);
  const auto t1 = std::chrono::high_resolution_clock::now();
  if (!m) return 1;
  m->save(out_path);
  const auto t2 = std::chrono::high_resolution_clock::now();
  *load_s = std::chrono::duration<double>(t1 - t0).count();
  *save_s = std::chrono::duration<double>(t2 - t1).count();
  return 0;
}

}  // extern "C"

// a model from flat node records (qr_node_t layout), as Mart::save would write it; `weights`:
// one per tree, or nullptr (= the shrinkage for every tree)
static int model_write(const char *path, int algo, size_t ntrees_cfg, double shrinkage, size_t nthresholds,
                       size_t nleaves, size_t minls, size_t esr, size_t depth, const qr_node_t *nodes,
                       size_t ntrees, size_t max_nodes, const double *weights) {
  using learning::forests::Mart;
  using learning::forests::RTNode;
  // build an XML document directly from the records through the same writer
  struct B {
    static std::unique_ptr<RTNode> rec(const qr_node_t *n, int i) {
      std::unique_ptr<RTNode> r(new RTNode());
      r->avglabel = n[i].value;
      if (n[i].feature >= 0) {
        r->featureidx = n[i].feature;
        r->featureid = (unsigned)n[i].feature + 1;
        r->threshold = n[i].threshold;
        r->left = rec(n, n[i].left);
        r->right = rec(n, n[i].right);
      }
      return r;
    }
  };
  Mart proto((Mart::Algo)algo, ntrees_cfg, shrinkage, nthresholds, nleaves, minls, esr, depth);
  auto doc = proto.get_xml_model();
  xml::Node *ens = nullptr;
  for (auto &c : doc->children)
    if (c->name == "ensemble") ens = c.get();
  for (size_t t = 0; t < ntrees; ++t) {
    xml::Node *tree = ens->append_child("tree");
    tree->append_attribute("id", std::to_string(t + 1));
    tree->append_attribute("weight", xml::fmt_double(weights ? weights[t] : shrinkage));
    B::rec(nodes + t * max_nodes, 0)->append_xml_model(tree);
  }
  return xml::save_file(*doc, path) ? 0 : 1;
}

extern "C" {

int qrh_model_write(const char *path, int algo, size_t ntrees_cfg, double shrinkage, size_t nthresholds,
                    size_t nleaves, size_t minls, size_t esr, size_t depth, const qr_node_t *nodes,
                    size_t ntrees, size_t max_nodes) {
  return model_write(path, algo, ntrees_cfg, shrinkage, nthresholds, nleaves, minls, esr, depth, nodes, ntrees,
                     max_nodes, nullptr);
}

// ... with a weight per tree (quickrank_amd/io.py: the Python trainer's Ensemble keeps one)
int qrh_model_write_w(const char *path, int algo, size_t ntrees_cfg, double shrinkage, size_t nthresholds,
                      size_t nleaves, size_t minls, size_t esr, size_t depth, const qr_node_t *nodes,
                      size_t ntrees, size_t max_nodes, const double *weights) {
  return model_write(path, algo, ntrees_cfg, shrinkage, nthresholds, nleaves, minls, esr, depth, nodes, ntrees,
                     max_nodes, weights);
}

// the <info> block of a model file: out[6] = trees, thresholds, leaves, min leaf support, early-stop
// rounds, depth
int qrh_model_info(const char *path, int *algo, size_t *out, double *shrinkage) {
  auto doc = xml::load_file(path);   // (the whole document: the block could sit behind the trees)
  if (!doc || doc->name != "ranker") return 1;
  // drop the trees before the model object is made: only the block is asked for
  for (auto &c : doc->children)
    if (c->name == "ensemble") c->children.clear();
  learning::forests::Mart::Algo a;
  const xml::Node *info = doc->child("info");
  if (!info || !learning::forests::Mart::algo_from_name(info->child_text("type"), &a)) return 1;
  learning::forests::Mart m(*doc);
  m.info(algo, out, shrinkage);
  return 0;
}

// flatten a model file into qr_node_t records (for scoring through the C-ABI)
int qrh_model_read(const char *path, qr_node_t *nodes, double *weights, size_t *ntrees,
                   size_t *max_nodes, size_t cap_nodes, size_t cap_trees) {
  auto m = learning::forests::Mart::load_model_from_file(path);
  if (!m) return 1;
  std::vector<qr_node_t> n;
  std::vector<double> w;
  const size_t mn = m->ensemble().flatten(&n, &w);
  *ntrees = w.size();
  *max_nodes = mn;
  if (nodes && weights) {
    if (n.size() > cap_nodes || w.size() > cap_trees) return 2;
    memcpy(nodes, n.data(), n.size() * sizeof(qr_node_t));
    memcpy(weights, w.data(), w.size() * sizeof(double));
  }
  return 0;
}

// One parse of a model file for everything a caller wants of it (quickrank_amd/io.load_model): the
// handle keeps the model object; info / sizes / records are read from it; nullptr for a file that is
// not a model of the four algorithms.
struct QrhModel {
  std::shared_ptr<learning::forests::Mart> m;
  std::vector<qr_node_t> nodes;
  std::vector<double> weights;
  size_t max_nodes = 0;
};
void *qrh_model_open(const char *path, int *algo, size_t *out, double *shrinkage, size_t *ntrees, size_t *max_nodes) {
  std::unique_ptr<QrhModel> h(new QrhModel());
  h->m = learning::forests::Mart::load_model_from_file(path);
  if (!h->m) return nullptr;
  h->m->info(algo, out, shrinkage);
  h->max_nodes = h->m->ensemble().flatten(&h->nodes, &h->weights);
  *ntrees = h->weights.size();
  *max_nodes = h->max_nodes;
  return h.release();
}
int qrh_model_copy(const void *handle, qr_node_t *nodes, double *weights, size_t cap_nodes, size_t cap_trees) {
  const QrhModel *h = (const QrhModel *)handle;
  if (h->nodes.size() > cap_nodes || h->weights.size() > cap_trees) return 2;
  memcpy(nodes, h->nodes.data(), h->nodes.size() * sizeof(qr_node_t));
  memcpy(weights, h->weights.data(), h->weights.size() * sizeof(double));
  return 0;
}
void qrh_model_close(void *handle) { delete (QrhModel *)handle; }

// `--model-file / --code-file / --generator` without the command line
int qrh_codegen(const char *generator, const char *model_file, const char *code_file) {
  return io::generate(generator, model_file, code_file);
}

}  // extern "C"
