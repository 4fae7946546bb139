// svml.cc -- SVMLight / LETOR reader and writer.
//
// Grammar and exit codes are the reference's (src/io/svml.cc:38-161 with the
// tokeniser of src/utils/strutils.cc:36-74); the implementation is not: the file is
// read into memory once, cut into one chunk per thread at line boundaries, every
// chunk is parsed independently into (label, qid, sparse features) runs, and the
// dense row-major matrix is filled in a second parallel pass once the number of
// features (the largest feature id of the whole file) is known.  The reference's
// serial getline + sscanf loop is the end-to-end bottleneck once training runs on
// the GPU.
//
// What a line means, restated (the reference works on a NUL-terminated copy of the
// line, so a NUL byte ends the line early):
//   * leading white space (" \t\n\v\f\r") is skipped; a line whose first other
//     character is '#' is a comment;
//   * the first token is the label (atof); a line without one -- blank lines
//     included -- ends the program with status 2;
//   * the next token must spell "qid:" as far as it goes (status 1), the rest is
//     read like atoi; a negative value is status 3; a missing token is qid 0;
//   * then feature tokens "id:value" up to a token that STARTS with '#'.  A '#'
//     inside a token ends that token and is swallowed (what follows is parsed as
//     more tokens).  A token that sscanf("%zu:%f") would not fully convert is
//     status 4.  The value is rounded as strtof does; trailing junk in a token is
//     ignored; the last occurrence of an id wins;
//   * the number of features is the largest id seen; queries are runs of equal
//     consecutive qids (dataset.cc:78-85).
// Deliberate differences, both on inputs where the reference has undefined
// behaviour: feature id 0 and signed feature ids (which %zu accepts and wraps
// around) are reported as malformed tokens (status 4).
#include "svml.h"

#include <omp.h>
#include <cerrno>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>

namespace quickrank {
namespace io {

namespace {

inline bool blank(char ch) {
  return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\v' || ch == '\f' || ch == '\r';
}

}  // namespace

// strtof, faster where the answer is certain.  A plain decimal ("[+-]digits[.digits][e[+-]digits]",
// at most 19 significant digits) is converted with ONE correctly rounded operation:
//   * mantissa below 2^24 and |decimal exponent| <= 10: both operands are exact floats, a single
//     float multiplication / division rounds once -- the correctly rounded result (Clinger 1990);
//   * mantissa below 2^53 and |decimal exponent| <= 22: the same in double; narrowing to float
//     rounds a second time, which changes the answer only if the double is EXACTLY half way
//     between two floats (every such midpoint is a double, and rounding to double is monotonic,
//     so any other double lies on the text's side of every midpoint) -- those, and results
//     outside the normal float range, go to strtof;
// everything else (longer mantissas, larger exponents, "inf", "nan", hex floats, no digits) goes to
// strtof as well.  `after` is set as strtof sets it: behind the longest valid prefix, an exponent
// marker without digits not included.  tests/test_host_cpu.py compares the two on millions of texts.
float parse_float(const char *s, char **after) {
  static const float p10f[] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
  static const double p10d[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char *p = s;
  bool neg = false;
  if (*p == '-' || *p == '+') neg = *p++ == '-';
  uint64_t m = 0;
  int sig = 0, ndig = 0, e10 = 0;
  bool leading = true;
  for (; *p >= '0' && *p <= '9'; ++p, ++ndig) {
    if (leading && *p == '0') continue;
    leading = false;
    if (++sig > 19) return strtof(s, after);
    m = m * 10 + (uint64_t)(*p - '0');
  }
  if (*p == '.') {
    ++p;
    for (; *p >= '0' && *p <= '9'; ++p, ++ndig) {
      if (leading && *p == '0') {
        --e10;
        continue;
      }
      leading = false;
      if (++sig > 19) return strtof(s, after);
      m = m * 10 + (uint64_t)(*p - '0');
      --e10;
    }
  }
  if (ndig == 0) return strtof(s, after);  // "inf", "nan", ".", junk: not ours to judge
  if (*p == 'e' || *p == 'E') {
    const char *q = p + 1;
    bool eneg = false;
    if (*q == '-' || *q == '+') eneg = *q++ == '-';
    if (*q >= '0' && *q <= '9') {
      int e = 0;
      for (; *q >= '0' && *q <= '9'; ++q)
        if (e < 100000) e = e * 10 + (*q - '0');
      e10 += eneg ? -e : e;
      p = q;
    }
  } else if ((*p == 'x' || *p == 'X') && ndig == 1 && p[-1] == '0') {
    return strtof(s, after);  // hexadecimal float
  }
  *after = const_cast<char *>(p);
  if (m == 0) return neg ? -0.0f : 0.0f;
  if (m < (1ull << 24) && e10 >= -10 && e10 <= 10) {
    const float f = e10 < 0 ? (float)m / p10f[-e10] : (float)m * p10f[e10];
    return neg ? -f : f;
  }
  if (m < (1ull << 53) && e10 >= -22 && e10 <= 22) {
    const double d = e10 < 0 ? (double)m / p10d[-e10] : (double)m * p10d[e10];
    uint64_t bits;
    memcpy(&bits, &d, 8);
    if (d >= 1e-37 && d <= 1e38 && (bits & 0x1FFFFFFFull) != 0x10000000ull) {
      const float f = (float)d;
      return neg ? -f : f;
    }
  }
  return strtof(s, after);
}

namespace {

// everything one thread extracted from its share of the file
struct Piece {
  std::vector<float> label;
  std::vector<unsigned> qid;
  std::vector<size_t> first;   // index of the row's first (id, value) pair
  std::vector<unsigned> fid;
  std::vector<float> val;
  size_t widest = 0;           // largest feature id
  size_t bad_at = (size_t)-1;  // byte offset of the first malformed line
  int bad_code = 0;
};

struct Cursor {
  const char *p, *end;
  void skip_blanks() {
    while (p < end && blank(*p)) ++p;
  }
  // [p, q): up to the next blank, the end of the line or (optionally) '#'; one
  // delimiter is consumed, whatever it is
  const char *token(bool hash_ends, const char **stop) {
    const char *b = p;
    while (p < end && !blank(*p) && !(hash_ends && *p == '#')) ++p;
    *stop = p;
    if (p < end) ++p;
    return b;
  }
};

// returns 0 or the exit status the line calls for
int parse_line(const char *b, const char *e, Piece &out) {
  if (const void *nul = memchr(b, '\0', (size_t)(e - b))) e = (const char *)nul;
  Cursor c{b, e};
  c.skip_blanks();
  if (c.p < c.end && *c.p == '#') return -1;  // comment
  if (c.p == c.end) return 2;                 // the label is mandatory
  const char *stop;
  const char *tok = c.token(false, &stop);
  const float label = (float)strtod(tok, nullptr);  // stops at the blank that ends the token
  // qid
  c.skip_blanks();
  long long q = 0;
  if (c.p < c.end) {
    tok = c.token(false, &stop);
    static const char want[] = "qid:";
    const char *s = tok;
    for (int i = 0; want[i] && s < stop; ++i, ++s)
      if (*s != want[i]) return 1;
    bool neg = false;
    if (s < stop && (*s == '-' || *s == '+')) neg = *s++ == '-';
    for (; s < stop && *s >= '0' && *s <= '9'; ++s) q = q * 10 + (*s - '0');
    if (neg) q = -q;
    if ((int)q < 0) return 3;
  }
  out.label.push_back(label);
  out.qid.push_back((unsigned)(int)q);
  out.first.push_back(out.fid.size());
  // features
  for (;;) {
    c.skip_blanks();
    if (c.p == c.end || *c.p == '#') break;  // end of line / trailing description
    tok = c.token(true, &stop);
    const char *s = tok;
    size_t id = 0;
    if (s == stop || *s < '0' || *s > '9') return 4;
    for (; s < stop && *s >= '0' && *s <= '9'; ++s) id = id * 10 + (size_t)(*s - '0');
    if (s == stop || *s != ':' || id == 0 || id > 0xFFFFFFFFull) return 4;
    ++s;
    if (s == stop) return 4;  // "id:" with nothing behind it
    char *after;
    const float v = parse_float(s, &after);  // cannot run past `stop`: blanks and '#' end a number
    if (after == s) return 4;
    out.fid.push_back((unsigned)id);
    out.val.push_back(v);
    if (id > out.widest) out.widest = id;
  }
  return 0;
}

void parse_range(const char *base, size_t lo, size_t hi, Piece &out) {
  size_t at = lo;
  while (at < hi) {
    const char *nl = (const char *)memchr(base + at, '\n', hi - at);
    const size_t stop = nl ? (size_t)(nl - base) : hi;
    // a '\n'-terminated empty line is a line (status 2); bytes after the last '\n'
    // are a line only if there are any (getline returns -1 at end of file)
    if (nl || stop > at) {
      const int rc = parse_line(base + at, base + stop, out);
      if (rc > 0) {
        out.bad_at = at;
        out.bad_code = rc;
        return;
      }
    }
    at = stop + 1;
  }
}

}  // namespace

std::unique_ptr<data::Dataset> Svml::read_horizontal(const std::string &filename) {
  const int fd = open(filename.c_str(), O_RDONLY);
  if (fd < 0) {
    std::cerr << "!!! Error while opening file " << filename << "." << std::endl;
    exit(EXIT_FAILURE);
  }
  struct stat st;
  if (fstat(fd, &st) != 0) {
    std::cerr << "!!! Error while opening file " << filename << "." << std::endl;
    exit(EXIT_FAILURE);
  }
  stats_.bytes = st.st_size;
  const auto t0 = std::chrono::high_resolution_clock::now();
  // the whole text in memory, read by all threads at once (pread: no shared file position),
  // into a buffer nobody clears first
  std::unique_ptr<char[]> text(new char[(size_t)stats_.bytes + 1]);
  size_t got = (size_t)stats_.bytes;
  {
    const size_t share = 64u << 20;
    const long nshare = (long)((got + share - 1) / share);
    bool failed = false;
#pragma omp parallel for schedule(dynamic, 1)
    for (long k = 0; k < nshare; ++k) {
      size_t at = (size_t)k * share;
      const size_t hi = std::min(got, at + share);
      while (at < hi) {
        const ssize_t n = pread(fd, text.get() + at, hi - at, (off_t)at);
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) {  // an I/O error, or the file is shorter than fstat said: a text cut somewhere in
                       // a line is not a dataset -- the reference's message and exit, not a shorter one
#pragma omp atomic write
          failed = true;
          break;
        }
        at += (size_t)n;
      }
    }
    if (failed) {
      std::cerr << "!!! Error while reading file " << filename << "." << std::endl;
      exit(EXIT_FAILURE);
    }
  }
  close(fd);
  text[got] = '\0';  // strtod / strtof may look one byte past the last token

  int nthreads = omp_get_max_threads();
  if ((size_t)nthreads > got / (1 << 16) + 1) nthreads = (int)(got / (1 << 16) + 1);
  std::vector<size_t> cut(nthreads + 1, got);
  cut[0] = 0;
  for (int t = 1; t < nthreads; ++t) {  // chunk borders on line starts
    size_t at = got / nthreads * t;
    if (at < cut[t - 1]) at = cut[t - 1];
    const char *nl = (const char *)memchr(text.get() + at, '\n', got - at);
    cut[t] = nl ? (size_t)(nl - text.get()) + 1 : got;
  }
  std::vector<Piece> pieces(nthreads);
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
  for (int t = 0; t < nthreads; ++t) parse_range(text.get(), cut[t], cut[t + 1], pieces[t]);
  // the reference stops at the FIRST malformed line of the file
  for (const Piece &p : pieces)
    if (p.bad_code) exit(p.bad_code);
  text.reset();  // everything needed is in the pieces: the text goes before the matrix comes
  const auto t1 = std::chrono::high_resolution_clock::now();

  size_t rows = 0, width = 0;
  std::vector<size_t> row0(nthreads + 1, 0);
  for (int t = 0; t < nthreads; ++t) {
    row0[t] = rows;
    rows += pieces[t].label.size();
    width = std::max(width, pieces[t].widest);
  }
  row0[nthreads] = rows;
  auto dataset = std::unique_ptr<data::Dataset>(new data::Dataset(rows, width));
  std::vector<QueryID> qids(rows);
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
  for (int t = 0; t < nthreads; ++t) {
    const Piece &p = pieces[t];
    for (size_t r = 0; r < p.label.size(); ++r) {
      const size_t i = row0[t] + r;
      qids[i] = p.qid[r];
      Feature *row = dataset->at(i, 0);
      const size_t hi = r + 1 < p.first.size() ? p.first[r + 1] : p.fid.size();
      for (size_t k = p.first[r]; k < hi; ++k) row[p.fid[k] - 1] = p.val[k];  // later pairs win
      dataset->set_label(i, p.label[r]);
    }
    pieces[t] = Piece();  // (released by the thread that made it, as soon as its rows stand)
  }
  dataset->close_rows(qids);
  const auto t2 = std::chrono::high_resolution_clock::now();
  stats_.parse_s = std::chrono::duration<double>(t1 - t0).count();
  stats_.fill_s = std::chrono::duration<double>(t2 - t1).count();
  return dataset;
}

bool write_scores(const std::string &path, const double *scores, size_t n) {
  FILE *out = fopen(path.c_str(), "wb");
  if (!out) return false;
  const size_t piece = 1u << 16;  // scores per task
  const long npieces = (long)((n + piece - 1) / piece);
  const int nt = omp_get_max_threads();
  bool ok = true;
  // rounds of one piece per thread: formatted side by side, written in order
  std::vector<std::vector<char>> text((size_t)nt);
  for (long base = 0; base < npieces && ok; base += nt) {
    const long cnt = std::min<long>(nt, npieces - base);
#pragma omp parallel for num_threads(nt) schedule(static, 1)
    for (long k = 0; k < cnt; ++k) {
      std::vector<char> &b = text[(size_t)k];
      const size_t lo = (size_t)(base + k) * piece, hi = std::min(n, lo + piece);
      b.resize((hi - lo) * 26);  // "-1.7976931348623157e+308\n" is 25 characters
      size_t at = 0;
      for (size_t i = lo; i < hi; ++i) {
        at += (size_t)snprintf(b.data() + at, 26, "%.17g", scores[i]);
        b[at++] = '\n';
      }
      b.resize(at);
    }
    for (long k = 0; k < cnt && ok; ++k) ok = fwrite(text[(size_t)k].data(), 1, text[(size_t)k].size(), out) == text[(size_t)k].size();
  }
  return (fclose(out) == 0) && ok;
}

// Same bytes as svml.cc:163-188 produces through its iostream manipulators: the
// very first label is printed in the default float format with precision 0
// ("%.0g"); std::fixed then stays set, so every later label comes out as "%.0f";
// feature values are "%.9f" (max_digits10 of float).
void Svml::write(const data::Dataset &dataset, const std::string &file) {
  FILE *out = fopen(file.c_str(), "w");
  if (!out) {
    std::cerr << "!!! Error while opening file " << file << "." << std::endl;
    exit(EXIT_FAILURE);
  }
  // the query of every row, then rows formatted in pieces on all host threads, written in order
  const size_t N = dataset.num_instances(), F = dataset.num_features();
  std::vector<size_t> qid(N);
  for (size_t q = 0; q < dataset.num_queries(); ++q)
    for (size_t r = dataset.offset(q); r < dataset.offset(q + 1); ++r) qid[r] = q + 1;
  const size_t per_row = 64 + F * 96;  // FLT_MAX in %.9f is 49 characters
  const size_t piece = std::max<size_t>(1, (4u << 20) / per_row);
  const long npieces = (long)((N + piece - 1) / piece);
  const int nt = omp_get_max_threads();
  std::vector<std::vector<char>> text((size_t)nt);
  for (long base = 0; base < npieces; base += nt) {
    const long cnt = std::min<long>(nt, npieces - base);
#pragma omp parallel for num_threads(nt) schedule(static, 1)
    for (long k = 0; k < cnt; ++k) {
      std::vector<char> &b = text[(size_t)k];
      const size_t lo = (size_t)(base + k) * piece, hi = std::min(N, lo + piece);
      b.resize((hi - lo) * per_row);
      size_t n = 0;
      for (size_t r = lo; r < hi; ++r) {
        n += (size_t)snprintf(b.data() + n, 64, r == 0 ? "%.0g qid:%zu" : "%.0f qid:%zu", (double)dataset.getLabel(r),
                              qid[r]);
        const Feature *x = dataset.at(r, 0);
        for (size_t i = 0; i < F; ++i) n += (size_t)snprintf(b.data() + n, 96, " %zu:%.9f", i + 1, (double)x[i]);
        b[n++] = '\n';
      }
      b.resize(n);
    }
    for (long k = 0; k < cnt; ++k) fwrite(text[(size_t)k].data(), 1, text[(size_t)k].size(), out);
  }
  fclose(out);
}

}  // namespace io
}  // namespace quickrank
