/*
 * qr_oracle.c -- CPU restatement of QuickRank's LambdaMART/GBRT hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see qr_oracle.h).  Plain C11 + OpenMP, built with
 * `gcc -O2 -ffp-contract=off -fopenmp` (generic x86-64, no FMA contraction:
 * SURVEY.md section 7 hard part 8 -- reproducible IEEE semantics).
 *
 * Reference citations are file:line in the upstream hpclab/quickrank tree.
 */
#define _GNU_SOURCE /* sigaction, MAP_ANONYMOUS (the guarded lists of the hunt's mode) */
#include "qr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ========================================================================== */
/* Self-checks.  The oracle is the judge of the device's results, so it checks */
/* its own state where the algorithm gives it two views of one fact (a child's  */
/* document count from the histogram vs the length of its sample list; a leaf's */
/* count vs the documents mapped to it).  A disagreement is not a property of   */
/* the algorithm: it means host memory changed under the run                    */
/* (profiles/r05_abort_hunt.md).  It is recorded here, the stores that would    */
/* leave their buffers are skipped, and oracle/__init__.py raises on it.        */
/* ========================================================================== */
#include <stdio.h>
static int qro_events_n = 0;
static char qro_events_msg[1024];
static void qro_event(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#include <stdarg.h>
static void qro_event(const char *fmt, ...) {
#pragma omp critical(qro_event_log)
  {
    if (qro_events_n++ == 0) {
      va_list ap;
      va_start(ap, fmt);
      vsnprintf(qro_events_msg, sizeof(qro_events_msg), fmt, ap);
      va_end(ap);
    }
  }
}
/* number of self-check events since the last call (and the first one's text) */
int qro_self_check(char *msg, size_t n) {
  const int k = qro_events_n;
  if (msg && n) {
    msg[0] = 0;
    if (k) {
      strncpy(msg, qro_events_msg, n - 1);
      msg[n - 1] = 0;
    }
  }
  qro_events_n = 0;
  return k;
}

/* Sample lists under guard (QRO_GUARD=1, the hunt's mode; profiles/r05_abort_hunt.md): a list is
 * written once (the partition of its parent) and only read afterwards, so it can live in pages of
 * its own that turn read-only once filled.  A CPU store into one -- from any thread, any library --
 * then faults AT THE WRITER: the handler prints the faulting address and the native backtrace of
 * that thread and aborts.  A list that changes WITHOUT a fault (the self-checks below still see
 * it) was changed by something that does not go through the CPU's page tables: a DMA.          */
#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>
static int qro_guard_on = -1;
static void qro_guard_segv(int sig, siginfo_t *si, void *uc) {
  (void)uc;
  static const char head[] = "\nqr_oracle: store into a guarded sample list -- the writer's stack:\n";
  if (write(2, head, sizeof(head) - 1) < 0) _exit(70);
  char line[96];
  const int n = snprintf(line, sizeof line, "  signal %d, address %p\n", sig, si ? si->si_addr : NULL);
  if (n > 0 && write(2, line, (size_t)n) < 0) _exit(70);
  void *frames[64];
  backtrace_symbols_fd(frames, backtrace(frames, 64), 2);
  signal(SIGABRT, SIG_DFL);
  abort();
}
static int qro_guard(void) {
  if (qro_guard_on < 0) {
    const char *e = getenv("QRO_GUARD");
    qro_guard_on = e && *e && *e != '0';
    if (qro_guard_on) {
      struct sigaction sa;
      memset(&sa, 0, sizeof sa);
      sa.sa_sigaction = qro_guard_segv;
      sa.sa_flags = SA_SIGINFO;
      sigaction(SIGSEGV, &sa, NULL);
    }
  }
  return qro_guard_on;
}
static size_t qro_page_round(size_t b) {
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  if (!b) b = 1;
  return (b + page - 1) / page * page;
}
/* Every sealed block is also remembered with a hash of its bytes; the hash is checked again when the
 * block is released.  A block whose bytes changed although no store ever faulted was written past
 * the CPU's page tables -- by a DMA -- and says so (round 6: the matrix, the bin map and the
 * thresholds of qro_train are sealed like the lists, a few hundred KB to tens of MB per run). */
#define QRO_SEALS 4096
static struct { const void *p; size_t bytes; uint64_t h; } qro_seals[QRO_SEALS];
static uint64_t qro_hash(const void *p, size_t bytes) {
  const uint64_t *w = (const uint64_t *)p;
  uint64_t h = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < bytes / 8; ++i) h = (h ^ w[i]) * 0x100000001B3ull + (h >> 29);
  const unsigned char *b = (const unsigned char *)p + (bytes & ~(size_t)7);
  for (size_t i = 0; i < (bytes & 7); ++i) h = (h ^ b[i]) * 0x100000001B3ull;
  return h;
}
static void *qro_guard_alloc(size_t bytes) {
  if (!qro_guard()) return malloc(bytes ? bytes : 1);
  void *p = mmap(NULL, qro_page_round(bytes), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  return p == MAP_FAILED ? NULL : p;
}
static void qro_guard_seal(void *p, size_t bytes) { /* filled: read-only from here on */
  if (!p || !qro_guard()) return;
  const uint64_t h = qro_hash(p, bytes);
#pragma omp critical(qro_seal_table)
  for (int i = 0; i < QRO_SEALS; ++i)
    if (!qro_seals[i].p) {
      qro_seals[i].p = p;
      qro_seals[i].bytes = bytes;
      qro_seals[i].h = h;
      break;
    }
  mprotect(p, qro_page_round(bytes), PROT_READ);
}
static void qro_guard_free(void *p, size_t bytes) {
  if (!p) return;
  if (!qro_guard()) {
    free(p);
    return;
  }
  int found = 0;
  uint64_t h0 = 0;
#pragma omp critical(qro_seal_table)
  for (int i = 0; i < QRO_SEALS; ++i)
    if (qro_seals[i].p == p) {
      found = 1;
      h0 = qro_seals[i].h;
      qro_seals[i].p = NULL;
      break;
    }
  if (found && qro_hash(p, bytes) != h0)
    qro_event("a sealed block of %zu bytes at %p changed while it was read-only and no store faulted: "
              "written past the page tables (a DMA)", bytes, p);
  munmap(p, qro_page_round(bytes));
}
static uint64_t *qro_list_alloc(size_t n) { return (uint64_t *)qro_guard_alloc(sizeof(uint64_t) * (n ? n : 1)); }
static void qro_list_seal(uint64_t *p, size_t n) { qro_guard_seal(p, sizeof(uint64_t) * (n ? n : 1)); }
static void qro_list_free(uint64_t *p, size_t n) { qro_guard_free(p, sizeof(uint64_t) * (n ? n : 1)); }

void qro_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ========================================================================== */
/* libstdc++ 11 std::sort behaviour, generic over a    */
/* comparator on u64 elements.  The element array holds either doc indices    */
/* (ranking sort) or float bit patterns (label sort).                         */
/* ========================================================================== */
typedef int (*cmp_fn)(const void *ctx, uint64_t a, uint64_t b);

/* queryresults.cc:37-45: operator()(int i, int j) { values_[i] > values_[j] } */
static int cmp_score_desc(const void *ctx, uint64_t a, uint64_t b) {
  const double *s = (const double *)ctx;
  return s[(int)a] > s[(int)b];
}
/* std::greater<int> applied to float elements (ndcg.cc:40-41): implicit      */
/* float -> int conversion of both operands.                                  */
static int cmp_label_int_desc(const void *ctx, uint64_t a, uint64_t b) {
  (void)ctx;
  float fa, fb;
  uint32_t ua = (uint32_t)a, ub = (uint32_t)b;
  memcpy(&fa, &ua, 4);
  memcpy(&fb, &ub, 4);
  return (int)fa > (int)fb;
}

static void s_swap(uint64_t *a, uint64_t *b) {
  uint64_t t = *a;
  *a = *b;
  *b = t;
}

static void s_push_heap(uint64_t *first, ptrdiff_t hole, ptrdiff_t top,
                        uint64_t value, cmp_fn c, const void *ctx) {
  ptrdiff_t parent = (hole - 1) / 2;
  while (hole > top && c(ctx, first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

static void s_adjust_heap(uint64_t *first, ptrdiff_t hole, ptrdiff_t len,
                          uint64_t value, cmp_fn c, const void *ctx) {
  const ptrdiff_t top = hole;
  ptrdiff_t child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (c(ctx, first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  s_push_heap(first, hole, top, value, c, ctx);
}

static void s_make_heap(uint64_t *first, ptrdiff_t len, cmp_fn c,
                        const void *ctx) {
  if (len < 2) return;
  ptrdiff_t parent = (len - 2) / 2;
  for (;;) {
    uint64_t v = first[parent];
    s_adjust_heap(first, parent, len, v, c, ctx);
    if (parent == 0) return;
    parent--;
  }
}

/* __partial_sort(first, last, last): make_heap + sort_heap                   */
static void s_heapsort(uint64_t *first, ptrdiff_t len, cmp_fn c,
                       const void *ctx) {
  s_make_heap(first, len, c, ctx);
  ptrdiff_t last = len;
  while (last > 1) {
    --last;
    uint64_t v = first[last];
    first[last] = first[0];
    s_adjust_heap(first, 0, last, v, c, ctx);
  }
}

static void s_move_median_to_first(uint64_t *result, uint64_t *a, uint64_t *b,
                                   uint64_t *cc, cmp_fn c, const void *ctx) {
  if (c(ctx, *a, *b)) {
    if (c(ctx, *b, *cc))
      s_swap(result, b);
    else if (c(ctx, *a, *cc))
      s_swap(result, cc);
    else
      s_swap(result, a);
  } else if (c(ctx, *a, *cc))
    s_swap(result, a);
  else if (c(ctx, *b, *cc))
    s_swap(result, cc);
  else
    s_swap(result, b);
}

static uint64_t *s_unguarded_partition(uint64_t *first, uint64_t *last,
                                       uint64_t *pivot, cmp_fn c,
                                       const void *ctx) {
  for (;;) {
    while (c(ctx, *first, *pivot)) ++first;
    --last;
    while (c(ctx, *pivot, *last)) --last;
    if (!(first < last)) return first;
    s_swap(first, last);
    ++first;
  }
}

static void s_introsort_loop(uint64_t *first, uint64_t *last, long depth,
                             cmp_fn c, const void *ctx) {
  while (last - first > 16) {
    if (depth == 0) {
      s_heapsort(first, last - first, c, ctx);
      return;
    }
    --depth;
    uint64_t *mid = first + (last - first) / 2;
    s_move_median_to_first(first, first + 1, mid, last - 1, c, ctx);
    uint64_t *cut = s_unguarded_partition(first + 1, last, first, c, ctx);
    s_introsort_loop(cut, last, depth, c, ctx);
    last = cut;
  }
}

static void s_unguarded_linear_insert(uint64_t *last, cmp_fn c,
                                      const void *ctx) {
  uint64_t val = *last;
  uint64_t *next = last - 1;
  while (c(ctx, val, *next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}

static void s_insertion_sort(uint64_t *first, uint64_t *last, cmp_fn c,
                             const void *ctx) {
  if (first == last) return;
  for (uint64_t *i = first + 1; i != last; ++i) {
    if (c(ctx, *i, *first)) {
      uint64_t val = *i;
      memmove(first + 1, first, (size_t)(i - first) * sizeof(uint64_t));
      *first = val;
    } else
      s_unguarded_linear_insert(i, c, ctx);
  }
}

static void gnu_sort(uint64_t *first, size_t n, cmp_fn c, const void *ctx) {
  if (n == 0) return;
  long lg = 0;
  for (size_t t = n; t > 1; t >>= 1) ++lg; /* std::__lg */
  uint64_t *last = first + n;
  s_introsort_loop(first, last, lg * 2, c, ctx);
  if (n > 16) {
    s_insertion_sort(first, first + 16, c, ctx);
    for (uint64_t *i = first + 16; i != last; ++i)
      s_unguarded_linear_insert(i, c, ctx);
  } else
    s_insertion_sort(first, last, c, ctx);
}

void qro_rank_by_score(const double *scores, size_t n, uint64_t *idx) {
  for (size_t i = 0; i < n; ++i) idx[i] = i; /* queryresults.cc:50-51 */
  gnu_sort(idx, n, cmp_score_desc, scores);  /* queryresults.cc:52    */
}

void qro_heapsort_by_score(const double *scores, size_t n, uint64_t *idx) {
  for (size_t i = 0; i < n; ++i) idx[i] = i;
  s_heapsort(idx, (ptrdiff_t)n, cmp_score_desc, scores);
}

void qro_sort_labels_desc_int(float *labels, size_t n) {
  uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    uint32_t u;
    memcpy(&u, &labels[i], 4);
    tmp[i] = u;
  }
  gnu_sort(tmp, n, cmp_label_int_desc, NULL);
  for (size_t i = 0; i < n; ++i) {
    uint32_t u = (uint32_t)tmp[i];
    memcpy(&labels[i], &u, 4);
  }
  free(tmp);
}

/* ========================================================================== */
/* Metric                                                                     */
/* ========================================================================== */
static size_t eff_cutoff(size_t k) { return k == 0 ? SIZE_MAX : k; } /* metric.h:65-67 */

/* dcg.cc:33-39.  `i + 2.0f` is formed in f32 then promoted; pow/log2 are the */
/* double-precision libm entry points.         */
double qro_dcg(const float *labels, size_t len, size_t cutoff) {
  const size_t k = eff_cutoff(cutoff);
  const size_t size = k < len ? k : len;
  double dcg = 0.0;
  for (size_t i = 0; i < size; ++i)
    dcg += (pow(2.0, (double)labels[i]) - 1.0f) / log2((double)((float)i + 2.0f));
  return dcg;
}

/* ndcg.cc:35-47 */
double qro_idcg(const float *labels, size_t n, size_t cutoff) {
  float *copy = (float *)malloc(sizeof(float) * (n ? n : 1));
  memcpy(copy, labels, sizeof(float) * n);
  qro_sort_labels_desc_int(copy, n);
  double d = qro_dcg(copy, n, cutoff);
  free(copy);
  return d;
}

/* dcg.cc:41-57 + queryresults.cc:55-62 */
double qro_dcg_query(const float *labels, const double *scores, size_t n,
                     size_t cutoff) {
  const size_t k = eff_cutoff(cutoff);
  const size_t size = k < n ? k : n;
  if (size == 0) return 0.0;
  uint64_t *idx = (uint64_t *)malloc(sizeof(uint64_t) * n);
  qro_rank_by_score(scores, n, idx);
  float *sorted = (float *)malloc(sizeof(float) * size);
  for (size_t i = 0; i < n && i < k; ++i) sorted[i] = labels[idx[i]];
  double d = qro_dcg(sorted, n, cutoff); /* compute_dcg(sorted_l, num_results) */
  free(sorted);
  free(idx);
  return d;
}

/* ndcg.cc:49-58 */
double qro_ndcg_query(const float *labels, const double *scores, size_t n,
                      size_t cutoff) {
  if (n == 0) return 0.0;
  const double idcg = qro_idcg(labels, n, cutoff);
  if (idcg > 0) return qro_dcg_query(labels, scores, n, cutoff) / idcg;
  return 0;
}

/* metric.h:77-106: serial sum in query order, divided by Q */
double qro_eval_dataset(int metric, const float *labels, const double *scores,
                        const uint64_t *qoff, size_t nq, size_t cutoff) {
  if (nq == 0) return 0.0;
  double *per = (double *)malloc(sizeof(double) * nq);
#pragma omp parallel for schedule(dynamic, 64)
  for (size_t q = 0; q < nq; ++q) {
    const size_t o = qoff[q], n = qoff[q + 1] - qoff[q];
    per[q] = metric ? qro_ndcg_query(labels + o, scores + o, n, cutoff)
                    : qro_dcg_query(labels + o, scores + o, n, cutoff);
  }
  double avg = 0.0;
  for (size_t q = 0; q < nq; ++q) avg += per[q]; /* same order as the serial loop */
  free(per);
  avg /= (double)nq;
  return avg;
}

/* symmatrix.h:28: packed upper-triangular index */
static size_t sm2v(size_t i, size_t j, size_t size) {
  return i * size - (i - 1) * i / 2 + j - i;
}
static size_t sym_at(size_t i, size_t j, size_t size) {
  return i < j ? sm2v(i, j, size) : sm2v(j, i, size);
}

/* ndcg.cc:60-93 (metric=1) / dcg.cc:59-83 (metric=0) */
void qro_jacobian(int metric, const float *sl, size_t n, size_t cutoff,
                  double *out) {
  const size_t tri = n * (n + 1) / 2;
  for (size_t i = 0; i < tri; ++i) out[i] = 0.0;
  double idcg = 1.0;
  if (metric) {
    idcg = qro_idcg(sl, n, cutoff);
    if (idcg <= 0.0) return;
  }
  const size_t k = eff_cutoff(cutoff);
  const size_t size = k < n ? k : n;
  for (size_t i = 0; i < size; ++i) {
    for (size_t j = i + 1; j < n; ++j) {
      if (sl[i] != sl[j]) {
        double v;
        if (j < size)
          v = (1.0f / log2((double)(j + 2)) - 1.0f / log2((double)(i + 2))) *
              (pow(2.0, (double)sl[i]) - pow(2.0, (double)sl[j]));
        else
          v = (-1.0f / log2((double)(i + 2))) *
              (pow(2.0, (double)sl[i]) - pow(2.0, (double)sl[j]));
        if (metric) v = v / idcg;
        out[sym_at(i, j, n)] = v;
      }
    }
  }
}

/* ========================================================================== */
/* Pseudo-responses                                                           */
/* ========================================================================== */
/* lambdamart.cc:62-152, sample_presence == NULL */
void qro_lambdas(int metric, const float *labels, const double *scores,
                 const uint64_t *qoff, size_t nq, size_t cutoff_in,
                 double *lambda, double *weight) {
  const size_t cutoff = eff_cutoff(cutoff_in);
#pragma omp parallel for schedule(dynamic, 16)
  for (size_t q = 0; q < nq; ++q) {
    const size_t offset = qoff[q];
    const size_t n = qoff[q + 1] - qoff[q];
    for (size_t j = offset; j < offset + n; ++j) lambda[j] = weight[j] = 0.0;
    if (n == 0) continue;
    /* RankedResults ctor, rankedresults.cc:27-41 */
    uint64_t *unmap = (uint64_t *)malloc(sizeof(uint64_t) * n);
    float *sl = (float *)malloc(sizeof(float) * n);
    qro_rank_by_score(scores + offset, n, unmap);
    for (size_t i = 0; i < n; ++i) sl[i] = labels[offset + unmap[i]];
    double *jac = (double *)malloc(sizeof(double) * (n * (n + 1) / 2));
    qro_jacobian(metric, sl, n, cutoff_in, jac);
    for (size_t j = 0; j < n; ++j) {
      const float jl = sl[j];
      const size_t j_abs = offset + unmap[j];
      for (size_t k = 0; k < n; ++k) {
        const size_t k_abs = offset + unmap[k];
        if (k != j) {
          if (j >= cutoff && k >= cutoff) break;
          const float kl = sl[k];
          if (jl > kl) {
            const double deltandcg = fabs(jac[sym_at(j, k, n)]);
            const double rho = 1.0 / (1.0 + exp(scores[j_abs] - scores[k_abs]));
            const double lam = rho * deltandcg;
            const double delta = rho * (1.0 - rho) * deltandcg;
            lambda[j_abs] += lam;
            lambda[k_abs] -= lam;
            weight[j_abs] += delta;
            weight[k_abs] += delta;
          }
        }
      }
    }
    free(jac);
    free(sl);
    free(unmap);
  }
}

/* mart.cc:418-431 */
void qro_residuals(const float *labels, const double *scores, size_t n,
                   double *out) {
  for (size_t i = 0; i < n; ++i) out[i] = labels[i] - scores[i];
}

/* ========================================================================== */
/* Binning                                                                    */
/* ========================================================================== */
/* radix.cc:28-30 */
static uint32_t flipf(uint32_t x) {
  return x ^ ((uint32_t)(-(int32_t)(x >> 31)) | 0x80000000u);
}

/* radix.cc:35-73: stable ascending argsort by the flipped bit pattern.  Two   */
/* 16-bit LSD counting passes, written here as plain stable counting sorts.   */
void qro_argsort_f32(const float *v, size_t n, uint64_t *idx) {
  uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
  size_t *cnt = (size_t *)calloc(65537, sizeof(size_t));
  for (size_t i = 0; i < n; ++i) {
    uint32_t u;
    memcpy(&u, &v[i], 4);
    key[i] = flipf(u);
  }
  for (size_t i = 0; i < n; ++i) cnt[(key[i] & 0xFFFF) + 1]++;
  for (size_t i = 0; i < 65536; ++i) cnt[i + 1] += cnt[i];
  for (size_t i = 0; i < n; ++i) tmp[cnt[key[i] & 0xFFFF]++] = i;
  memset(cnt, 0, 65537 * sizeof(size_t));
  for (size_t i = 0; i < n; ++i) cnt[(key[i] >> 16) + 1]++;
  for (size_t i = 0; i < 65536; ++i) cnt[i + 1] += cnt[i];
  for (size_t i = 0; i < n; ++i) idx[cnt[key[tmp[i]] >> 16]++] = tmp[i];
  free(cnt);
  free(tmp);
  free(key);
}

/* mart.cc:136-170 */
void qro_thresholds(const float *colmajor, size_t N, size_t F,
                    size_t nthresholds, float *thr, uint64_t *thr_size,
                    size_t cap) {
#pragma omp parallel for
  for (size_t f = 0; f < F; ++f) {
    const float *x = colmajor + f * N;
    float *out = thr + f * cap;
    uint64_t *idx = (uint64_t *)malloc(sizeof(uint64_t) * (N ? N : 1));
    qro_argsort_f32(x, N, idx);
    float *uniqs = (float *)malloc(
        sizeof(float) * (nthresholds == 0 ? N + 1 : nthresholds + 1));
    size_t us = 0;
    uniqs[us++] = x[idx[0]];
    for (size_t j = 1; j < N && (nthresholds == 0 || us != nthresholds + 1); ++j) {
      const float fv = x[idx[j]];
      if (uniqs[us - 1] < fv) uniqs[us++] = fv;
    }
    if (us <= nthresholds || nthresholds == 0) {
      uniqs[us++] = FLT_MAX;
      thr_size[f] = us;
      memcpy(out, uniqs, sizeof(float) * us);
    } else {
      thr_size[f] = nthresholds + 1;
      float t = x[idx[0]];
      const float step = (float)fabs(x[idx[N - 1]] - t) / nthresholds;
      for (size_t j = 0; j != nthresholds; t += step) out[j++] = t;
      out[nthresholds] = FLT_MAX;
    }
    for (size_t j = thr_size[f]; j < cap; ++j) out[j] = FLT_MAX;
    free(uniqs);
    free(idx);
  }
}

/* rtnode_histogram.cc:227-253 */
void qro_binmap(const float *colmajor, size_t N, size_t F, const float *thr,
                const uint64_t *thr_size, size_t cap, uint32_t *stmap,
                uint64_t *count0) {
#pragma omp parallel for
  for (size_t f = 0; f < F; ++f) {
    const float *x = colmajor + f * N;
    const float *th = thr + f * cap;
    uint64_t *idx = (uint64_t *)malloc(sizeof(uint64_t) * (N ? N : 1));
    qro_argsort_f32(x, N, idx);
    size_t last = (size_t)-1, j;
    for (size_t t = 0; t < thr_size[f]; ++t) {
      for (j = last + 1; j < N; ++j) {
        const size_t k = idx[j];
        if (x[k] > th[t]) break;
        stmap[f * N + k] = (uint32_t)t;
      }
      last = j - 1;
      count0[f * cap + t] = j;
    }
    free(idx);
  }
}

/* ========================================================================== */
/* Histograms                                                                 */
/* ========================================================================== */
/* rtnode_histogram.cc:41-70 (child ctor) and :172-204 (update): identical     */
/* arithmetic -- per feature, scatter-add in sample order, then sequential     */
/* prefix sums; squares_sum_ serial in sample order.                           */
double qro_hist_build(const uint32_t *stmap, size_t N, size_t F,
                      const uint64_t *thr_size, size_t cap,
                      const double *labels, const uint64_t *sampleids,
                      size_t ns, double *sum, uint64_t *count) {
#pragma omp parallel for
  for (size_t f = 0; f < F; ++f) {
    double *s = sum + f * cap;
    uint64_t *c = count + f * cap;
    const uint32_t *m = stmap + f * N;
    for (size_t t = 0; t < thr_size[f]; ++t) {
      s[t] = 0.0;
      c[t] = 0;
    }
    for (size_t i = 0; i < ns; ++i) {
      const size_t d = sampleids ? sampleids[i] : i;
      const size_t t = m[d];
      s[t] += labels[d];
      c[t]++;
    }
    for (size_t t = 1; t < thr_size[f]; ++t) {
      s[t] += s[t - 1];
      c[t] += c[t - 1];
    }
  }
  double ss = 0.0;
  for (size_t i = 0; i < ns; ++i) {
    const size_t d = sampleids ? sampleids[i] : i;
    ss += labels[d] * labels[d];
  }
  return ss;
}

/* rtnode_histogram.cc:72-87, 206-217 */
void qro_hist_subtract(size_t F, const uint64_t *thr_size, size_t cap,
                       const double *psum, const uint64_t *pcount,
                       const double *lsum, const uint64_t *lcount, double *rsum,
                       uint64_t *rcount) {
#pragma omp parallel for
  for (size_t f = 0; f < F; ++f)
    for (size_t t = 0; t < thr_size[f]; ++t) {
      rsum[f * cap + t] = psum[f * cap + t] - lsum[f * cap + t];
      rcount[f * cap + t] = pcount[f * cap + t] - lcount[f * cap + t];
    }
}

/* ========================================================================== */
/* Split scan: rt.cc:257-312.  Lexicographically first (f,t) attaining the     */
/* maximum (per-thread strict `>` over a contiguous chunk, then thread-order   */
/* strict `>`), initial best -1.                                               */
/* ========================================================================== */
void qro_split_find(size_t f0, size_t f1, const uint64_t *thr_size, size_t cap,
                    const double *sum, const uint64_t *count, uint64_t minls,
                    qro_split_t *out) {
  const size_t nf = f1 > f0 ? f1 - f0 : 0;
  double *bs = (double *)malloc(sizeof(double) * (nf ? nf : 1));
  uint64_t *bt = (uint64_t *)malloc(sizeof(uint64_t) * (nf ? nf : 1));
#pragma omp parallel for
  for (size_t i = 0; i < nf; ++i) {
    const size_t f = f0 + i;
    const double *sl = sum + f * cap;
    const uint64_t *sc = count + f * cap;
    const size_t ts = thr_size[f];
    const double s = sl[ts - 1];
    const uint64_t c = sc[ts - 1];
    double best = -1;
    uint64_t bestt = UINT64_MAX;
    for (size_t t = 0; t < ts; ++t) {
      const uint64_t lc = sc[t];
      const uint64_t rc = c - lc;
      if (lc >= minls && rc >= minls) {
        const double ls = sl[t];
        const double rs = s - ls;
        const double score = ls * ls / (double)lc + rs * rs / (double)rc;
        if (score > best) {
          best = score;
          bestt = t;
        }
      }
    }
    bs[i] = best;
    bt[i] = bestt;
  }
  out->score = -1;
  out->feature = UINT64_MAX;
  out->thr_id = UINT64_MAX;
  out->lcount = out->rcount = 0;
  for (size_t i = 0; i < nf; ++i)
    if (bs[i] > out->score) {
      out->score = bs[i];
      out->feature = f0 + i;
      out->thr_id = bt[i];
    }
  if (out->feature != UINT64_MAX) {
    const size_t f = out->feature;
    const uint64_t c = count[f * cap + thr_size[f] - 1];
    out->lcount = count[f * cap + out->thr_id];
    out->rcount = c - out->lcount;
  }
  free(bs);
  free(bt);
}

/* ========================================================================== */
/* Leaf-wise tree: rt.cc:49-90, 154-160, 209-362                               */
/* ========================================================================== */
typedef struct {
  uint64_t *samples;
  size_t ns, cap; /* cap: entries allocated (the histogram's count) */
  double *hsum;
  uint64_t *hcount;
  double ss; /* squares_sum_ */
} live_t;

/* RTNode(sampleids, hist): rtnode.h:97-107 */
static void node_from_hist(qro_node_t *nd, const live_t *lv, size_t cap,
                           const uint64_t *thr_size) {
  const size_t last = thr_size[0] - 1;
  (void)cap;
  nd->feature = -1;
  nd->thr_id = -1;
  nd->threshold = 0.0f;
  nd->left = nd->right = -1;
  nd->nsamples = lv->hcount[last];
  const double sumlabel = lv->hsum[last];
  nd->value = nd->nsamples ? sumlabel / (double)nd->nsamples : 0.0;
  nd->deviance = lv->ss - sumlabel * sumlabel / nd->nsamples; /* pow(sum,2) */
}

/* maxheap.h:58-88 */
typedef struct {
  double key;
  int32_t val;
} hitem_t;
typedef struct {
  hitem_t *arr;
  size_t size, maxsize;
} heap_t;
static void heap_init(heap_t *h, size_t init) {
  h->maxsize = init + 2;
  h->arr = (hitem_t *)malloc(sizeof(hitem_t) * h->maxsize);
  h->size = 0;
  h->arr[0].key = DBL_MAX;
  h->arr[0].val = -1;
}
static void heap_push(heap_t *h, double key, int32_t val) {
  if (++h->size == h->maxsize) {
    h->maxsize = 2 * h->maxsize + 1;
    h->arr = (hitem_t *)realloc(h->arr, sizeof(hitem_t) * h->maxsize);
  }
  size_t p = h->size;
  while (key > h->arr[p >> 1].key) {
    h->arr[p] = h->arr[p >> 1];
    p >>= 1;
  }
  h->arr[p].key = key;
  h->arr[p].val = val;
}
static void heap_pop(heap_t *h) {
  const hitem_t last = h->arr[h->size--];
  size_t child, p = 1;
  while (p << 1 <= h->size) {
    child = p << 1;
    if (child < h->size && h->arr[child + 1].key > h->arr[child].key) ++child;
    if (last.key < h->arr[child].key)
      h->arr[p] = h->arr[child];
    else
      break;
    p = child;
  }
  h->arr[p] = last;
}

/* test hooks: the heap and the packed index as the tree / jacobian code uses them,
 * pinned against the reference's own headers by tests/test_oracle_vs_ref.py */
void qro_heap_trace(const double *keys, const int32_t *ops, size_t n, size_t initsize,
                    int32_t *top_out, uint64_t *size_out) {
  heap_t h;
  heap_init(&h, initsize);
  for (size_t i = 0; i < n; ++i) {
    if (ops[i] >= 0)
      heap_push(&h, keys[i], ops[i]);
    else if (h.size != 0)
      heap_pop(&h);
    top_out[i] = h.size != 0 ? h.arr[1].val : -1;
    size_out[i] = h.size;
  }
  free(h.arr);
}
void qro_sym_index(size_t size, uint64_t *out) {
  for (size_t i = 0; i < size; ++i)
    for (size_t j = 0; j < size; ++j) out[i * size + j] = sym_at(i, j, size);
}

static void live_free(live_t *lv) {
  free(lv->hsum);
  free(lv->hcount);
  lv->hsum = NULL;
  lv->hcount = NULL;
}

/* A child's sample list is sized by the histogram's count and filled by the raw
 * comparison (rt.cc:314-332 does the same): the two agree by construction of the
 * bin map.  When they do not, say which documents and which of the two views moved. */
static void partition_mismatch(const qro_train_data_t *d, const live_t *lv, int32_t node,
                               size_t bf, size_t bt, float thr, uint64_t lcount,
                               uint64_t rcount, size_t lsize, size_t rsize) {
  char who[512];
  size_t off = 0, shown = 0;
  who[0] = 0;
  const float *x = d->colmajor + bf * d->N;
  for (size_t i = 0; i < lv->ns && shown < 4; ++i) {
    const uint64_t s = lv->samples[i];
    if (s >= d->N) {
      off += (size_t)snprintf(who + off, sizeof(who) - off, " list[%zu]=%llu out of range;", i,
                              (unsigned long long)s);
      ++shown;
      continue;
    }
    const int raw = x[s] <= thr, bin = d->stmap[bf * d->N + s] <= bt;
    if (raw != bin) {
      uint32_t xb;
      memcpy(&xb, &x[s], 4);
      off += (size_t)snprintf(who + off, sizeof(who) - off, " doc %llu: x bits %08x slot %u;",
                              (unsigned long long)s, xb, d->stmap[bf * d->N + s]);
      ++shown;
    }
    if (off >= sizeof(who)) break;
  }
  uint32_t tb;
  memcpy(&tb, &thr, 4);
  qro_event("split of node %d at feature %zu slot %zu (threshold bits %08x): the histogram counts "
            "%llu | %llu documents, the raw comparison over the node's list of %zu sends %zu | %zu;%s",
            (int)node, bf, bt, tb, (unsigned long long)lcount, (unsigned long long)rcount, lv->ns,
            lsize, rsize, shown ? who : " every listed document agrees with its slot (the list itself changed)");
}

/* RegressionTree::split, rt.cc:209-362 (max_features == 1) */
static int tree_split(const qro_train_data_t *d, const double *labels,
                      uint64_t minls, qro_node_t *nodes, live_t *live,
                      size_t *nnodes, int32_t node, int is_root,
                      qro_split_t *rec) {
  qro_node_t *nd = &nodes[node];
  if (!(nd->deviance > 0.0f)) return 0;
  live_t *lv = &live[node];
  qro_split_t sp;
  qro_split_find(0, d->F, d->thr_size, d->cap, lv->hsum, lv->hcount, minls, &sp);
  if (sp.score == -1) return 0;
  const size_t bf = sp.feature, bt = sp.thr_id;
  const float best_threshold = d->thr[bf * d->cap + bt];
  const uint64_t lcount = sp.lcount, rcount = sp.rcount;
  uint64_t *ls = qro_list_alloc(lcount);
  uint64_t *rs = qro_list_alloc(rcount);
  size_t lsize = 0, rsize = 0;
  const float *x = d->colmajor + bf * d->N;
  for (size_t i = 0; i < lv->ns; ++i) {
    const uint64_t s = lv->samples[i];
    if (s >= d->N) continue; /* (self-check below) */
    if (x[s] <= best_threshold) {
      if (lsize < lcount) ls[lsize] = s;
      ++lsize;
    } else {
      if (rsize < rcount) rs[rsize] = s;
      ++rsize;
    }
  }
  if (lsize != lcount || rsize != rcount) {
    partition_mismatch(d, lv, node, bf, bt, best_threshold, lcount, rcount, lsize, rsize);
    if (lsize > lcount) lsize = lcount;
    if (rsize > rcount) rsize = rcount;
  }
  const size_t hs = d->F * d->cap;
  const int32_t li = (int32_t)(*nnodes), ri = li + 1;
  *nnodes += 2;
  live_t *ll = &live[li], *rl = &live[ri];
  ll->samples = ls;
  ll->ns = lsize;
  ll->cap = lcount;
  qro_list_seal(ls, lcount);
  qro_list_seal(rs, rcount);
  ll->hsum = (double *)malloc(sizeof(double) * hs);
  ll->hcount = (uint64_t *)malloc(sizeof(uint64_t) * hs);
  ll->ss = qro_hist_build(d->stmap, d->N, d->F, d->thr_size, d->cap, labels, ls,
                          lsize, ll->hsum, ll->hcount);
  rl->samples = rs;
  rl->ns = rsize;
  rl->cap = rcount;
  if (is_root) { /* rt.cc:340-341: new object */
    rl->hsum = (double *)malloc(sizeof(double) * hs);
    rl->hcount = (uint64_t *)malloc(sizeof(uint64_t) * hs);
    qro_hist_subtract(d->F, d->thr_size, d->cap, lv->hsum, lv->hcount, ll->hsum,
                      ll->hcount, rl->hsum, rl->hcount);
  } else { /* rt.cc:343-346: transform parent in place */
    qro_hist_subtract(d->F, d->thr_size, d->cap, lv->hsum, lv->hcount, ll->hsum,
                      ll->hcount, lv->hsum, lv->hcount);
    rl->hsum = lv->hsum;
    rl->hcount = lv->hcount;
    lv->hsum = NULL;
    lv->hcount = NULL;
  }
  rl->ss = lv->ss - ll->ss;
  nd->feature = (int32_t)bf;
  nd->thr_id = (int32_t)bt;
  nd->threshold = best_threshold;
  nd->left = li;
  nd->right = ri;
  node_from_hist(&nodes[li], ll, d->cap, d->thr_size);
  node_from_hist(&nodes[ri], rl, d->cap, d->thr_size);
  if (rec) *rec = sp;
  return 1;
}

/* rtnode.cc:34-46 */
static void save_leaves(const qro_node_t *nodes, int32_t n, int32_t *leaf_nodes,
                        size_t *nl) {
  if (nodes[n].feature < 0)
    leaf_nodes[(*nl)++] = n;
  else {
    save_leaves(nodes, nodes[n].left, leaf_nodes, nl);
    save_leaves(nodes, nodes[n].right, leaf_nodes, nl);
  }
}

size_t qro_tree_fit(const qro_train_data_t *d, const double *labels,
                    size_t nleaves, uint64_t minls, qro_node_t *nodes,
                    int32_t *leaf_of_doc, int32_t *leaf_nodes,
                    size_t *nleaves_out, qro_split_t *split_log,
                    size_t *nsplits_out) {
  const size_t maxnodes = 2 * nleaves + 1;
  const size_t hs = d->F * d->cap;
  live_t *live = (live_t *)calloc(maxnodes, sizeof(live_t));
  size_t nnodes = 1, nsplits = 0, taken = 0;
  /* root histogram: hist_->update(pseudoresponses_, n, sampleids), mart.cc:335 */
  live[0].ns = d->N;
  live[0].samples = qro_list_alloc(d->N);
  live[0].cap = d->N;
  for (size_t i = 0; i < d->N; ++i) live[0].samples[i] = i;
  qro_list_seal(live[0].samples, d->N);
  live[0].hsum = (double *)malloc(sizeof(double) * hs);
  live[0].hcount = (uint64_t *)malloc(sizeof(uint64_t) * hs);
  live[0].ss = qro_hist_build(d->stmap, d->N, d->F, d->thr_size, d->cap, labels,
                              NULL, d->N, live[0].hsum, live[0].hcount);
  node_from_hist(&nodes[0], &live[0], d->cap, d->thr_size);
  heap_t heap;
  heap_init(&heap, nleaves);
  qro_split_t rec;
  if (tree_split(d, labels, minls, nodes, live, &nnodes, 0, 1, &rec)) {
    if (split_log) split_log[nsplits] = rec;
    nsplits++;
    heap_push(&heap, nodes[nodes[0].left].deviance, nodes[0].left);
    heap_push(&heap, nodes[nodes[0].right].deviance, nodes[0].right);
  }
  while (heap.size != 0 && (nleaves == 0 || taken + heap.size < nleaves)) {
    const int32_t node = heap.arr[1].val;
    heap_pop(&heap);
    if (tree_split(d, labels, minls, nodes, live, &nnodes, node, 0, &rec)) {
      if (split_log) split_log[nsplits] = rec;
      nsplits++;
      heap_push(&heap, nodes[nodes[node].left].deviance, nodes[node].left);
      heap_push(&heap, nodes[nodes[node].right].deviance, nodes[node].right);
    } else
      ++taken;
    live_free(&live[node]);
  }
  free(heap.arr);
  size_t nl = 0;
  save_leaves(nodes, 0, leaf_nodes, &nl);
  for (size_t i = 0; i < d->N; ++i) leaf_of_doc[i] = -1;
  for (size_t l = 0; l < nl; ++l) {
    const live_t *lv = &live[leaf_nodes[l]];
    if (lv->ns != nodes[leaf_nodes[l]].nsamples)
      qro_event("leaf node %d: the histogram counts %llu documents, its list holds %zu",
                (int)leaf_nodes[l], (unsigned long long)nodes[leaf_nodes[l]].nsamples, lv->ns);
    for (size_t i = 0; i < lv->ns; ++i) {
      const uint64_t s = lv->samples[i];
      if (s >= d->N || leaf_of_doc[s] != -1) { /* a list entry that is not this leaf's own */
        qro_event("leaf node %d: list[%zu] = %llu is %s", (int)leaf_nodes[l], i, (unsigned long long)s,
                  s >= d->N ? "out of range" : "a document another leaf's list holds too");
        continue;
      }
      leaf_of_doc[s] = (int32_t)l;
    }
  }
  for (size_t i = 0; i < nnodes; ++i) {
    qro_list_free(live[i].samples, live[i].cap);
    live_free(&live[i]);
  }
  free(live);
  *nleaves_out = nl;
  if (nsplits_out) *nsplits_out = nsplits;
  return nnodes;
}

/* ========================================================================== */
/* Oblivious tree: ot.cc:32-201                                                */
/* ========================================================================== */
#define OT_INVALID (-DBL_MAX) /* ot.h: const double invalid = -DBL_MAX */

size_t qro_oblivious_fit(const qro_train_data_t *d, const double *labels,
                         size_t depth_max, uint64_t minls, qro_node_t *nodes,
                         int32_t *leaf_of_doc, int32_t *leaf_nodes,
                         size_t *nleaves_out, qro_split_t *split_log,
                         size_t *nsplits_out) {
  const size_t maxnodes = ((size_t)1 << (depth_max + 1)) - 1;
  const size_t hs = d->F * d->cap;
  live_t *live = (live_t *)calloc(maxnodes, sizeof(live_t));
  char *present = (char *)calloc(maxnodes, 1);
  for (size_t i = 0; i < maxnodes; ++i) {
    nodes[i].feature = -2; /* absent */
    nodes[i].left = nodes[i].right = -1;
    nodes[i].thr_id = -1;
    nodes[i].threshold = 0;
    nodes[i].value = 0;
    nodes[i].deviance = 0;
    nodes[i].nsamples = 0;
  }
  live[0].ns = d->N;
  live[0].samples = qro_list_alloc(d->N);
  live[0].cap = d->N;
  for (size_t i = 0; i < d->N; ++i) live[0].samples[i] = i;
  qro_list_seal(live[0].samples, d->N);
  live[0].hsum = (double *)malloc(sizeof(double) * hs);
  live[0].hcount = (uint64_t *)malloc(sizeof(uint64_t) * hs);
  live[0].ss = qro_hist_build(d->stmap, d->N, d->F, d->thr_size, d->cap, labels,
                              NULL, d->N, live[0].hsum, live[0].hcount);
  node_from_hist(&nodes[0], &live[0], d->cap, d->thr_size);
  present[0] = 1;
  double *sum_scores = (double *)malloc(sizeof(double) * hs);
  size_t nsplits = 0;
  for (size_t depth = 0; depth < depth_max; ++depth) {
    const size_t lbegin = ((size_t)1 << depth) - 1;
    const size_t lend = ((size_t)1 << (depth + 1)) - 1;
    for (size_t i = 0; i < hs; ++i) sum_scores[i] = 0.0;
    /* fill(), ot.cc:177-201 */
    for (size_t i = lbegin; i < lend; ++i) {
      const live_t *lv = &live[i];
#pragma omp parallel for
      for (size_t f = 0; f < d->F; ++f) {
        const double *sl = lv->hsum + f * d->cap;
        const uint64_t *sc = lv->hcount + f * d->cap;
        const size_t ts = d->thr_size[f];
        const double s = sl[ts - 1];
        const uint64_t c = sc[ts - 1];
        double *sv = sum_scores + f * d->cap;
        for (size_t t = 0; t < ts; ++t)
          if (sv[t] != OT_INVALID) {
            const uint64_t lc = sc[t];
            const uint64_t rc = c - lc;
            if (lc >= minls && rc >= minls) {
              const double ls = sl[t];
              const double rs = s - ls;
              sv[t] += ls * ls / lc + rs * rs / rc;
            } else
              sv[t] = OT_INVALID;
          }
      }
    }
    /* argmax, ot.cc:67-92: strict >, initial 0.0 */
    double max_score = 0.0;
    size_t bf = SIZE_MAX, bt = SIZE_MAX;
    for (size_t f = 0; f < d->F; ++f)
      for (size_t t = 0; t < d->thr_size[f]; ++t) {
        const double v = sum_scores[f * d->cap + t];
        if (v != OT_INVALID && v > max_score) {
          max_score = v;
          bf = f;
          bt = t;
        }
      }
    if (max_score == OT_INVALID || max_score == 0.0) break;
    if (split_log) {
      split_log[nsplits].score = max_score;
      split_log[nsplits].feature = bf;
      split_log[nsplits].thr_id = bt;
      split_log[nsplits].lcount = split_log[nsplits].rcount = 0;
    }
    nsplits++;
    const float best_threshold = d->thr[bf * d->cap + bt];
    const float *x = d->colmajor + bf * d->N;
    for (size_t i = lbegin; i < lend; ++i) {
      live_t *lv = &live[i];
      qro_node_t *nd = &nodes[i];
      const size_t lastt = d->thr_size[bf] - 1;
      const uint64_t lcount = lv->hcount[bf * d->cap + bt];
      const uint64_t rcount = lv->hcount[bf * d->cap + lastt] - lcount;
      uint64_t *ls = qro_list_alloc(lcount);
      uint64_t *rs = qro_list_alloc(rcount);
      size_t lsize = 0, rsize = 0;
      for (size_t j = 0; j < lv->ns; ++j) {
        const uint64_t k = lv->samples[j];
        if (k >= d->N) continue; /* (self-check below) */
        if (x[k] <= best_threshold) {
          if (lsize < lcount) ls[lsize] = k;
          ++lsize;
        } else {
          if (rsize < rcount) rs[rsize] = k;
          ++rsize;
        }
      }
      if (lsize != lcount || rsize != rcount) {
        partition_mismatch(d, lv, (int32_t)i, bf, bt, best_threshold, lcount, rcount, lsize, rsize);
        if (lsize > lcount) lsize = lcount;
        if (rsize > rcount) rsize = rcount;
      }
      const size_t li = 2 * i + 1, ri = 2 * i + 2;
      live_t *ll = &live[li], *rl = &live[ri];
      ll->samples = ls;
      ll->ns = lsize;
      ll->cap = lcount;
      rl->samples = rs;
      rl->ns = rsize;
      rl->cap = rcount;
      qro_list_seal(ls, lcount);
      qro_list_seal(rs, rcount);
      present[li] = present[ri] = 1;
      if (depth != depth_max - 1) {
        ll->hsum = (double *)malloc(sizeof(double) * hs);
        ll->hcount = (uint64_t *)malloc(sizeof(uint64_t) * hs);
        ll->ss = qro_hist_build(d->stmap, d->N, d->F, d->thr_size, d->cap,
                                labels, ls, lsize, ll->hsum, ll->hcount);
        rl->hsum = (double *)malloc(sizeof(double) * hs);
        rl->hcount = (uint64_t *)malloc(sizeof(uint64_t) * hs);
        qro_hist_subtract(d->F, d->thr_size, d->cap, lv->hsum, lv->hcount,
                          ll->hsum, ll->hcount, rl->hsum, rl->hcount);
        rl->ss = lv->ss - ll->ss;
        node_from_hist(&nodes[li], ll, d->cap, d->thr_size);
        node_from_hist(&nodes[ri], rl, d->cap, d->thr_size);
      } else { /* ot.cc:141-149 */
        const double lsum = lv->hsum[bf * d->cap + bt];
        const double rsum = lv->hsum[bf * d->cap + lastt] - lsum;
        qro_node_t *a = &nodes[li], *b = &nodes[ri];
        a->feature = b->feature = -1;
        a->nsamples = lsize;
        b->nsamples = rsize;
        a->value = lsum / lsize;
        b->value = rsum / rsize;
        a->deviance = b->deviance = 0.0;
      }
      nd->feature = (int32_t)bf;
      nd->thr_id = (int32_t)bt;
      nd->threshold = best_threshold;
      nd->left = (int32_t)li;
      nd->right = (int32_t)ri;
    }
  }
  free(sum_scores);
  size_t nl = 0;
  save_leaves(nodes, 0, leaf_nodes, &nl);
  for (size_t i = 0; i < d->N; ++i) leaf_of_doc[i] = -1;
  for (size_t l = 0; l < nl; ++l) {
    const live_t *lv = &live[leaf_nodes[l]];
    if (lv->ns != nodes[leaf_nodes[l]].nsamples)
      qro_event("leaf node %d: the histogram counts %llu documents, its list holds %zu",
                (int)leaf_nodes[l], (unsigned long long)nodes[leaf_nodes[l]].nsamples, lv->ns);
    for (size_t i = 0; i < lv->ns; ++i) {
      const uint64_t s = lv->samples[i];
      if (s >= d->N || leaf_of_doc[s] != -1) { /* a list entry that is not this leaf's own */
        qro_event("leaf node %d: list[%zu] = %llu is %s", (int)leaf_nodes[l], i, (unsigned long long)s,
                  s >= d->N ? "out of range" : "a document another leaf's list holds too");
        continue;
      }
      leaf_of_doc[s] = (int32_t)l;
    }
  }
  size_t used = 0;
  for (size_t i = 0; i < maxnodes; ++i) {
    if (present[i]) used = i + 1;
    qro_list_free(live[i].samples, live[i].cap);
    live_free(&live[i]);
  }
  free(live);
  free(present);
  *nleaves_out = nl;
  if (nsplits_out) *nsplits_out = nsplits;
  return used;
}

/* rt.cc:165-207.  Leaf sample lists are ascending doc ids (stable partition  */
/* of the identity list), so iterating docs in order reproduces the order.    */
void qro_update_output(qro_node_t *nodes, const int32_t *leaf_nodes,
                       size_t nleaves, const int32_t *leaf_of_doc, size_t N,
                       const double *pseudo, const double *weights) {
  double *s1 = (double *)calloc(nleaves ? nleaves : 1, sizeof(double));
  double *s2 = (double *)calloc(nleaves ? nleaves : 1, sizeof(double));
  uint64_t *cn = (uint64_t *)calloc(nleaves ? nleaves : 1, sizeof(uint64_t));
  for (size_t i = 0; i < N; ++i) {
    const int32_t l = leaf_of_doc[i];
    if (l < 0) continue;
    s1[l] += pseudo[i];
    if (weights) s2[l] += weights[i];
    cn[l]++;
  }
  for (size_t l = 0; l < nleaves; ++l) {
    qro_node_t *nd = &nodes[leaf_nodes[l]];
    if (cn[l] != nd->nsamples)
      qro_event("leaf node %d: the histogram counts %llu documents, %llu are mapped to it",
                (int)leaf_nodes[l], (unsigned long long)nd->nsamples, (unsigned long long)cn[l]);
    if (weights)
      nd->value = s2[l] >= DBL_EPSILON ? s1[l] / s2[l] : 0.0;
    else
      nd->value = s1[l] / cn[l];
  }
  free(s1);
  free(s2);
  free(cn);
}

/* rtnode.h:134-152 */
double qro_tree_score(const qro_node_t *nodes, const float *x, size_t stride) {
  int32_t n = 0;
  while (nodes[n].feature >= 0)
    n = x[(size_t)nodes[n].feature * stride] <= nodes[n].threshold
            ? nodes[n].left
            : nodes[n].right;
  return nodes[n].value;
}

/* mart.cc:447-468 */
void qro_update_scores(const qro_node_t *nodes, const float *data, size_t N,
                       size_t F, int colmajor, double shrinkage,
                       double *scores) {
#pragma omp parallel for
  for (size_t i = 0; i < N; ++i) {
    const double v = colmajor ? qro_tree_score(nodes, data + i, N)
                              : qro_tree_score(nodes, data + i * F, 1);
    scores[i] += shrinkage * v;
  }
}

/* ========================================================================== */
/* Mart::learn, mart.cc:208-416                                                */
/* ========================================================================== */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int qro_train(const qro_params_t *p, const float *train, const float *labels,
              const uint64_t *qoff, size_t nq, size_t N, size_t F,
              const float *valid, const float *vlabels, const uint64_t *vqoff,
              size_t vnq, size_t vN, qro_model_t *out) {
  memset(out, 0, sizeof(*out));
  const int lambda = (p->algo == 1 || p->algo == 3);
  const int obliv = (p->algo >= 2);
  const size_t L = obliv ? ((size_t)1 << p->depth) : p->nleaves;
  const size_t max_nodes = obliv ? (((size_t)1 << (p->depth + 1)) - 1) : 2 * L + 1;
  /* VerticalDataset, vertical_dataset.cc:45-51 */
  float *col = (float *)qro_guard_alloc(sizeof(float) * N * F);
#pragma omp parallel for
  for (size_t i = 0; i < N; ++i)
    for (size_t f = 0; f < F; ++f) col[f * N + i] = train[i * F + f];
  qro_guard_seal(col, sizeof(float) * N * F); /* (QRO_GUARD=1: read-only from here on, hashed) */
  const size_t cap = p->nthresholds ? p->nthresholds + 1 : N + 1;
  float *thr = (float *)malloc(sizeof(float) * F * cap);
  uint64_t *thr_size = (uint64_t *)malloc(sizeof(uint64_t) * F);
  uint32_t *stmap = (uint32_t *)qro_guard_alloc(sizeof(uint32_t) * F * N);
  uint64_t *count0 = (uint64_t *)malloc(sizeof(uint64_t) * F * cap);
  qro_thresholds(col, N, F, p->nthresholds, thr, thr_size, cap);
  qro_binmap(col, N, F, thr, thr_size, cap, stmap, count0);
  free(count0);
  qro_guard_seal(stmap, sizeof(uint32_t) * F * N);
  qro_train_data_t d = {N, F, cap, col, stmap, thr, thr_size};
  double *scores = (double *)calloc(N, sizeof(double));
  double *vscores = valid ? (double *)calloc(vN ? vN : 1, sizeof(double)) : NULL;
  double *pseudo = (double *)calloc(N, sizeof(double));
  double *weights = lambda ? (double *)calloc(N, sizeof(double)) : NULL;
  int32_t *leaf_of_doc = (int32_t *)malloc(sizeof(int32_t) * N);
  int32_t *leaf_nodes = (int32_t *)malloc(sizeof(int32_t) * (L + 1));
  out->max_nodes = max_nodes;
  out->nodes = (qro_node_t *)calloc(p->ntrees * max_nodes, sizeof(qro_node_t));
  out->nnodes = (uint64_t *)calloc(p->ntrees, sizeof(uint64_t));
  out->train_metric = (double *)calloc(p->ntrees, sizeof(double));
  out->valid_metric = (double *)calloc(p->ntrees, sizeof(double));
  out->iter_seconds = (double *)calloc(p->ntrees, sizeof(double));
  double best_valid = -DBL_MAX, best_train = -DBL_MAX;
  size_t best_model = 0, size = 0;
  for (size_t m = 0; m < p->ntrees; ++m) {
    if (valid && (p->esr && m > best_model + p->esr)) break;
    const double t0 = now_s();
    if (lambda)
      qro_lambdas(p->metric, labels, scores, qoff, nq, p->cutoff, pseudo, weights);
    else
      qro_residuals(labels, scores, N, pseudo);
    qro_node_t *nodes = out->nodes + m * max_nodes;
    size_t nl = 0, nn;
    if (obliv)
      nn = qro_oblivious_fit(&d, pseudo, p->depth, p->minls, nodes, leaf_of_doc,
                             leaf_nodes, &nl, NULL, NULL);
    else
      nn = qro_tree_fit(&d, pseudo, p->nleaves, p->minls, nodes, leaf_of_doc,
                        leaf_nodes, &nl, NULL, NULL);
    qro_update_output(nodes, leaf_nodes, nl, leaf_of_doc, N, pseudo,
                      lambda ? weights : NULL);
    out->nnodes[m] = nn;
    size = m + 1;
    qro_update_scores(nodes, col, N, F, 1, p->shrinkage, scores);
    const double mt = qro_eval_dataset(p->metric, labels, scores, qoff, nq, p->cutoff);
    out->train_metric[m] = mt;
    if (valid) {
      qro_update_scores(nodes, valid, vN, F, 0, p->shrinkage, vscores);
      const double mv = qro_eval_dataset(p->metric, vlabels, vscores, vqoff, vnq, p->cutoff);
      out->valid_metric[m] = mv;
      if (mv > best_valid) {
        best_train = mt;
        best_valid = mv;
        best_model = size - 1;
      }
    } else if (mt > best_train) {
      best_train = mt;
      best_model = size - 1;
    }
    out->iter_seconds[m] = now_s() - t0;
  }
  out->ntrees_built = size;
  out->ntrees = size;
  if (valid)
    while (out->ntrees > 0 && out->ntrees > best_model + 1) out->ntrees--;
  out->best_model = best_model;
  out->train_scores = scores;
  out->thr = thr;
  out->thr_size = thr_size;
  out->cap = cap;
  free(vscores);
  free(pseudo);
  free(weights);
  free(leaf_of_doc);
  free(leaf_nodes);
  qro_guard_free(stmap, sizeof(uint32_t) * F * N);
  qro_guard_free(col, sizeof(float) * N * F);
  return 0;
}

void qro_model_free(qro_model_t *m) {
  free(m->nodes);
  free(m->nnodes);
  free(m->train_metric);
  free(m->valid_metric);
  free(m->train_scores);
  free(m->thr);
  free(m->thr_size);
  free(m->iter_seconds);
  memset(m, 0, sizeof(*m));
}

/* ========================================================================== */
/* Inference                                                                  */
/* ========================================================================== */
/* ltr_algorithm.cc:44-52 + ensemble.cc:111-118: sum += tree(x) * weight       */
void qro_ensemble_score(const qro_node_t *nodes, const uint64_t *nnodes,
                        size_t ntrees, size_t max_nodes, const double *weights,
                        const float *rowmajor, size_t N, size_t F,
                        double *scores) {
  (void)nnodes;
#pragma omp parallel for
  for (size_t i = 0; i < N; ++i) {
    double sum = 0.0f;
    for (size_t t = 0; t < ntrees; ++t)
      sum += qro_tree_score(nodes + t * max_nodes, rowmajor + i * F, 1) * weights[t];
    scores[i] = sum;
  }
}

/* generate_oblivious.cc:237-324: leafidx |= (v[fid] > thr) << (m-1-i);        */
/* score += tree_weight(f32) * leaf_outputs[tree][leafidx]                     */
void qro_oblivious_score(const uint32_t *feat, const float *thr,
                         const double *leaves, const float *weights,
                         size_t ntrees, size_t depth, const float *rowmajor,
                         size_t N, size_t F, double *scores) {
  const size_t nl = (size_t)1 << depth;
#pragma omp parallel for
  for (size_t i = 0; i < N; ++i) {
    const float *v = rowmajor + i * F;
    double score = 0;
    for (size_t t = 0; t < ntrees; ++t) {
      unsigned leafidx = 0;
      for (size_t l = 0; l < depth; ++l)
        leafidx |= (unsigned)(v[feat[t * depth + l]] > thr[t * depth + l])
                   << (depth - 1 - l);
      score += weights[t] * leaves[t * nl + leafidx];
    }
    scores[i] = score;
  }
}
