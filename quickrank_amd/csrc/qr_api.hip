// qr_api.hip -- implementation of the C-ABI declared in include/qr_hip.h:
// device memory management, the (tiny) host-side arithmetic of Mart::init
// (mart.cc:147-169) and Ndcg::compute_idcg (ndcg.cc:35-47), and the launch
// sequences.  No CPU fallback: every compute entry point needs the gfx950
// device the context was created on.
#include <sys/mman.h>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <atomic>
#include <functional>

#include "qr_internal.h"

static thread_local std::string g_create_err;

// QR_POISON=1 (a debugging aid, tests/tools/abort_hunt.py --poison): every device allocation of the
// library starts as 0xA5 bytes -- a read of something nobody wrote gives the same garbage in every
// process, not zeros in a fresh one and an earlier context's data in a long one.
static bool qr_poison() {
  static const bool on = getenv("QR_POISON") && atoi(getenv("QR_POISON")) != 0;
  return on;
}
template <class T>
static hipError_t dalloc(T **p, size_t n) {
  const hipError_t e = hipMalloc((void **)p, (n ? n : 1) * sizeof(T));
  if (e == hipSuccess && qr_poison()) return hipMemset(*p, 0xA5, (n ? n : 1) * sizeof(T));
  return e;
}
template <class T>
static void dfree(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

#include <dlfcn.h>
#include <sched.h>

extern "C" {

void qr_ctx_destroy(qr_ctx *c);
static int tree_settle(qr_ctx *c, bool keep_scores = false);

// Wait for a read-back: the publishing kernel stores `want` into the pinned block after its
// data (system-scope release).  Polls the word; every so often asks the stream whether it
// has failed or drained, so that a faulted kernel ends the wait with an error and a
// platform where the stores only land at the end of the kernel is merely slower.
static inline void cpu_relax(unsigned spin) {
  // a short spin first (the word usually arrives within microseconds), then yield the core:
  // with --gpus 8 there are eight of these waiting threads on the host
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#endif
  if (spin > 20000u) sched_yield();
}

static int wait_seq_impl(qr_ctx *c, const void *word, const int bytes, const int64_t want, const char *what) {
  auto reached = [&]() {
    return bytes == 4 ? (int64_t)__atomic_load_n((const int32_t *)word, __ATOMIC_ACQUIRE) == want
                      : __atomic_load_n((const int64_t *)word, __ATOMIC_ACQUIRE) == want;
  };
  for (unsigned spin = 1;; ++spin) {
    if (reached()) return QR_OK;
    if ((spin & 1023u) == 0) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) {  // everything enqueued has run
        if (reached()) return QR_OK;
        QR_CHECK(c, hipStreamSynchronize(c->stream));
        if (reached()) return QR_OK;
        c->err = std::string("internal: ") + what + " were not published";
        return QR_ERR_STATE;
      }
      if (e != hipErrorNotReady) QR_CHECK(c, e);
    }
    cpu_relax(spin);
  }
}
static int wait_seq32(qr_ctx *c, const int32_t *word, const int32_t want, const char *what) {
  return wait_seq_impl(c, word, 4, want, what);
}
static int wait_seq64(qr_ctx *c, const int64_t *word, const int64_t want, const char *what) {
  return wait_seq_impl(c, word, 8, want, what);
}

int qr_ctx_create(int device, qr_ctx **out) {
  if (out) *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    g_create_err = "no HIP device visible (this library has no CPU fallback)";
    return QR_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) {
    g_create_err = "device index out of range";
    return QR_ERR_ARG;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_create_err = "hipSetDevice failed";
    return QR_ERR_HIP;
  }
  qr_ctx *c = new qr_ctx();
  c->no_batch = getenv("QR_NO_BATCH") != nullptr;
  c->no_defer = getenv("QR_NO_DEFER_PREP") != nullptr;
  c->no_lazy_scores = getenv("QR_LAZY_SCORES") && atoi(getenv("QR_LAZY_SCORES")) == 0;  // (A/B: tests/test_gpu_parity.py)
  c->obl_own_launches = getenv("QR_OBL_OWN_LAUNCHES") != nullptr;
  c->exact_tail = getenv("QR_EXACT_TAIL") != nullptr;
  c->spec_debug = getenv("QR_SPEC_DEBUG") != nullptr;   // (diagnostic lines on stderr; no code path depends on it)
  if (getenv("QR_LEAF_BY_POSITION")) c->leaf_by_position = true;
  c->x_eager = getenv("QR_X_EAGER") != nullptr;
  if (const char *e = getenv("QR_FUSE_MAX_DOCS")) c->fuse_max_docs = (size_t)std::max(0l, atol(e));
  if (const char *e = getenv("QR_STEPS_HINT")) c->steps_force = atol(e);  // steps to enqueue, whatever the tree
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    c->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    //  the LDS budgets of the fast paths follow the device: a workgroup's opt-in maximum
    // and a CU's total.  gfx950 has 160 KB of both whatever the runtime's property fields say (the
    // kernels were measured there); any other device gets what it reports.
    size_t blk = std::max<size_t>(prop.sharedMemPerBlockOptin, prop.sharedMemPerBlock);
    size_t cu = std::max<size_t>(prop.maxSharedMemoryPerMultiProcessor, blk);
    if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) {
      blk = std::max<size_t>(blk, 160 * 1024);
      cu = std::max<size_t>(cu, 160 * 1024);
    }
    if (blk) c->lds_block = blk;
    if (cu) c->lds_cu = cu;
    if (c->spec_debug)
      fprintf(stderr, "qr: %s, %d CUs, LDS per workgroup %zu (optin %zu, default %zu), per CU %zu\n", prop.gcnArchName,
              c->ncu, c->lds_block, (size_t)prop.sharedMemPerBlockOptin, (size_t)prop.sharedMemPerBlock,
              (size_t)prop.maxSharedMemoryPerMultiProcessor);
  }
  if (hipStreamCreate(&c->stream) != hipSuccess) {
    delete c;
    g_create_err = "hipStreamCreate failed";
    return QR_ERR_HIP;
  }
  c->own_stream = true;
  if (dalloc(&c->d_scalars, 1) != hipSuccess) {
    delete c;
    g_create_err = "hipMalloc failed";
    return QR_ERR_HIP;
  }
  (void)hipMemset(c->d_scalars, 0, sizeof(QrScalars));
  // (coherent = fine-grained: a kernel's stores are visible to the host while it runs)
#ifdef QR_DEBUG_CHECKS
  // (the hunt's build, VERDICT r5 item 1: the block is pages of this context's OWN -- mapped here,
  // registered with the device, and at destruction unregistered, poisoned, made inaccessible and
  // never handed back: a kernel that still writes it faults on the device, a host store through a
  // stale pointer faults on the CPU, and no later allocation of anybody can come to lie there)
  {
    const size_t pb = (sizeof(QrPinned) + 4095) / 4096 * 4096;
    void *m = mmap(nullptr, pb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m != MAP_FAILED && hipHostRegister(m, pb, hipHostRegisterMapped) == hipSuccess) c->h_pin = (QrPinned *)m;
  }
  if (!c->h_pin ||
#else
  if (hipHostMalloc((void **)&c->h_pin, sizeof(QrPinned), hipHostMallocCoherent) != hipSuccess ||
#endif
      hipHostGetDevicePointer((void **)&c->d_pin, c->h_pin, 0) != hipSuccess ||
      hipMalloc((void **)&c->d_prep_part, QR_PREP_WORDS * 8) != hipSuccess ||
      hipMemset(c->d_prep_part, 0, QR_PREP_WORDS * 8) != hipSuccess) {
    qr_ctx_destroy(c);
    g_create_err = "allocating the read-back buffers failed";
    return QR_ERR_HIP;
  }
  memset(c->h_pin, 0, sizeof(QrPinned));
  *out = c;
  return QR_OK;
}

static void free_long(qr_ctx *c, int which);

static void free_train(qr_ctx *c) {
  free_long(c, 0);
  dfree(c->d_raw); dfree(c->d_labels); dfree(c->d_qoff);
  dfree(c->d_scores); dfree(c->d_lambda); dfree(c->d_weight);
  dfree(c->d_idcg); dfree(c->d_qmetric); dfree(c->d_ranks); dfree(c->d_ssq); dfree(c->d_qmax);
  dfree(c->d_blocks); dfree(c->d_lf2gf); dfree(c->d_gf2lf); dfree(c->d_bins); dfree(c->d_bins_fm);
  dfree(c->d_thr); dfree(c->d_thr_size);
  dfree(c->d_woff); dfree(c->d_wthr); dfree(c->d_wbins); dfree(c->d_wbins16);
  if (c->d_wpart) (void)hipFree(c->d_wpart);
  c->d_wpart = nullptr;
  c->wpart_cap = 0;
  dfree(c->d_wchunk); dfree(c->d_wchunk0); dfree(c->d_wtot_s); dfree(c->d_wtot_c); dfree(c->d_wcbest);
  qr_k_exact_free(c);
  c->wchunks = 0;
  c->wide = false;
  c->wcells = 0;
  c->wmax = 0;
  dfree(c->d_order[0]); dfree(c->d_order[1]); dfree(c->d_partials);
  if (c->d_xh) {  // document-sharded: the reduced histogram lives in the exchange buffer
    c->d_red_sum = nullptr;
    c->d_red_cnt = nullptr;
  }
  dfree(c->d_xh); dfree(c->d_xscal); dfree(c->d_xleaf); dfree(c->d_xlevel); dfree(c->d_xb);
  c->xb_len = 0;
  c->xleaf_cap = 0;
  dfree(c->d_red_sum); dfree(c->d_red_cnt);
  dfree(c->d_hsum); dfree(c->d_hcnt); dfree(c->d_featrec); dfree(c->d_featthr); dfree(c->d_lscan_wg);
  dfree(c->d_recs_local); dfree(c->d_recs_all); dfree(c->d_mask);
  dfree(c->d_red_cnt_loc); dfree(c->d_hcnt_loc); dfree(c->d_part_state); dfree(c->d_part_ss); dfree(c->d_lpart_ss); dfree(c->d_lpart_ss2); dfree(c->d_jobsum); dfree(c->d_bpart_state); dfree(c->d_tree); dfree(c->d_tree2); dfree(c->d_leafpart); dfree(c->d_leafb);
  dfree(c->d_lhist_map); dfree(c->d_lpart_map); dfree(c->d_lpartials); dfree(c->d_lhistsum);
  dfree(c->d_lpart_state);
  dfree(c->d_lhist_wg); dfree(c->d_lpart_wg); dfree(c->d_lplan);
  c->lhist_cap = c->lpart_cap = c->lslots_cap = c->lred_nodes = 0;
  dfree(c->d_present); dfree(c->d_sample_count);
  if (c->d_sample_work) (void)hipFree(c->d_sample_work);
  c->d_sample_work = nullptr;
  c->sub_k = 0;
  c->mf_k = 0;
  c->spec_pending = c->spec_scores_enqueued = false;
  c->lazy_scores = false;
  c->steps_hint = 0;
  c->binned = false;
  c->tree_valid = false;
  c->hist_slots = 0;
}
static void free_long(qr_ctx *c, int which) {
  if (c->d_long_flag[which]) (void)hipFree(c->d_long_flag[which]);
  if (c->d_long_list[which]) (void)hipFree(c->d_long_list[which]);
  c->d_long_flag[which] = nullptr;
  c->d_long_list[which] = nullptr;
  c->long_tag[which] = -1;
  if (c->d_qclass[which]) (void)hipFree(c->d_qclass[which]);
  c->d_qclass[which] = nullptr;
  if (c->d_lu_list[which]) (void)hipFree(c->d_lu_list[which]);
  c->d_lu_list[which] = nullptr;
  c->lu_on[which] = false;
  c->h_qclass[which].clear();
}

static void free_valid(qr_ctx *c) {
  free_long(c, 1);
  dfree(c->d_vraw); dfree(c->d_vlabels); dfree(c->d_vqoff); dfree(c->d_vscores);
  dfree(c->d_vidcg); dfree(c->d_vqmetric); dfree(c->d_vranks);
  c->vN = c->vQ = 0;
}

void qr_ctx_destroy(qr_ctx *c) {
  if (!c) return;
  if (c->spec_debug && c->spec_trees)
    fprintf(stderr, "qr: %llu trees with a guessed step count, %llu continued (guess too low), last hint %zu\n",
            (unsigned long long)c->spec_trees, (unsigned long long)c->spec_misses, c->steps_hint);
  if (c->readback_retries)   // (never silent: this is the evidence round 5's hunt was after)
    fprintf(stderr, "qr: %llu re-reads of tree records / scalars that did not fit their sequence number yet\n",
            (unsigned long long)c->readback_retries);
  (void)hipSetDevice(c->device);
  // Teardown order: every stream this context created is drained BEFORE any
  // memory its kernels may touch is released -- the lambda pass's size classes run on the
  // auxiliary streams and write the pinned read-back block's neighbours; they are joined to the
  // main stream by events, but a join that was enqueued and never waited for (an error return
  // between fork and join) would otherwise leave work behind hipHostFree / hipFree.
  for (int i = 0; i < 4; ++i)
    if (c->aux_stream[i]) (void)hipStreamSynchronize(c->aux_stream[i]);
  (void)hipStreamSynchronize(c->stream);
  free_train(c);
  free_valid(c);
  dfree(c->d_lg2); dfree(c->d_ilg2); dfree(c->d_scalars); dfree(c->d_ens); dfree(c->d_ens_w);
  if (c->d_lscratch) (void)hipFree(c->d_lscratch);
  for (int i = 0; i < 4; ++i) {
    if (c->aux_stream[i]) (void)hipStreamDestroy(c->aux_stream[i]);
    if (c->aux_join[i]) (void)hipEventDestroy(c->aux_join[i]);
  }
  if (c->aux_fork) (void)hipEventDestroy(c->aux_fork);
#ifdef QR_DEBUG_CHECKS
  if (c->h_pin) {
    const size_t pb = (sizeof(QrPinned) + 4095) / 4096 * 4096;
    (void)hipHostUnregister(c->h_pin);
    memset((void *)c->h_pin, 0xA5, pb);
    (void)mprotect((void *)c->h_pin, pb, PROT_NONE);   // (leaked on purpose: see qr_ctx_create)
  }
#else
  if (c->h_pin) (void)hipHostFree(c->h_pin);
#endif
  if (c->d_prep_part) (void)hipFree(c->d_prep_part);
  if (c->d_root_wg) (void)hipFree(c->d_root_wg);
  if (c->d_root_scan) (void)hipFree(c->d_root_scan);
  dfree(c->d_keys); dfree(c->d_tied);
  dfree(c->d_obl_feat); dfree(c->d_obl_thr); dfree(c->d_obl_leaves); dfree(c->d_obl_w);
  dfree(c->d_obl_depths); dfree(c->d_ob_fk); dfree(c->d_ob_thr); dfree(c->d_ob_thr_cnt);
  dfree(c->d_obs_trees); dfree(c->d_obs_leaves);
  if (c->d_sb_nodes) (void)hipFree(c->d_sb_nodes);
  if (c->d_sb_bins) (void)hipFree(c->d_sb_bins);
  dfree(c->d_sb_leaves); dfree(c->d_sb_root); dfree(c->d_sb_thr); dfree(c->d_sb_thr_cnt);
  dfree(c->d_p4_batches); dfree(c->d_p4_depth);
  for (auto &p : c->prof_events) {
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  for (auto &p : c->prof_events_child) {
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *qr_last_error(const qr_ctx *c) {
  return c ? c->err.c_str() : g_create_err.c_str();
}

int qr_ctx_set_stream(qr_ctx *c, void *s) {
  if (!c) return QR_ERR_ARG;
  // (a tree with a guessed step count settles on the stream it was enqueued on: its
  // continuation reads state captured at enqueue time -- ADVICE r2)
  { const int src = tree_settle(c); if (src) return src; }
  (void)hipStreamSynchronize(c->stream);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  c->stream = (hipStream_t)s;
  c->own_stream = false;
  return QR_OK;
}

int qr_ctx_stream(qr_ctx *c, void **stream_out) {
  if (!c || !stream_out) return QR_ERR_ARG;
  *stream_out = (void *)c->stream;
  return QR_OK;
}

int qr_ctx_set_shard(qr_ctx *c, int rank, int world) {
  if (!c) return QR_ERR_ARG;
  if (world < 1 || rank < 0 || rank >= world) QR_FAIL(c, QR_ERR_ARG, "bad rank/world");
  if (c->binned) QR_FAIL(c, QR_ERR_STATE, "qr_ctx_set_shard must precede qr_bins_build");
  c->rank = rank;
  c->world = world;
  c->dmode = 0;
  return QR_OK;
}

int qr_ctx_set_doc_shard(qr_ctx *c, int rank, int world, uint64_t n_global,
                         uint64_t q_global) {
  if (!c) return QR_ERR_ARG;
  if (world < 1 || rank < 0 || rank >= world) QR_FAIL(c, QR_ERR_ARG, "bad rank/world");
  if (c->binned) QR_FAIL(c, QR_ERR_STATE, "qr_ctx_set_doc_shard must precede the bin build");
  if (n_global >= (1ull << 32)) QR_FAIL(c, QR_ERR_UNSUPPORTED, "global N must be < 2^32");
  c->rank = rank;
  c->world = world;
  c->dmode = 1;
  c->Nglobal = n_global;
  c->Qglobal = q_global;
  return QR_OK;
}

int qr_synchronize(qr_ctx *c) {
  // (a document-sharded tree that waits for qr_tree_batch_settle: what is enqueued drains, the
  // tree is carried on by the caller, with its all-reduces)
  if (!c->dbatch_pending) { const int src_ = tree_settle(c); if (src_) return src_; }
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  return QR_OK;
}

// bench.py's `roofline.lds_atomic_bound`: the microbenchmark is a library of its own
// (lib/libqr_ubench.so, csrc/k_ubench.hip) next to this one, loaded here on first use -- the
// product library holds no measurement kernel
int qr_prof_lds_atomic(qr_ctx *c, double *cycles_per_instr, double *shader_ghz, double *ns_per_instr,
                       double *root_wave_instr_per_cu) {
  if (!c) return QR_ERR_ARG;
  typedef int (*ubench_fn)(int, int, void *, double *, double *);
  static ubench_fn fn = nullptr;
  if (!fn) {
    Dl_info info;
    std::string path = "libqr_ubench.so";
    if (dladdr((const void *)&qr_prof_lds_atomic, &info) && info.dli_fname) {
      const std::string self = info.dli_fname;
      const size_t slash = self.rfind('/');
      if (slash != std::string::npos) path = self.substr(0, slash + 1) + "libqr_ubench.so";
    }
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (h) fn = (ubench_fn)dlsym(h, "qr_ubench_lds_atomic");
    if (!fn) QR_FAIL(c, QR_ERR_UNSUPPORTED, "libqr_ubench.so (the LDS atomic microbenchmark, quickrank_amd/build.py) is not next to this library");
  }
  double cyc = 0.0, ghz = 0.0;
  if (fn(c->device, c->ncu, (void *)c->stream, &cyc, &ghz) != 0) QR_FAIL(c, QR_ERR_HIP, "the LDS atomic microbenchmark failed");
  if (cycles_per_instr) *cycles_per_instr = cyc;
  if (shader_ghz) *shader_ghz = ghz;
  if (ns_per_instr) *ns_per_instr = ghz > 0.0 ? cyc / ghz : 0.0;
  // what the root launch of the context's data set asks of ONE CU: a wave covers 64 / (fw / 16)
  // documents of a block per sixteen instructions (hist_accumulate, k_tree.hip)
  if (root_wave_instr_per_cu) {
    double w = 0.0;
    const double n = (double)(c->sub_k ? c->sub_n : c->N);
    for (const auto &b : c->blocks) w += n * 16.0 / (double)(64 / (b.fw / 16));
    *root_wave_instr_per_cu = c->binned ? w / (double)c->ncu : 0.0;
  }
  return QR_OK;
}

int qr_debug_check(qr_ctx *c) {
  if (!c) return QR_ERR_ARG;
  return qr_k_debug_check(c);
}

int qr_readback_retries(qr_ctx *c, unsigned long long *count) {
  if (!c || !count) return QR_ERR_ARG;
  *count = c->readback_retries;
  return QR_OK;
}

int qr_tree_pending(qr_ctx *c, int *pending) {
  if (!c || !pending) return QR_ERR_ARG;
  *pending = c->dbatch_pending ? 1 : 0;
  return QR_OK;
}

// ---------------------------------------------------------------------------
static int upload_queries(qr_ctx *c, const uint64_t *qoff, size_t Q, size_t N,
                          uint32_t **d_qoff, size_t *maxq) {
  std::vector<uint32_t> q32(Q + 1);
  size_t mq = 0;
  for (size_t i = 0; i <= Q; ++i) {
    if (qoff[i] > N || (i && qoff[i] < qoff[i - 1]))
      QR_FAIL(c, QR_ERR_ARG, "query offsets must be non-decreasing and <= N");
    q32[i] = (uint32_t)qoff[i];
    if (i) mq = std::max(mq, (size_t)(qoff[i] - qoff[i - 1]));
  }
  if (Q && (qoff[0] != 0 || qoff[Q] != N))
    QR_FAIL(c, QR_ERR_ARG, "query offsets must span [0, N]");
  QR_CHECK(c, dalloc(d_qoff, Q + 1));
  QR_CHECK(c, hipMemcpy(*d_qoff, q32.data(), (Q + 1) * 4, hipMemcpyHostToDevice));
  *maxq = mq;
  return QR_OK;
}

static int ensure_rank_scratch(qr_ctx *c) {
  const size_t nk = std::max(c->N, c->vN), nq = std::max(c->Q, c->vQ);
  if (nk > c->keys_cap) {
    dfree(c->d_keys);
    QR_CHECK(c, dalloc(&c->d_keys, nk));
    c->keys_cap = nk;
  }
  if (nq > c->tied_cap || !c->d_tied) {
    dfree(c->d_tied);
    QR_CHECK(c, dalloc(&c->d_tied, nq + 1));
    c->tied_cap = nq;
  }
  return QR_OK;
}

static int ensure_lg2(qr_ctx *c) {
  int rc0 = ensure_rank_scratch(c);
  if (rc0) return rc0;
  const size_t need = std::max(c->maxq, c->vmaxq) + 2;
  if (need <= c->lg2_len) return QR_OK;
  std::vector<double> t(need);
  // dcg.cc:37 forms (i + 2.0f) in f32 then promotes; ndcg.cc:79 uses
  // (double)(j + 2): identical values while r + 2 < 2^24.
  for (size_t r = 0; r < need; ++r) t[r] = log2((double)((float)r + 2.0f));
  dfree(c->d_lg2);
  dfree(c->d_ilg2);
  QR_CHECK(c, dalloc(&c->d_lg2, need));
  QR_CHECK(c, hipMemcpy(c->d_lg2, t.data(), need * 8, hipMemcpyHostToDevice));
  // ndcg.cc:78-79: 1.0f / log2((double)(j + 2)) -- one IEEE division, tabulated
  for (size_t r = 0; r < need; ++r) t[r] = 1.0 / t[r];
  QR_CHECK(c, dalloc(&c->d_ilg2, need));
  QR_CHECK(c, hipMemcpy(c->d_ilg2, t.data(), need * 8, hipMemcpyHostToDevice));
  c->lg2_len = need;
  return QR_OK;
}

int qr_dataset_upload(qr_ctx *c, const float *x, size_t N, size_t F,
                      const float *labels, const uint64_t *qoff, size_t Q) {
  if (!c) return QR_ERR_ARG;
  if (!x || !labels || !qoff || N == 0 || F == 0)
    QR_FAIL(c, QR_ERR_ARG, "empty dataset");
  if (N >= (1ull << 31)) QR_FAIL(c, QR_ERR_UNSUPPORTED, "N must be < 2^31");
  if ((F + 63) / 64 > QR_MAXBLK * (size_t)c->world)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "too many features");
  QR_CHECK(c, hipSetDevice(c->device));
  free_train(c);
  c->N = N; c->F = F; c->Q = Q;
  QR_CHECK(c, dalloc(&c->d_raw, N * F));
  // (hipMemcpyDefault: `x` may be a host buffer -- the reference's Dataset, dataset.h:65-67 -- or rows
  // that already live on this device, e.g. bench.py's device-generated 32M-document set)
  QR_CHECK(c, hipMemcpy(c->d_raw, x, N * F * 4, hipMemcpyDefault));
  QR_CHECK(c, dalloc(&c->d_labels, N));
  QR_CHECK(c, hipMemcpy(c->d_labels, labels, N * 4, hipMemcpyHostToDevice));
  c->h_labels.assign(labels, labels + N);
  c->h_qoff.assign(qoff, qoff + Q + 1);
  int rc = upload_queries(c, qoff, Q, N, &c->d_qoff, &c->maxq);
  if (rc) return rc;
  QR_CHECK(c, dalloc(&c->d_scores, N));
  QR_CHECK(c, dalloc(&c->d_lambda, N));
  QR_CHECK(c, dalloc(&c->d_weight, N));
  QR_CHECK(c, hipMemset(c->d_scores, 0, N * 8));
  QR_CHECK(c, hipMemset(c->d_lambda, 0, N * 8));
  QR_CHECK(c, hipMemset(c->d_weight, 0, N * 8));
  QR_CHECK(c, dalloc(&c->d_idcg, Q));
  QR_CHECK(c, dalloc(&c->d_qmetric, Q));
  QR_CHECK(c, dalloc(&c->d_ranks, N));
  QR_CHECK(c, dalloc(&c->d_ssq, 2 * std::max(Q, N / QR_SLICE + 2)));
  QR_CHECK(c, dalloc(&c->d_qmax, std::max(Q, N / QR_SLICE + 2)));
  c->idcg_metric = -1;
  return ensure_lg2(c);
}

// rows of width F re-laid to width Fto: missing columns read 0, surplus ones are dropped
static std::vector<float> repad_rows(const float *x, size_t N, size_t F, size_t Fto) {
  std::vector<float> out(N * Fto, 0.0f);
  const size_t w = std::min(F, Fto);
  for (size_t i = 0; i < N; ++i) memcpy(&out[i * Fto], x + i * F, w * sizeof(float));
  return out;
}

int qr_valid_upload(qr_ctx *c, const float *x, size_t N, size_t F, const float *labels,
                    const uint64_t *qoff, size_t Q) {
  if (!c) return QR_ERR_ARG;
  if (!c->F) QR_FAIL(c, QR_ERR_STATE, "upload the training set first");
  if (!x || !labels || !qoff || N == 0 || F == 0) QR_FAIL(c, QR_ERR_ARG, "empty validation set");
  { const int src = tree_settle(c); if (src) return src; }  // (a pending score update walks the old rows)
  free_valid(c);
  c->vN = N; c->vQ = Q;
  QR_CHECK(c, dalloc(&c->d_vraw, N * c->F));
  if (F == c->F) {
    QR_CHECK(c, hipMemcpy(c->d_vraw, x, N * c->F * 4, hipMemcpyHostToDevice));
  } else {  // the validation file's own width: re-laid to the training stride
    const std::vector<float> padded = repad_rows(x, N, F, c->F);
    QR_CHECK(c, hipMemcpy(c->d_vraw, padded.data(), N * c->F * 4, hipMemcpyHostToDevice));
  }
  QR_CHECK(c, dalloc(&c->d_vlabels, N));
  QR_CHECK(c, hipMemcpy(c->d_vlabels, labels, N * 4, hipMemcpyHostToDevice));
  c->h_vlabels.assign(labels, labels + N);
  c->h_vqoff.assign(qoff, qoff + Q + 1);
  int rc = upload_queries(c, qoff, Q, N, &c->d_vqoff, &c->vmaxq);
  if (rc) return rc;
  QR_CHECK(c, dalloc(&c->d_vscores, N));
  QR_CHECK(c, hipMemset(c->d_vscores, 0, N * 8));
  QR_CHECK(c, dalloc(&c->d_vidcg, Q));
  QR_CHECK(c, dalloc(&c->d_vqmetric, Q));
  QR_CHECK(c, dalloc(&c->d_vranks, N));
  c->vidcg_metric = -1;
  return ensure_lg2(c);
}

// ---------------------------------------------------------------------------
static inline uint32_t h_flip(uint32_t x) {
  return x ^ ((uint32_t)(-(int32_t)(x >> 31)) | 0x80000000u);
}
static inline uint32_t h_unflip(uint32_t x) {
  return x ^ (((x >> 31) - 1) | 0x80000000u);
}
static inline float bits2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// distinct values (at most limit + 1 kept) / min / max per column of the rank's
// documents; values are f32 bit patterns, min/max in radix-flipped form
static int bins_stats(qr_ctx *c, uint32_t limit, std::vector<uint32_t> &vals,
                      std::vector<uint32_t> &cnt, std::vector<uint32_t> &mm) {
  const size_t N = c->N, F = c->F;
  float *d_col = nullptr;
  uint32_t *d_vals = nullptr, *d_cnt = nullptr, *d_mm = nullptr;
  QR_CHECK(c, dalloc(&d_col, N * F));
  QR_CHECK(c, dalloc(&d_vals, F * (size_t)(limit + 1)));
  QR_CHECK(c, dalloc(&d_cnt, F));
  QR_CHECK(c, dalloc(&d_mm, 2 * F));
  int rc = qr_k_transpose(c, c->d_raw, d_col, N, F);
  if (rc) return rc;
  rc = qr_k_colstats(c, d_col, N, F, limit, d_vals, d_cnt, d_mm);
  if (rc) return rc;
  vals.resize(F * (size_t)(limit + 1));
  cnt.resize(F);
  mm.resize(2 * F);
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(vals.data(), d_vals, vals.size() * 4, hipMemcpyDeviceToHost));
  QR_CHECK(c, hipMemcpy(cnt.data(), d_cnt, F * 4, hipMemcpyDeviceToHost));
  QR_CHECK(c, hipMemcpy(mm.data(), d_mm, 2 * F * 4, hipMemcpyDeviceToHost));
  dfree(d_col); dfree(d_vals); dfree(d_cnt); dfree(d_mm);
  return QR_OK;
}

// thresholds: mart.cc:147-169 with the same f32 operations, from the column
// statistics of `nranks` document shards (their union is the training set)
static const char *thresholds_from_stats(size_t F, size_t nthresholds, size_t nranks,
                                         const uint32_t *vals, const uint32_t *cnt,
                                         const uint32_t *mm, float *thr, uint32_t *thr_size) {
  const uint32_t limit = (uint32_t)(nthresholds ? nthresholds + 1 : 256);
  const size_t vstride = F * (size_t)(limit + 1);
  for (size_t f = 0; f < F; ++f) {
    float *out = &thr[f * QR_MAX_BINS];
    for (size_t i = 0; i < QR_MAX_BINS; ++i) out[i] = FLT_MAX;
    bool equal_width = false;
    std::vector<uint32_t> keys;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
    for (size_t r = 0; r < nranks; ++r) {
      const uint32_t n = cnt[r * F + f];
      if (n > limit) equal_width = true;
      if (n) {  // a rank without documents contributes nothing
        kmin = std::min(kmin, mm[r * 2 * F + 2 * f]);
        kmax = std::max(kmax, mm[r * 2 * F + 2 * f + 1]);
      }
      if (!equal_width)
        for (uint32_t i = 0; i < n; ++i)
          keys.push_back(h_flip(vals[r * vstride + f * (size_t)(limit + 1) + i]));
    }
    std::vector<float> uniqs;
    if (!equal_width) {
      std::sort(keys.begin(), keys.end());  // radix order (radix.cc:28-30)
      for (uint32_t k : keys) {
        const float v = bits2f(h_unflip(k));
        if (uniqs.empty() || uniqs.back() < v) uniqs.push_back(v);  // mart.cc:149-151
      }
      if (!(uniqs.size() <= nthresholds || nthresholds == 0)) equal_width = true;
    }
    if (!equal_width) {
      if (uniqs.size() + 1 > QR_MAX_BINS)
        return "nthresholds == 0 with a feature of more than 255 distinct values";
      for (size_t i = 0; i < uniqs.size(); ++i) out[i] = uniqs[i];
      out[uniqs.size()] = FLT_MAX;
      thr_size[f] = (uint32_t)uniqs.size() + 1;
    } else {
      if (nthresholds == 0)
        return "nthresholds == 0 with a feature of more than 255 distinct values";
      const float fmin = bits2f(h_unflip(kmin));
      const float fmax = bits2f(h_unflip(kmax));
      float t = fmin;
      const float step = (float)fabs(fmax - t) / nthresholds;  // mart.cc:164-165
      for (size_t j = 0; j != nthresholds; t += step) out[j++] = t;
      out[nthresholds] = FLT_MAX;
      thr_size[f] = (uint32_t)nthresholds + 1;
    }
  }
  return nullptr;
}

// feature blocks, bin map and the tree working set for the thresholds in
// c->h_thr / c->h_thr_size
static int bins_finish(qr_ctx *c) {
  const size_t N = c->N, F = c->F;
  // ---- feature blocks owned by this rank.  Feature-sharded: rank r owns the
  // contiguous range [r*ceil(F/world), (r+1)*ceil(F/world)) (SURVEY.md section
  // 8e); document-sharded and single-GPU contexts own every feature.
  const size_t fworld = c->dmode ? 1 : (size_t)c->world;
  const size_t frank = c->dmode ? 0 : (size_t)c->rank;
  const size_t per_rank = (F + fworld - 1) / fworld;
  const size_t f_lo = std::min(F, per_rank * frank);
  const size_t f_hi = std::min(F, f_lo + per_rank);
  c->root_wg_valid = false;  // (the root launch's ready-made shares follow the blocks: rebuilt at the next root)
  ++c->blocks_gen;
  c->blocks.clear();
  c->h_gf2lf.assign(F, -1);
  c->h_lf2gf.clear();
  size_t off = 0;
  int lf = 0;
  // The owned features are cut into 16-column chunks and the chunks spread as evenly as
  // possible over the fewest blocks of at most 4 chunks: 136 features = 9 chunks
  // = 3 blocks of 48 columns rather than 64 + 64 + 8.  Equal blocks give every
  // workgroup of a histogram launch the same shape of work (a narrow last block
  // was measured 35 % slower per unit of work than the wide ones: it needs several
  // flushes, each re-priming the load pipeline, and held the root launch back by
  // 13 us).
  const size_t chunks = (f_hi - f_lo + 15) / 16;
  const size_t nblk = (chunks + 3) / 4;
  size_t g0 = f_lo;
  for (size_t bi = 0; bi < nblk; ++bi) {
    QrBlock b;
    b.f0 = (int)g0;
    b.fw = 16 * (int)(chunks / nblk + (bi < chunks % nblk ? 1 : 0));
    b.nreal = (int)std::min<size_t>((size_t)b.fw, f_hi - g0);
    g0 += (size_t)b.nreal;
    b.lf0 = lf;
    b.off = off;
    off += N * (size_t)b.fw;
    off = (off + 255) & ~(size_t)255;
    for (int i = 0; i < b.nreal; ++i) {
      c->h_gf2lf[b.f0 + i] = lf + i;
      c->h_lf2gf.push_back(b.f0 + i);
    }
    lf += b.nreal;
    c->blocks.push_back(b);
  }
  c->nblocks = (int)c->blocks.size();
  c->flocal = lf;
  if (c->nblocks == 0) QR_FAIL(c, QR_ERR_ARG, "this rank owns no feature (world > F)");
  if (c->nblocks > QR_MAXBLK) QR_FAIL(c, QR_ERR_UNSUPPORTED, "too many feature blocks");
  c->bins_bytes = off;
  QR_CHECK(c, dalloc(&c->d_thr, F * QR_MAX_BINS));
  QR_CHECK(c, dalloc(&c->d_thr_size, F));
  QR_CHECK(c, hipMemcpy(c->d_thr, c->h_thr.data(), F * QR_MAX_BINS * 4, hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_thr_size, c->h_thr_size.data(), F * 4, hipMemcpyHostToDevice));
  // ---- bin map
  QR_CHECK(c, dalloc(&c->d_bins, c->bins_bytes));
  QR_CHECK(c, dalloc(&c->d_bins_fm, (size_t)c->flocal * N));
  QR_CHECK(c, dalloc(&c->d_blocks, (size_t)c->nblocks));
  QR_CHECK(c, hipMemcpy(c->d_blocks, c->blocks.data(), c->nblocks * sizeof(QrBlock), hipMemcpyHostToDevice));
  QR_CHECK(c, dalloc(&c->d_lf2gf, (size_t)c->flocal));
  QR_CHECK(c, dalloc(&c->d_gf2lf, F));
  QR_CHECK(c, hipMemcpy(c->d_lf2gf, c->h_lf2gf.data(), c->flocal * 4, hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_gf2lf, c->h_gf2lf.data(), F * 4, hipMemcpyHostToDevice));
  int rc = qr_k_binning(c);
  if (rc) return rc;
  // ---- tree working set
  QR_CHECK(c, dalloc(&c->d_order[0], N));
  QR_CHECK(c, dalloc(&c->d_order[1], N));
  QrPlan plan;
  qr_make_plan((uint32_t)N, c->nblocks, c->blocks.data(),
               qr_plan_quantum((unsigned long long)N * qr_plan_wsum(c->nblocks, c->blocks.data()),
                               c->ncu - c->nblocks),
               &plan);
  c->partial_slots = (size_t)c->ncu * plan.kmax;
  QR_CHECK(c, dalloc(&c->d_partials, c->partial_slots * 256 * 64));
  {
    size_t cells = 0;
    for (const auto &b : c->blocks) cells += (size_t)256 * b.fw;
    if (c->dmode) {
      // one int64 buffer carries the reduced sums, the counts (as int64, read
      // back through their low words) and 2 doubles per rank: ONE sum all-reduce
      // per histogram
      c->xh_cells = cells;
      c->xh_len = 2 * cells + 2 * (size_t)c->world;
      QR_CHECK(c, dalloc(&c->d_xh, c->xh_len));
      QR_CHECK(c, hipMemset(c->d_xh, 0, c->xh_len * 8));
      c->d_red_sum = c->d_xh;
      c->d_red_cnt = reinterpret_cast<uint32_t *>(c->d_xh + cells);
      QR_CHECK(c, dalloc(&c->d_red_cnt_loc, cells));  // this rank's counts, kept aside
      QR_CHECK(c, dalloc(&c->d_xscal, 4 * (size_t)c->world));
      QR_CHECK(c, hipMemset(c->d_xscal, 0, 4 * (size_t)c->world * 8));
    } else {
      QR_CHECK(c, dalloc(&c->d_red_sum, cells));
      QR_CHECK(c, dalloc(&c->d_red_cnt, cells));
    }
  }
  QR_CHECK(c, dalloc(&c->d_featrec, 2 * QR_BATCH * (size_t)c->flocal));
  QR_CHECK(c, dalloc(&c->d_featthr, 2 * QR_BATCH * (size_t)c->flocal));
  QR_CHECK(c, dalloc(&c->d_lscan_wg, QR_BATCH * (size_t)c->flocal));
  QR_CHECK(c, hipMemset(c->d_lscan_wg, 0, QR_BATCH * (size_t)c->flocal * sizeof(QrScanWg)));
  QR_CHECK(c, dalloc(&c->d_lpart_ss, 2 * (N / QR_PART_SLICE + QR_BATCH + 2)));
  QR_CHECK(c, dalloc(&c->d_lpart_ss2, 2 * (N / QR_PART_SLICE + QR_BATCH + 2)));
  QR_CHECK(c, dalloc(&c->d_jobsum, 2 * QR_BATCH));
  QR_CHECK(c, dalloc(&c->d_bpart_state, N / QR_PART_SLICE + QR_BATCH + 2));
  QR_CHECK(c, hipMemset(c->d_bpart_state, 0, (N / QR_PART_SLICE + QR_BATCH + 2) * 8));
  c->bepoch = 0;
  QR_CHECK(c, dalloc(&c->d_recs_local, (size_t)2));
  QR_CHECK(c, dalloc(&c->d_recs_all, 2 * (size_t)c->world));
  c->mask_words = (N + 31) / 32;
  // (+ QR_MAXLEVEL words behind the bits: the left counts of a level's nodes, k_obl_mark)
  QR_CHECK(c, dalloc(&c->d_mask, c->mask_words + QR_MAXLEVEL));
  QR_CHECK(c, hipMemset(c->d_mask, 0, (c->mask_words + QR_MAXLEVEL) * 4));
  QR_CHECK(c, dalloc(&c->d_part_state, N / QR_PART_SLICE + 2));
  QR_CHECK(c, hipMemset(c->d_part_state, 0, (N / QR_PART_SLICE + 2) * 8));
  QR_CHECK(c, dalloc(&c->d_part_ss, 2 * (N / QR_PART_SLICE + 2)));
  QR_CHECK(c, dalloc(&c->d_tree, (size_t)1));
  QR_CHECK(c, hipMemset(c->d_tree, 0, sizeof(QrTreeState)));
  QR_CHECK(c, dalloc(&c->d_tree2, (size_t)1));
  QR_CHECK(c, hipMemset(c->d_tree2, 0, sizeof(QrTreeState)));
  QR_CHECK(c, dalloc(&c->d_leafpart, std::max<size_t>(2 * (N / QR_SLICE + QR_MAXNODES + 4), 32 * (N / QR_SLICE + 2))));
  QR_CHECK(c, dalloc(&c->d_leafb, N + 16));
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  c->binned = true;
  return QR_OK;
}

static int bins_precheck(qr_ctx *c, size_t nthresholds) {
  if (!c->d_raw) QR_FAIL(c, QR_ERR_STATE, "no dataset uploaded");
  if (c->binned) QR_FAIL(c, QR_ERR_STATE, "bins already built: upload the dataset again first");
  if (nthresholds > 255)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "the device path stores uint8 bins: nthresholds <= 255");
  QR_CHECK(c, hipSetDevice(c->device));
  return QR_OK;
}

int qr_bins_build(qr_ctx *c, size_t nthresholds, float *thr_out,
                  uint32_t *thr_size_out) {
  if (!c) return QR_ERR_ARG;
  if (c->dmode && c->world > 1)
    QR_FAIL(c, QR_ERR_STATE,
            "document-sharded contexts need the thresholds of the whole training set: "
            "qr_bins_stats -> all_gather -> qr_thresholds_from_stats -> qr_bins_build_with");
  int rc = bins_precheck(c, nthresholds);
  if (rc) return rc;
  const uint32_t limit = (uint32_t)(nthresholds ? nthresholds + 1 : 256);
  std::vector<uint32_t> vals, cnt, mm;
  if ((rc = bins_stats(c, limit, vals, cnt, mm))) return rc;
  c->h_thr.assign(c->F * QR_MAX_BINS, FLT_MAX);
  c->h_thr_size.assign(c->F, 0);
  const char *e = thresholds_from_stats(c->F, nthresholds, 1, vals.data(), cnt.data(), mm.data(),
                                        c->h_thr.data(), c->h_thr_size.data());
  if (e) QR_FAIL(c, QR_ERR_UNSUPPORTED, e);
  if (thr_out) memcpy(thr_out, c->h_thr.data(), c->F * QR_MAX_BINS * 4);
  if (thr_size_out) memcpy(thr_size_out, c->h_thr_size.data(), c->F * 4);
  return bins_finish(c);
}

int qr_bins_stats(qr_ctx *c, size_t nthresholds, uint32_t *vals_out, uint32_t *cnt_out,
                  uint32_t *minmax_out) {
  if (!c || !vals_out || !cnt_out || !minmax_out) return QR_ERR_ARG;
  int rc = bins_precheck(c, nthresholds);
  if (rc) return rc;
  const uint32_t limit = (uint32_t)(nthresholds ? nthresholds + 1 : 256);
  std::vector<uint32_t> vals, cnt, mm;
  if ((rc = bins_stats(c, limit, vals, cnt, mm))) return rc;
  memcpy(vals_out, vals.data(), vals.size() * 4);
  memcpy(cnt_out, cnt.data(), cnt.size() * 4);
  memcpy(minmax_out, mm.data(), mm.size() * 4);
  return QR_OK;
}

int qr_thresholds_from_stats(size_t F, size_t nthresholds, size_t nranks, const uint32_t *vals,
                             const uint32_t *cnt, const uint32_t *minmax, float *thr_out,
                             uint32_t *thr_size_out) {
  if (!vals || !cnt || !minmax || !thr_out || !thr_size_out || nthresholds > 255 || !nranks)
    return QR_ERR_ARG;
  return thresholds_from_stats(F, nthresholds, nranks, vals, cnt, minmax, thr_out, thr_size_out)
             ? QR_ERR_UNSUPPORTED
             : QR_OK;
}

int qr_bins_build_with(qr_ctx *c, const float *thr, const uint32_t *thr_size) {
  if (!c || !thr || !thr_size) return QR_ERR_ARG;
  int rc = bins_precheck(c, 255);
  if (rc) return rc;
  for (size_t f = 0; f < c->F; ++f) {
    const uint32_t n = thr_size[f];
    if (n < 1 || n > QR_MAX_BINS || thr[f * QR_MAX_BINS + n - 1] != FLT_MAX)
      QR_FAIL(c, QR_ERR_ARG, "every threshold row needs 1..256 slots ending in FLT_MAX");
  }
  c->h_thr.assign(thr, thr + c->F * QR_MAX_BINS);
  c->h_thr_size.assign(thr_size, thr_size + c->F);
  return bins_finish(c);
}

// ---- more than 255 thresholds per feature (k_wide.hip) --------------------------
static int bins_build_wide_impl(qr_ctx *c, size_t nthresholds, float *&d_col, size_t *cells_out,
                                size_t *max_slots_out, bool given);
static int bins_build_wide_any(qr_ctx *c, size_t nthresholds, size_t *cells_out, size_t *max_slots_out, bool given);

int qr_bins_build_wide(qr_ctx *c, size_t nthresholds, size_t *cells_out, size_t *max_slots_out) {
  if (!c) return QR_ERR_ARG;
  if (c->dmode)
    QR_FAIL(c, QR_ERR_UNSUPPORTED,
            "more than 255 thresholds per feature on a document-sharded context: the thresholds of the WHOLE "
            "set come first (qr_bins_stats_wide on every rank -> qr_thresholds_from_stats_wide -> "
            "qr_bins_build_wide_with)");
  return bins_build_wide_any(c, nthresholds, cells_out, max_slots_out, false);
}

// ---- thresholds of a document-sharded set with more than 255 of them per feature -----------------
int qr_bins_stats_wide(qr_ctx *c, size_t limit, uint32_t *vals_out, uint32_t *cnt_out, uint32_t *minmax_out) {
  if (!c || !vals_out || !cnt_out || !minmax_out || !limit) return QR_ERR_ARG;
  if (!c->d_raw) QR_FAIL(c, QR_ERR_STATE, "no dataset uploaded");
  QR_CHECK(c, hipSetDevice(c->device));
  float *d_col = nullptr;
  QR_CHECK(c, dalloc(&d_col, c->N * c->F + 1));
  int rc = c->N ? qr_k_transpose(c, c->d_raw, d_col, c->N, c->F) : QR_OK;
  if (!rc) rc = qr_k_wide_stats(c, d_col, limit, vals_out, cnt_out, minmax_out);
  dfree(d_col);
  return rc;
}

// mart.cc:140-169 over the union of `nranks` shards' column statistics (qr_bins_stats_wide with
// limit = nthresholds + 1, or the caller's bound on distinct values when nthresholds == 0): the
// distinct values themselves while there are at most `nthresholds` of them (or nthresholds == 0),
// else `nthresholds` equal steps from the set's minimum, a running f32 sum; every row ends in
// FLT_MAX.  thr_out: ragged rows one after the other (thr_cap floats); pure host code.
int qr_thresholds_from_stats_wide(size_t F, size_t nthresholds, size_t nranks, size_t limit, const uint32_t *vals,
                                  const uint32_t *cnt, const uint32_t *minmax, float *thr_out, size_t thr_cap,
                                  uint32_t *thr_size_out, size_t *cells_out) {
  if (!vals || !cnt || !minmax || !thr_size_out || !nranks || !limit) return QR_ERR_ARG;
  size_t o = 0;
  auto put = [&](float v) {
    if (thr_out && o < thr_cap) thr_out[o] = v;
    ++o;
  };
  for (size_t f = 0; f < F; ++f) {
    bool equal_width = false;
    std::vector<uint32_t> keys;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
    for (size_t r = 0; r < nranks; ++r) {
      const uint32_t n = cnt[r * F + f];
      if (n > limit) equal_width = true;   // (this shard alone has more than `limit` distinct values)
      if (n) {
        kmin = std::min(kmin, minmax[(r * F + f) * 2]);
        kmax = std::max(kmax, minmax[(r * F + f) * 2 + 1]);
      }
      for (uint32_t i = 0; i < std::min<size_t>(n, limit); ++i) keys.push_back(h_flip(vals[(r * F + f) * limit + i]));
    }
    std::vector<float> uniqs;
    if (!equal_width) {
      std::sort(keys.begin(), keys.end());  // radix order (radix.cc:28-30)
      for (uint32_t k : keys) {
        const float v = bits2f(h_unflip(k));
        if (uniqs.empty() || uniqs.back() < v) uniqs.push_back(v);  // mart.cc:149-151
      }
      if (nthresholds && uniqs.size() > nthresholds) equal_width = true;
    }
    const size_t o0 = o;
    if (!equal_width) {
      for (float v : uniqs) put(v);
    } else {
      if (nthresholds == 0) return QR_ERR_UNSUPPORTED;  // more distinct values than the caller gathered
      float t = bits2f(h_unflip(kmin));
      const float step = (float)fabs(bits2f(h_unflip(kmax)) - t) / nthresholds;  // mart.cc:164-165
      for (size_t j = 0; j != nthresholds; t += step, ++j) put(t);
    }
    put(FLT_MAX);
    thr_size_out[f] = (uint32_t)(o - o0);
  }
  if (cells_out) *cells_out = o;
  return thr_out && o > thr_cap ? QR_ERR_ARG : QR_OK;
}

// the wide bins for GIVEN thresholds: ragged rows, feature f's `thr_size[f]` values (ascending, the
// last one FLT_MAX) one after the other -- the document-sharded counterpart of qr_bins_build_with
#define QR_DOC_WIDE_MAX_CELLS ((size_t)4 << 20)   /* cells of an all-reduced node histogram (64 MB of int64) */
int qr_bins_build_wide_with(qr_ctx *c, const float *thr, const uint32_t *thr_size, size_t *cells_out,
                            size_t *max_slots_out) {
  if (!c || !thr || !thr_size) return QR_ERR_ARG;
  if (c->world > 1 && !c->dmode)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "qr_bins_build_wide_with: single-GPU and document-sharded contexts (a "
                                   "feature-sharded rank computes its own features' thresholds: qr_bins_build_wide)");
  size_t cells = 0;
  for (size_t f = 0; f < c->F; ++f) {
    if (thr_size[f] < 1 || thr[cells + thr_size[f] - 1] != FLT_MAX)
      QR_FAIL(c, QR_ERR_ARG, "every threshold row needs at least one slot and ends in FLT_MAX");
    cells += thr_size[f];
  }
  if (cells >= 0xFFFFFFF0ull) QR_FAIL(c, QR_ERR_UNSUPPORTED, "more than 2^32 threshold slots in all");
  if (c->dmode && cells > QR_DOC_WIDE_MAX_CELLS)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "document-sharded contexts all-reduce a node histogram per split: at most 4M "
                                   "threshold slots in all (use --num-thresholds N, or --shard features)");
  c->h_wthr.assign(thr, thr + cells);
  c->h_thr_size.assign(thr_size, thr_size + c->F);
  c->h_woff.assign(c->F + 1, 0);
  for (size_t f = 0; f < c->F; ++f) c->h_woff[f + 1] = c->h_woff[f] + thr_size[f];
  return bins_build_wide_any(c, 0, cells_out, max_slots_out, true);
}

static int bins_build_wide_any(qr_ctx *c, size_t nthresholds, size_t *cells_out, size_t *max_slots_out, bool given) {
  if (!c->d_raw) QR_FAIL(c, QR_ERR_STATE, "no dataset uploaded");
  if (c->binned) QR_FAIL(c, QR_ERR_STATE, "bins already built: upload the dataset again first");
  QR_CHECK(c, hipSetDevice(c->device));
  float *d_col = nullptr;
  const int rc_all = bins_build_wide_impl(c, nthresholds, d_col, cells_out, max_slots_out, given);
  dfree(d_col);  // (whatever happened: the transposed copy is one-time scratch)
  if (rc_all != QR_OK) {
    // nothing half-built stays behind: the wide tables and the tree working set go, the
    // uploaded dataset (raw rows, labels, queries, scores) stays for another attempt
    const std::string keep = c->err;
    (void)hipStreamSynchronize(c->stream);
    dfree(c->d_lf2gf); dfree(c->d_gf2lf); dfree(c->d_thr_size);
    dfree(c->d_woff); dfree(c->d_wthr); dfree(c->d_wbins); dfree(c->d_wbins16);
    dfree(c->d_wchunk); dfree(c->d_wchunk0); dfree(c->d_wtot_s); dfree(c->d_wtot_c); dfree(c->d_wcbest);
    qr_k_exact_free(c);
    dfree(c->d_order[0]); dfree(c->d_order[1]); dfree(c->d_featrec); dfree(c->d_featthr);
    dfree(c->d_recs_local); dfree(c->d_recs_all); dfree(c->d_mask); dfree(c->d_part_state);
    dfree(c->d_part_ss); dfree(c->d_tree); dfree(c->d_leafpart); dfree(c->d_leafb);
    dfree(c->d_tree2); dfree(c->d_lpart_ss); dfree(c->d_lpart_ss2); dfree(c->d_jobsum); dfree(c->d_bpart_state);
    c->wchunks = 0;
    c->wcells = 0;
    c->wmax = 0;
    c->wide = c->binned = false;
    c->err = keep;
  }
  return rc_all;
}

static int bins_build_wide_impl(qr_ctx *c, size_t nthresholds, float *&d_col, size_t *cells_out,
                                size_t *max_slots_out, bool given) {
  const size_t N = c->N, F = c->F;
  {
    // the rank's own features: all of them on one GPU and on a document-sharded rank, the contiguous
    // range [r ceil(F / world), (r + 1) ceil(F / world)) on a feature-sharded context (as the u8 path)
    const size_t fworld = c->dmode ? 1 : (size_t)c->world;
    const size_t per_rank = (F + fworld - 1) / fworld;
    const size_t f0 = std::min(F, per_rank * (c->dmode ? 0 : (size_t)c->rank)), f1 = std::min(F, f0 + per_rank);
    if (f1 <= f0) QR_FAIL(c, QR_ERR_ARG, "this rank owns no feature (world > F)");
    c->blocks.clear();   // no u8 blocks
    c->nblocks = 0;
    c->flocal = (int)(f1 - f0);
    c->h_gf2lf.assign(F, -1);
    c->h_lf2gf.resize(f1 - f0);
    for (size_t f = f0; f < f1; ++f) {
      c->h_gf2lf[f] = (int32_t)(f - f0);
      c->h_lf2gf[f - f0] = (int32_t)f;
    }
    QR_CHECK(c, dalloc(&c->d_lf2gf, (size_t)c->flocal));
    QR_CHECK(c, dalloc(&c->d_gf2lf, F));
    QR_CHECK(c, hipMemcpy(c->d_lf2gf, c->h_lf2gf.data(), c->h_lf2gf.size() * 4, hipMemcpyHostToDevice));
    QR_CHECK(c, hipMemcpy(c->d_gf2lf, c->h_gf2lf.data(), F * 4, hipMemcpyHostToDevice));
  }
  QR_CHECK(c, dalloc(&d_col, N * F));
  int rc = qr_k_transpose(c, c->d_raw, d_col, N, F);
  if (!rc && !given) rc = qr_k_wide_thresholds(c, d_col, nthresholds);  // (given: h_wthr / h_woff / h_thr_size are set)
  if (rc) return rc;
  c->wcells = c->h_wthr.size();
  c->wmax = 0;
  for (size_t f = 0; f < F; ++f) c->wmax = std::max(c->wmax, c->h_thr_size[f]);
  const size_t FL = c->h_lf2gf.size();  // the rank's own features (all of them on one GPU)
  QR_CHECK(c, dalloc(&c->d_thr_size, F));
  QR_CHECK(c, hipMemcpy(c->d_thr_size, c->h_thr_size.data(), F * 4, hipMemcpyHostToDevice));
  QR_CHECK(c, dalloc(&c->d_woff, FL + 1));
  QR_CHECK(c, hipMemcpy(c->d_woff, c->h_woff.data(), (FL + 1) * 4, hipMemcpyHostToDevice));
  QR_CHECK(c, dalloc(&c->d_wthr, c->wcells));
  QR_CHECK(c, hipMemcpy(c->d_wthr, c->h_wthr.data(), c->wcells * 4, hipMemcpyHostToDevice));
  QR_CHECK(c, dalloc(&c->d_wbins, N * FL));
  if (qr_k_wide_fast_rows(c, c->wmax))  // short rows: the blocked u16 copy the fast histogram kernel reads
    QR_CHECK(c, dalloc(&c->d_wbins16, N * 16 * ((FL + 15) / 16)));
  {  // the chunk table of the chunked scan: (feature, first slot) of every QR_WCHUNK slots of a row
    std::vector<uint32_t> ch, first(FL + 1, 0);
    for (size_t f = 0; f < FL; ++f) {
      first[f] = (uint32_t)(ch.size() / 2);
      const uint32_t size = c->h_woff[f + 1] - c->h_woff[f];
      for (uint32_t t0 = 0; t0 < size; t0 += QR_WCHUNK) {
        ch.push_back((uint32_t)f);
        ch.push_back(t0);
      }
    }
    first[FL] = (uint32_t)(ch.size() / 2);
    c->wchunks = ch.size() / 2;
    QR_CHECK(c, dalloc(&c->d_wchunk, ch.size()));
    QR_CHECK(c, hipMemcpy(c->d_wchunk, ch.data(), ch.size() * 4, hipMemcpyHostToDevice));
    QR_CHECK(c, dalloc(&c->d_wchunk0, first.size()));
    QR_CHECK(c, hipMemcpy(c->d_wchunk0, first.data(), first.size() * 4, hipMemcpyHostToDevice));
    QR_CHECK(c, dalloc(&c->d_wtot_s, c->wchunks));
    QR_CHECK(c, dalloc(&c->d_wtot_c, c->wchunks));
    QR_CHECK(c, dalloc(&c->d_wcbest, 4 * c->wchunks));  // two 16-byte records per chunk
  }
  rc = qr_k_wide_binning(c, d_col);
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  if (rc) return rc;
  // rows of more than QR_X_MIN_SLOTS slots (the reference's default --num-thresholds 0 on real-valued
  // columns): slot-indexed node histograms cost 0.8 GB of cells per node whatever its size; the
  // pre-sorted lists of k_exact.hip cost 8 bytes per (document, feature) of the node.  Single-GPU
  // and (round 6) FEATURE-sharded contexts -- a rank holds every document and the lists of its own
  // features; the records it hands the all-gather and the mask it takes back are the slot path's;
  // 3 x F_local x N x 8 bytes of lists (QR_WIDE_EXACT=1 / QR_WIDE_NO_EXACT=1 force either way).
  if (!c->dmode && !getenv("QR_WIDE_NO_EXACT") &&
      (c->wmax > QR_X_MIN_SLOTS || getenv("QR_WIDE_EXACT")) && 3 * FL * N * 8 <= ((size_t)64 << 30)) {
    // (decided by the memory that is really free, and a build that still fails --
    // another context took it in between -- leaves the slot-indexed path of round 3 in place
    // instead of failing the bin build)
    size_t mfree = 0, mtotal = 0;
    const size_t need = 3 * FL * N * 8 + N * 16 + ((size_t)64 << 20);  // lists + sort scratch + words
    if (hipMemGetInfo(&mfree, &mtotal) != hipSuccess || mfree >= need + need / 16) {
      if ((rc = qr_k_exact_build(c))) {
        (void)hipGetLastError();  // (an allocation failure is not sticky)
        qr_k_exact_free(c);
        if (c->spec_debug) fprintf(stderr, "qr: pre-sorted lists not built (%s): slot-indexed histograms\n", c->err.c_str());
        rc = QR_OK;
      }
    } else if (c->spec_debug)
      fprintf(stderr, "qr: %zu MB free, the pre-sorted lists need %zu MB: slot-indexed histograms\n", mfree >> 20, need >> 20);
  }
  // ---- tree working set of the one-split-per-step path
  QR_CHECK(c, dalloc(&c->d_order[0], N));
  QR_CHECK(c, dalloc(&c->d_order[1], N));
  QR_CHECK(c, dalloc(&c->d_featrec, 2 * QR_BATCH * F));
  QR_CHECK(c, dalloc(&c->d_featthr, 2 * QR_BATCH * F));
  QR_CHECK(c, dalloc(&c->d_recs_local, (size_t)2));
  QR_CHECK(c, dalloc(&c->d_recs_all, 2 * (size_t)c->world));
  c->mask_words = (N + 31) / 32;
  // (+ 1 word on feature-sharded contexts: the winning slot's threshold VALUE rides behind the
  // go-left bits -- only its owner knows it, every rank's tree records need it)
  QR_CHECK(c, dalloc(&c->d_mask, c->mask_words + 2));
  QR_CHECK(c, hipMemset(c->d_mask, 0, (c->mask_words + 2) * 4));
  QR_CHECK(c, dalloc(&c->d_part_state, N / QR_PART_SLICE + 2));
  QR_CHECK(c, hipMemset(c->d_part_state, 0, (N / QR_PART_SLICE + 2) * 8));
  QR_CHECK(c, dalloc(&c->d_part_ss, 2 * (N / QR_PART_SLICE + 2)));
  QR_CHECK(c, dalloc(&c->d_tree, (size_t)1));
  QR_CHECK(c, hipMemset(c->d_tree, 0, sizeof(QrTreeState)));
  if (c->world == 1) {  // batched growth (two splits per step): second copy of the tree state, job sums
    QR_CHECK(c, dalloc(&c->d_tree2, (size_t)1));
    QR_CHECK(c, hipMemset(c->d_tree2, 0, sizeof(QrTreeState)));
    QR_CHECK(c, dalloc(&c->d_lpart_ss, 2 * (N / QR_PART_SLICE + QR_BATCH + 2)));
    QR_CHECK(c, dalloc(&c->d_lpart_ss2, 2 * (N / QR_PART_SLICE + QR_BATCH + 2)));
    QR_CHECK(c, dalloc(&c->d_jobsum, 2 * QR_BATCH));
    QR_CHECK(c, dalloc(&c->d_bpart_state, N / QR_PART_SLICE + QR_BATCH + 2));
    QR_CHECK(c, hipMemset(c->d_bpart_state, 0, (N / QR_PART_SLICE + QR_BATCH + 2) * 8));
  }
  QR_CHECK(c, dalloc(&c->d_leafpart, std::max<size_t>(2 * (N / QR_SLICE + QR_MAXNODES + 4), 32 * (N / QR_SLICE + 2))));
  QR_CHECK(c, dalloc(&c->d_leafb, N + 16));
  if (c->dmode) {
    // the histogram exchange buffer of the document-sharded protocol: [cells] sums, [cells] counts
    // (int64), 2 doubles per rank -- ONE sum all-reduce per node histogram, as on the u8 path
    dfree(c->d_xh); dfree(c->d_xscal);
    c->xh_cells = c->wcells;
    c->xh_len = 2 * c->wcells + 2 * (size_t)c->world;
    QR_CHECK(c, dalloc(&c->d_xh, c->xh_len));
    QR_CHECK(c, hipMemset(c->d_xh, 0, c->xh_len * 8));
    QR_CHECK(c, dalloc(&c->d_xscal, 4 * (size_t)c->world));
    QR_CHECK(c, hipMemset(c->d_xscal, 0, 4 * (size_t)c->world * 8));
  }
  c->wide = true;
  c->binned = true;
  if (cells_out) *cells_out = c->wcells;
  if (max_slots_out) *max_slots_out = c->wmax;
  return QR_OK;
}

int qr_thresholds_read(qr_ctx *c, float *thr_out, uint32_t *thr_size_out) {
  if (!c) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  if (thr_size_out) memcpy(thr_size_out, c->h_thr_size.data(), c->F * 4);
  if (!thr_out) return QR_OK;
  if (c->wide) {
    memcpy(thr_out, c->h_wthr.data(), c->wcells * 4);
  } else {
    size_t o = 0;
    for (size_t f = 0; f < c->F; ++f)
      for (uint32_t t = 0; t < c->h_thr_size[f]; ++t) thr_out[o++] = c->h_thr[f * QR_MAX_BINS + t];
  }
  return QR_OK;
}

int qr_bins_read_u32(qr_ctx *c, uint32_t *out) {
  if (!c || !out) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  if (c->wide) {
    std::vector<uint32_t> h(c->N * (size_t)c->flocal);
    QR_CHECK(c, hipMemcpy(h.data(), c->d_wbins, h.size() * 4, hipMemcpyDeviceToHost));
    for (size_t f = 0; f < c->F; ++f) {
      const int lf = c->h_gf2lf[f];   // (-1: another rank's feature)
      for (size_t d = 0; d < c->N; ++d) out[d * c->F + f] = lf < 0 ? 0xFFFFFFFFu : h[(size_t)lf * c->N + d];
    }
    return QR_OK;
  }
  std::vector<uint8_t> b(c->N * c->F);
  const int rc = qr_bins_read(c, b.data());
  if (rc) return rc;
  for (size_t i = 0; i < b.size(); ++i) out[i] = b[i] == 0xFF && c->h_gf2lf[i % c->F] < 0 ? 0xFFFFFFFFu : b[i];
  return QR_OK;
}

int qr_bins_read(qr_ctx *c, uint8_t *out) {
  if (!c || !out) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  if (c->wide) QR_FAIL(c, QR_ERR_STATE, "this context has more than 255 thresholds per feature: qr_bins_read_u32");
  std::vector<uint8_t> h(c->bins_bytes);
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(h.data(), c->d_bins, c->bins_bytes, hipMemcpyDeviceToHost));
  memset(out, 0xFF, c->N * c->F);
  for (const QrBlock &b : c->blocks)
    for (size_t d = 0; d < c->N; ++d)
      for (int i = 0; i < b.nreal; ++i)
        out[d * c->F + b.f0 + i] = h[b.off + d * (size_t)b.fw + i];
  return QR_OK;
}

int qr_bins_read_fm(qr_ctx *c, uint8_t *out) {
  if (!c || !out) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  if (c->wide) QR_FAIL(c, QR_ERR_STATE, "this context has more than 255 thresholds per feature: qr_bins_read_u32");
  std::vector<uint8_t> h((size_t)c->flocal * c->N);
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(h.data(), c->d_bins_fm, h.size(), hipMemcpyDeviceToHost));
  memset(out, 0xFF, c->N * c->F);
  for (const QrBlock &b : c->blocks)
    for (int i = 0; i < b.nreal; ++i)
      for (size_t d = 0; d < c->N; ++d) out[d * c->F + b.f0 + i] = h[(size_t)(b.lf0 + i) * c->N + d];
  return QR_OK;
}

int qr_bins_verify(qr_ctx *c, unsigned long long *bad_rows, unsigned long long *bad_fm) {
  if (!c || !bad_rows || !bad_fm) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  if (c->wide) QR_FAIL(c, QR_ERR_STATE, "this context has more than 255 thresholds per feature");
  { const int src_ = tree_settle(c); if (src_) return src_; }
  return qr_k_bins_verify(c, bad_rows, bad_fm);
}

int qr_debug_bins_clobber(qr_ctx *c, int which, size_t first_doc, size_t ndocs) {
  if (!c || which < 0 || which > 3) return QR_ERR_ARG;
  if (which == 3) {  // the next tree's records are read with a root `first_doc` documents short
    c->debug_root_short = first_doc;
    return QR_OK;
  }
  if (which == 2) {  // the next `first_doc` builds of the map lose documents [8, 16) behind their kernels
    c->debug_lose_binning = (int)first_doc;
    return QR_OK;
  }
  if (!c->binned || c->wide) QR_FAIL(c, QR_ERR_STATE, "u8 bins not built");
  if (first_doc > c->N || ndocs > c->N - first_doc) QR_FAIL(c, QR_ERR_ARG, "documents out of range");
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  for (const QrBlock &b : c->blocks) {
    if (which == 0) {
      QR_CHECK(c, hipMemset(c->d_bins + b.off + first_doc * (size_t)b.fw, 0, ndocs * (size_t)b.fw));
    } else {
      for (int i = 0; i < b.nreal; ++i)
        QR_CHECK(c, hipMemset(c->d_bins_fm + (size_t)(b.lf0 + i) * c->N + first_doc, 0, ndocs));
    }
  }
  return QR_OK;
}

int qr_debug_sample_key_mask(qr_ctx *c, uint32_t mask) {
  if (!c) return QR_ERR_ARG;
  c->sub_key_mask = mask;
  return QR_OK;
}

// ---------------------------------------------------------------------------
int qr_scores_reset(qr_ctx *c) {
  if (!c || !c->d_scores) return QR_ERR_STATE;
  { const int src_ = tree_settle(c); if (src_) return src_; }
  QR_CHECK(c, hipMemsetAsync(c->d_scores, 0, c->N * 8, c->stream));
  if (c->d_vscores) QR_CHECK(c, hipMemsetAsync(c->d_vscores, 0, c->vN * 8, c->stream));
  return QR_OK;
}
int qr_scores_set(qr_ctx *c, const double *s) {
  if (!c || !c->d_scores || !s) return QR_ERR_ARG;
  { const int src_ = tree_settle(c); if (src_) return src_; }
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(c->d_scores, s, c->N * 8, hipMemcpyHostToDevice));
  return QR_OK;
}
int qr_valid_scores_set(qr_ctx *c, const double *s) {
  if (!c || !s) return QR_ERR_ARG;
  if (!c->d_vscores) QR_FAIL(c, QR_ERR_STATE, "no validation set uploaded");
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(c->d_vscores, s, c->vN * 8, hipMemcpyHostToDevice));
  return QR_OK;
}
int qr_scores_get(qr_ctx *c, double *s) {
  if (!c || !c->d_scores || !s) return QR_ERR_ARG;
  { const int src_ = tree_settle(c); if (src_) return src_; }
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(s, c->d_scores, c->N * 8, hipMemcpyDeviceToHost));
  return QR_OK;
}
int qr_valid_scores_get(qr_ctx *c, double *s) {
  if (!c || !c->d_vscores || !s) return QR_ERR_ARG;
  { const int src_ = tree_settle(c); if (src_) return src_; }
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(s, c->d_vscores, c->vN * 8, hipMemcpyDeviceToHost));
  return QR_OK;
}
int qr_pseudo_get(qr_ctx *c, double *l, double *w) {
  if (!c || !c->d_lambda) return QR_ERR_ARG;
  { const int src_ = tree_settle(c); if (src_) return src_; }
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  if (l) QR_CHECK(c, hipMemcpy(l, c->d_lambda, c->N * 8, hipMemcpyDeviceToHost));
  if (w) QR_CHECK(c, hipMemcpy(w, c->d_weight, c->N * 8, hipMemcpyDeviceToHost));
  return QR_OK;
}

// host side of the scale derivation for caller-supplied pseudo-responses
int qr_pseudo_set(qr_ctx *c, const double *l, const double *w) {
  if (!c || !c->d_lambda || !l) return QR_ERR_ARG;
  { const int src_ = tree_settle(c); if (src_) return src_; }
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(c->d_lambda, l, c->N * 8, hipMemcpyHostToDevice));
  if (w) QR_CHECK(c, hipMemcpy(c->d_weight, w, c->N * 8, hipMemcpyHostToDevice));
  // maxabs + per-slice sum of squares with the same kernels' conventions:
  // reuse k_residual's reduction by computing on the host what it would write
  double mx = 0.0;
  const size_t ns = (c->N + QR_SLICE - 1) / QR_SLICE;
  std::vector<double> ssq(2 * ns, 0.0);
  for (size_t i = 0; i < c->N; ++i) {
    mx = std::max(mx, std::fabs(l[i]));
    ssq[2 * (i / QR_SLICE)] += l[i] * l[i];
    ssq[2 * (i / QR_SLICE) + 1] += l[i];
  }
  QrScalars s;
  QR_CHECK(c, hipMemcpy(&s, c->d_scalars, sizeof(s), hipMemcpyDeviceToHost));
  memcpy(&s.maxabs_bits, &mx, 8);
  QR_CHECK(c, hipMemcpy(c->d_scalars, &s, sizeof(s), hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_ssq, ssq.data(), 2 * ns * 8, hipMemcpyHostToDevice));
  c->nqmax = 0;  // the maximum is in the scalars already
  int rc = qr_k_prep(c, ns, 0);
  if (rc || !c->dmode) return rc;
  return qr_k_prep_pack(c);  // then: all-reduce the scalar buffer, qr_lambda_finish
}

// ---------------------------------------------------------------------------
// Ndcg::compute_idcg (ndcg.cc:35-47) per query: labels sorted with
// std::greater<int> (the same libstdc++ std::sort the reference calls), then
// Dcg::compute_dcg (dcg.cc:33-39).  Labels never change, so this runs once per
// (metric, cutoff) on the host and is uploaded.
static void host_idcg(const std::vector<float> &labels, const std::vector<uint64_t> &qoff,
                      size_t cutoff, std::vector<double> &out) {
  const size_t Q = qoff.size() - 1;
  out.assign(Q, 0.0);
  const size_t k = cutoff == 0 ? SIZE_MAX : cutoff;
  std::vector<float> copy;
  for (size_t q = 0; q < Q; ++q) {
    const size_t n = qoff[q + 1] - qoff[q];
    copy.assign(labels.begin() + qoff[q], labels.begin() + qoff[q + 1]);
    std::sort(copy.begin(), copy.end(), std::greater<int>());
    const size_t size = std::min(k, n);
    double dcg = 0.0;
    for (size_t i = 0; i < size; ++i)
      dcg += (pow(2.0, (double)copy[i]) - 1.0f) / log2((double)((float)i + 2.0f));
    out[q] = dcg;
  }
}

static int ensure_idcg(qr_ctx *c, int which, int metric, size_t cutoff) {
  if (metric != QR_METRIC_NDCG) return QR_OK;
  int &m = which ? c->vidcg_metric : c->idcg_metric;
  size_t &k = which ? c->vidcg_cutoff : c->idcg_cutoff;
  if (m == metric && k == cutoff) return QR_OK;
  std::vector<double> v;
  host_idcg(which ? c->h_vlabels : c->h_labels, which ? c->h_vqoff : c->h_qoff, cutoff, v);
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(which ? c->d_vidcg : c->d_idcg, v.data(), v.size() * 8, hipMemcpyHostToDevice));
  m = metric;
  k = cutoff;
  return QR_OK;
}

// snapshot of the per-iteration scalars into pinned host memory, stream-ordered:
// qr_metric_last waits for the publishing kernel only, not for the work enqueued after it
static int snapshot_scalars(qr_ctx *c) {
  // the kernel that finished the scalars has written them into the pinned block
  // itself (no copy launch), then its sequence number (c->scal_seq)
  c->scal_pending = true;
  return QR_OK;
}

int qr_lambda_compute(qr_ctx *c, int metric, size_t cutoff) {
  if (!c) return QR_ERR_ARG;
  if (!c->d_scores) QR_FAIL(c, QR_ERR_STATE, "no dataset uploaded");
  // (a score update left pending rides in this pass: k_lambda.hip)
  { const int src_ = tree_settle(c, true); if (src_) return src_; }
  if (metric != QR_METRIC_NDCG && metric != QR_METRIC_DCG)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "metric must be DCG or NDCG");
  int rc = ensure_idcg(c, 0, metric, cutoff);
  if (rc) return rc;
  if (c->sub_k && (rc = qr_k_sample_draw(c))) return rc;  // this iteration's sample
  rc = qr_k_lambda(c, 0, metric, cutoff, 0);
  if (rc) return rc;
  // sum of squares / sum / quantisation scale and the metric of the ranking, one launch
  // (documents outside a sample have lambda == 0: the sums are the sample's)
  if (!c->dmode && !c->no_defer) {
    // the launch that finishes them is deferred: the tree's root scan launch carries its
    // workgroups (k_tree.hip: launch_hist_scan), anything else finishes them first
    c->prep_deferred = true;
    c->prep_nss = c->Q;
    c->prep_with_metric = 1;
    c->prep_publish = 1;
    return snapshot_scalars(c);
  }
  if ((rc = qr_k_prep(c, c->Q, 1, c->dmode ? 0 : 1))) return rc;
  if (!c->dmode) return snapshot_scalars(c);
  return qr_k_prep_pack(c);  // then: all-reduce the scalar buffer, qr_lambda_finish
}

int qr_lambda_finish(qr_ctx *c) {
  if (!c) return QR_ERR_ARG;
  if (!c->dmode) QR_FAIL(c, QR_ERR_STATE, "qr_lambda_finish is for document-sharded contexts");
  if (!c->d_xscal) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  int rc = qr_k_prep_global(c);
  if (rc) return rc;
  return snapshot_scalars(c);
}

int qr_residual_compute(qr_ctx *c) {
  if (!c) return QR_ERR_ARG;
  if (!c->d_scores) QR_FAIL(c, QR_ERR_STATE, "no dataset uploaded");
  { const int src_ = tree_settle(c); if (src_) return src_; }
  int rc = 0;
  if (c->sub_k && (rc = qr_k_sample_draw(c))) return rc;  // this iteration's sample
  if ((rc = qr_k_residual(c))) return rc;  // every document (mart.cc:418-431)
  size_t nss = (c->N + QR_SLICE - 1) / QR_SLICE;
  if (c->sub_k) {  // the root's statistics run over the sample (rtnode_histogram.cc:199-203)
    if ((rc = qr_k_sample_sums(c))) return rc;
    nss = std::max<size_t>(1, (c->sub_n + QR_SLICE - 1) / QR_SLICE);
  }
  if ((rc = qr_k_prep(c, nss, 0)) || !c->dmode) return rc;
  return qr_k_prep_pack(c);
}

static int metric_finish(qr_ctx *c, int which, double *out) {
  int rc = qr_k_metric_reduce(c, which);
  if (rc) return rc;
  QrScalars s;
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(&s, c->d_scalars, sizeof(s), hipMemcpyDeviceToHost));
  const size_t Q = which ? c->vQ : c->Q;
  // metric.h:93-105: avg_score /= num_queries (0 queries -> 0.0)
  *out = Q ? s.metric_sum / (double)Q : 0.0;
  return QR_OK;
}

int qr_metric_eval(qr_ctx *c, int which, int metric, size_t cutoff, double *out) {
  if (!c || !out) return QR_ERR_ARG;
  if (which ? !c->d_vscores : !c->d_scores) QR_FAIL(c, QR_ERR_STATE, "dataset not uploaded");
  { const int src_ = tree_settle(c); if (src_) return src_; }
  int rc = ensure_idcg(c, which, metric, cutoff);
  if (rc) return rc;
  rc = qr_k_lambda(c, which, metric, cutoff, 1);
  if (rc) return rc;
  return metric_finish(c, which, out);
}

int qr_metric_last(qr_ctx *c, double *out) {
  if (!c || !out) return QR_ERR_ARG;
  if (!c->scal_pending)
    QR_FAIL(c, QR_ERR_STATE, "qr_metric_last follows qr_lambda_compute (+ qr_lambda_finish)");
  // (scalars nobody has finished yet: no tree followed the lambda pass)
  { const int frc = qr_k_prep_flush(c); if (frc) return frc; }
  // waits for the lambda pass only; whatever was enqueued after it keeps running
  { const int wrc = wait_seq32(c, &c->h_pin->scal.pad, c->scal_seq, "the iteration's scalars"); if (wrc) return wrc; }
  // (self-validating, like the tree records: the two sums must fit the sequence number just seen)
  QrScalars s;
  for (long spin = 0;; ++spin) {
    const volatile QrScalars *v = &c->h_pin->scal;
    s.metric_sum = v->metric_sum;
    s.metric_gsum = v->metric_gsum;
    const unsigned long long tag = v->tag;
    unsigned long long a, b;
    memcpy(&a, &s.metric_sum, 8);
    memcpy(&b, &s.metric_gsum, 8);
    if (tag == qr_scal_tag(a, b, c->scal_seq)) break;
    ++c->readback_retries;
    if (spin > 20000000L) QR_FAIL(c, QR_ERR_STATE, "the iteration's scalars do not fit their sequence number (pinned read-back)");
    cpu_relax(spin);
  }
  // metric.h:93-105: avg_score /= num_queries (0 queries -> 0.0)
  if (c->dmode)  // the sum over all ranks came with the scalar exchange
    *out = c->Qglobal ? s.metric_gsum / (double)c->Qglobal : 0.0;
  else
    *out = c->Q ? s.metric_sum / (double)c->Q : 0.0;
  return QR_OK;
}

// ---------------------------------------------------------------------------
static int ensure_hist_slots(qr_ctx *c, size_t slots) {
  if (slots <= c->hist_slots) return QR_OK;
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  dfree(c->d_hsum);
  dfree(c->d_hcnt);
  const size_t per_slot = c->wide ? c->wcells : (size_t)c->flocal * 256;
  QR_CHECK(c, dalloc(&c->d_hsum, slots * per_slot));
  QR_CHECK(c, dalloc(&c->d_hcnt, slots * per_slot));
  if (c->dmode) {
    dfree(c->d_hcnt_loc);
    QR_CHECK(c, dalloc(&c->d_hcnt_loc, slots * per_slot));
  }
  c->hist_slots = slots;
  return QR_OK;
}

// scratch of the level-wise (oblivious) launches, sized for the widest level
static int ensure_level_buffers(qr_ctx *c, size_t depth) {
  const size_t nodes = (size_t)1 << (depth - 1);
  if (nodes > QR_MAXLEVEL) QR_FAIL(c, QR_ERR_UNSUPPORTED, "tree depth must be in [1, 9]");
  size_t wsum = 0;
  for (const auto &b : c->blocks) wsum += b.fw / 16;
  const size_t G = (size_t)c->ncu;
  const size_t hist_wgs = G + nodes * (size_t)(c->nblocks + 1);
  const size_t part_wgs = c->N / QR_PART_SLICE + nodes + 2;
  // k_obl_plan's quantum is <= (N/2) * wsum / (G/4) + 1 units, so a workgroup gets at
  // most 2 * wsum * N / G + 1 + QR_SLICE documents, hence flushes per workgroup:
  // (a document-sharded rank's own part of a level's "smaller" children can be all of its
  // documents -- which child is built is decided by the GLOBAL counts -- hence N, not N / 2)
  const size_t kmax = ((c->dmode ? 4 : 2) * wsum * c->N / G + 2 * QR_SLICE + QR_DPW - 1) / QR_DPW;
  const size_t slots = hist_wgs * kmax;
  if (hist_wgs > c->lhist_cap || part_wgs > c->lpart_cap || slots > c->lslots_cap ||
      nodes > c->lred_nodes) {
    QR_CHECK(c, hipStreamSynchronize(c->stream));
    dfree(c->d_lhist_map); dfree(c->d_lpart_map); dfree(c->d_lpartials); dfree(c->d_lhistsum);
    dfree(c->d_lpart_state);
    dfree(c->d_lhist_wg); dfree(c->d_lpart_wg); dfree(c->d_lplan);
    QR_CHECK(c, dalloc(&c->d_lhist_wg, hist_wgs));
    QR_CHECK(c, dalloc(&c->d_lpart_wg, part_wgs));
    QR_CHECK(c, dalloc(&c->d_lplan, nodes));
    QR_CHECK(c, dalloc(&c->d_lhist_map, hist_wgs));
    QR_CHECK(c, dalloc(&c->d_lpart_map, part_wgs));
    QR_CHECK(c, dalloc(&c->d_lpart_state, part_wgs));
    QR_CHECK(c, hipMemset(c->d_lpart_state, 0, part_wgs * 8));
    QR_CHECK(c, dalloc(&c->d_lpartials, slots * 256 * 64));
    QR_CHECK(c, dalloc(&c->d_lhistsum, 2 * slots));
    c->lhist_cap = hist_wgs;
    c->lpart_cap = part_wgs;
    c->lslots_cap = slots;
    c->lred_nodes = nodes;
  }
  return QR_OK;
}

// feature-sharded ranks hold every document: each draws the same sample (a pure function of
// seed and iteration).  Document-sharded ranks (qr_subsample_set_doc) draw it from the keys of
// ALL ranks' documents -- a function of the global document index, so every rank finds the same
// sample without an exchange -- and keep their own part.
static int subsample_set(qr_ctx *c, float subsample, uint64_t seed, size_t first_doc) {
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  { const int src = tree_settle(c); if (src) return src; }  // (the pending tree's leaf kernels read the old sample)
  if (!(subsample > 0.0f)) QR_FAIL(c, QR_ERR_ARG, "subsample must be > 0");
  const size_t Nall = c->dmode ? (size_t)c->Nglobal : c->N;
  if (c->dmode && first_doc + c->N > Nall) QR_FAIL(c, QR_ERR_ARG, "the rank's documents lie outside the global range");
  // mart.cc:289-297: > 1 is a number of documents, < 1 a fraction (rounded down)
  size_t k = subsample > 1.0f ? std::min((size_t)subsample, Nall)
                              : (size_t)std::floor(subsample * (float)Nall);
  if (subsample == 1.0f || k >= Nall) k = 0;  // the whole set: no sampling
  if (subsample != 1.0f && k == 0 && (size_t)std::floor(subsample * (float)Nall) == 0 && subsample < 1.0f)
    QR_FAIL(c, QR_ERR_ARG, "subsample leaves no document");
  c->sub_k = k;
  c->sub_n = k;
  c->sub_first = first_doc;
  c->sub_seed = seed * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull;
  c->sub_iter = 0;
  if (k && !c->d_present) {
    QR_CHECK(c, dalloc(&c->d_present, c->N));
    QR_CHECK(c, dalloc(&c->d_sample_count, (size_t)1));
    QR_CHECK(c, hipMalloc(&c->d_sample_work, qr_k_sample_work_bytes(c->N)));
  }
  return QR_OK;
}

int qr_subsample_set(qr_ctx *c, float subsample, uint64_t seed) {
  if (!c) return QR_ERR_ARG;
  if (c->dmode)
    QR_FAIL(c, QR_ERR_STATE, "document-sharded contexts: qr_subsample_set_doc (it needs the rank's first document)");
  return subsample_set(c, subsample, seed, 0);
}

int qr_subsample_set_doc(qr_ctx *c, float subsample, uint64_t seed, size_t first_doc) {
  if (!c) return QR_ERR_ARG;
  if (!c->dmode) QR_FAIL(c, QR_ERR_STATE, "qr_subsample_set_doc is for document-sharded contexts");
  return subsample_set(c, subsample, seed, first_doc);
}

int qr_tree_set_max_features(qr_ctx *c, float max_features, uint64_t seed) {
  if (!c) return QR_ERR_ARG;
  if (!c->F) QR_FAIL(c, QR_ERR_STATE, "no dataset uploaded");
  if (!(max_features > 0.0f)) QR_FAIL(c, QR_ERR_ARG, "max_features must be > 0");
  { const int src = tree_settle(c); if (src) return src; }
  // rt.cc:227-233: > 1 is a number of features, < 1 a fraction (rounded up)
  size_t k = max_features > 1.0f ? (size_t)max_features : (size_t)std::ceil(max_features * (float)c->F);
  if (max_features == 1.0f || k >= c->F) k = 0;  // every feature: no sampling
  c->mf_k = (uint32_t)k;
  c->mf_seed = seed * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull;
  c->tree_counter = 0;
  return QR_OK;
}

int qr_tree_begin(qr_ctx *c, size_t nleaves, uint64_t minls) {
  if (!c) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  if (nleaves < 1 || 2 * nleaves + 1 > QR_MAXNODES)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "nleaves must be in [1, 511]");
  int rc = tree_settle(c);
  if (rc) return rc;
  // (pre-sorted wide contexts build no node histograms: k_exact.hip)
  rc = qr_exact_active(c) ? QR_OK : ensure_hist_slots(c, 2 * nleaves + 1);
  if (rc) return rc;
  if (c->dmode && 2 * nleaves * (size_t)c->world > c->xleaf_cap) {
    QR_CHECK(c, hipStreamSynchronize(c->stream));
    dfree(c->d_xleaf);
    c->xleaf_cap = 2 * nleaves * (size_t)c->world;
    QR_CHECK(c, dalloc(&c->d_xleaf, c->xleaf_cap));
    QR_CHECK(c, hipMemset(c->d_xleaf, 0, c->xleaf_cap * 8));
  }
  c->cur_nleaves = nleaves;
  c->leaf_cap = nleaves;
  c->cur_maxnodes = 2 * nleaves + 1;
  c->tree_open = true;
  c->tree_valid = false;
  c->dbatch = false;
  return qr_k_tree_begin(c, nleaves, minls);
}

int qr_tree_decide(qr_ctx *c) {
  if (!c || !c->tree_open) return QR_ERR_STATE;
  return qr_k_tree_decide(c);
}

int qr_tree_apply(qr_ctx *c) {
  if (!c || !c->tree_open) return QR_ERR_STATE;
  return qr_k_tree_apply(c);
}

// the finished tree's compact records travel to pinned host memory behind the
// kernels that produced them; qr_tree_nodes waits for that copy only
static int snapshot_nodes(qr_ctx *c) {
  // k_leaf_final / k_leaf_global wrote the records into the pinned block directly, then
  // their sequence number (c->nodes_seq)
  c->nodes_pending = true;
  // (the documents this tree was grown on, for qr_tree_nodes' check of the records' arithmetic)
  c->nodes_grown_on = c->sub_k ? (uint64_t)c->sub_k : (uint64_t)(c->dmode ? c->Nglobal : c->N);
  return QR_OK;
}

// Batched growth enqueues a guessed number of steps (qr_k_tree_fit_batch).  Waits for the
// tree's last control call; if the device reports that the guess was too low, carries the tree on,
// repeats the leaf kernels (and the score update, if it was enqueued behind them: it left
// at once on the incomplete tree) and waits again.  Everything that consumes the tree or
// the scores calls this first; in the usual loop qr_tree_nodes does.
// (the word the last control call publishes: QrPinned::early)
static int wait_early(qr_ctx *c, int64_t *word_out) {
  for (unsigned spin = 1;; ++spin) {
    const int64_t w = __atomic_load_n(&c->h_pin->early[0], __ATOMIC_ACQUIRE);
    if ((w >> 16) == c->early_seq) {
      *word_out = w;
      return QR_OK;
    }
    if ((spin & 1023u) == 0) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) {  // everything enqueued has run
        QR_CHECK(c, hipStreamSynchronize(c->stream));
        const int64_t w2 = __atomic_load_n(&c->h_pin->early[0], __ATOMIC_ACQUIRE);
        if ((w2 >> 16) == c->early_seq) {
          *word_out = w2;
          return QR_OK;
        }
        QR_FAIL(c, QR_ERR_STATE, "internal: the tree's last control call was not published");
      }
      if (e != hipErrorNotReady) QR_CHECK(c, e);
    }
    cpu_relax(spin);
  }
}

// (a lambda pass whose scalars are still to be finished -- qr_lambda_compute defers them to the
// root scan launch of the tree that follows, qr_prep.h -- gets them finished here, in a launch
// of their own: whoever settles is about to read or change something they depend on.  The two
// entry points that can let them ride, qr_tree_fit and qr_oblivious_fit, settle with
// tree_settle_keep.)
static int tree_settle_keep(qr_ctx *c);
// keep_scores: a score update left to the next lambda pass stays pending (qr_lambda_compute:
// that pass is coming); everybody else is about to read or overwrite what it changes
static int tree_settle(qr_ctx *c, bool keep_scores) {
  int rc = tree_settle_keep(c);
  if (rc) return rc;
  if (!keep_scores && (rc = qr_k_scores_flush(c))) return rc;
  return qr_k_prep_flush(c);
}
static int tree_settle_keep(qr_ctx *c) {
  if (c->dbatch_pending) {
    // a document-sharded tree ended behind a guessed number of steps: carrying it on takes
    // all-reduces, which only the caller can enqueue (qr_tree_batch_settle)
    int64_t w = 0;
    const int rc = wait_early(c, &w);
    if (rc) return rc;
    if (w & 1)
      QR_FAIL(c, QR_ERR_STATE, "the last tree's enqueued steps did not suffice: qr_tree_batch_settle and the "
                               "steps it asks for come first (include/qr_hip.h)");
    c->dbatch_pending = c->dbatch_unsettled = false;
    ++c->spec_trees;
    c->steps_hint = (size_t)((w >> 1) & 0x7fff);
    if (c->steps_hint < 1) c->steps_hint = 1;
    c->spec_scores_enqueued = false;
  }
  if (!c->spec_pending) return QR_OK;
  // (the last control call's word, QrPinned::early -- not the records: the leaf kernels and the
  // score update behind that call are still running, and whatever the caller enqueues next
  // lines up under them)
  int64_t w = 0;
  int rc = wait_early(c, &w);
  if (rc) return rc;
  c->spec_pending = false;
  ++c->spec_trees;
  if (w & 1) ++c->spec_misses;
  // The guess was too low: the tree is carried on one step at first (the
  // tree usually needs just one more, and the worst case left would be a dozen launches that
  // find nothing to do), looking at the last control call's word after each piece.
  // (the piece doubles on every further miss -- 1, 2, 4, ... steps -- so a deep tree
  // behind a shallow one costs O(log k) host round trips, not k)
  size_t done = (size_t)c->tree_step;
  size_t piece_len = 1;
  while (w & 1) {
    // (pre-sorted lists, split search at the pop: a step that found no valid split used one of the
    // enqueued steps; every node is popped at most once, so 2 L + 1 steps bound the tree)
    const size_t cap = c->spec_exact ? 2 * c->cur_nleaves + 2 : c->cur_nleaves - 1;
    const size_t worst = cap > done ? cap - done : 1;
    const size_t piece = std::min(worst, piece_len);
    piece_len *= 2;
    if (c->spec_exact)
      rc = qr_k_exact_continue(c, piece);
    else
      rc = qr_k_tree_continue(c, c->cur_nleaves, c->cur_minls, done, piece);
    if (rc) return rc;
    done += piece;
    if ((rc = qr_k_tree_finish(c, c->spec_newton))) return rc;
    if (c->spec_scores_enqueued && (rc = qr_k_scores_update(c, c->spec_shrinkage, true))) return rc;
    if ((rc = wait_early(c, &w))) return rc;
    if ((w & 1) && piece == worst) QR_FAIL(c, QR_ERR_STATE, "internal: tree still incomplete after its last step");
  }
  c->spec_scores_enqueued = false;
  if (c->spec_exact) {  // (always nleaves - 1 steps: nothing to learn from this tree)
    c->spec_exact = false;
    return QR_OK;
  }
  // the next tree: as many steps as this one needed
  c->steps_hint = (size_t)((w >> 1) & 0x7fff);
  if (c->steps_hint < 1) c->steps_hint = 1;
  return QR_OK;
}

int qr_tree_nodes(qr_ctx *c, qr_node_t *nodes_out, size_t *nnodes_out) {
  if (!c) return QR_ERR_ARG;
  if (!c->nodes_pending) QR_FAIL(c, QR_ERR_STATE, "no fitted tree");
  // (the hosts read tree i's records between iteration i + 1's lambda pass and its tree: the
  // records do not depend on that pass's scalars, which stay deferred for the tree's root scan)
  int rc = tree_settle_keep(c);
  if (rc) return rc;
  { const int wrc = wait_seq64(c, &c->h_pin->tree.pad[2], c->nodes_seq, "the tree's records"); if (wrc) return wrc; }
  // The records validate themselves (QrNodeWire): the header's tag and every record's must fit the
  // sequence number just seen.  A record that does not fit yet is read again -- it has been
  // written ahead of the number; if it is not here, it is on its way -- and counted
  // (QR_SPEC_DEBUG=1 prints the count at context destruction).
  const uint64_t seq = (uint64_t)c->nodes_seq;
  size_t n = 0;
  std::vector<QrNodeWire> rec;
  for (long spin = 0;; ++spin) {
    const volatile QrNodesOut *t = &c->h_pin->tree;
    n = (size_t)t->nnodes;
    const int64_t inc = t->pad[0], steps = t->pad[1], htag = t->pad[3];
    bool ok = n <= QR_MAXNODES &&
              htag == (int64_t)(seq * 0x9E3779B97F4A7C15ull ^ ((uint64_t)n << 40) ^ ((uint64_t)inc << 32) ^ (uint64_t)steps);
    if (ok) {
      rec.resize(n);
      memcpy((void *)rec.data(), (const void *)c->h_pin->tree.nodes, n * sizeof(QrNodeWire));
      std::atomic_thread_fence(std::memory_order_acquire);
      for (size_t i = 0; i < n && ok; ++i) ok = rec[i].tag == qr_node_tag(rec[i], seq);
    }
    if (ok) break;
    ++c->readback_retries;
    if (spin > 20000000L) QR_FAIL(c, QR_ERR_STATE, "the tree's records do not fit their sequence number (pinned read-back)");
    cpu_relax(spin);
  }
  // The records' own arithmetic (round 6): the root holds the documents the tree was grown on, an internal
  // node as many as its two children -- counts that come from DIFFERENT launches (a node's from its parent's
  // scan, its children's from its own histogram).  Under the load of profiles/r06_hunt.md an XCD can lose
  // the stores of a histogram workgroup's partial: the tree then counts fewer documents than it was given,
  // which the hunt's run 7 (a root 4,096 documents short) and run 104 went on to train with.  Not any more.
  if (c->debug_root_short && n) {  // (test aid, qr_debug_bins_clobber(which = 3): a root histogram that lost a partial)
    rec[0].nsamples -= std::min<uint64_t>(rec[0].nsamples, c->debug_root_short);
    c->debug_root_short = 0;
  }
  {
    const uint64_t grown_on = c->nodes_grown_on;
    char what[200] = "";
    if (n && rec[0].nsamples != grown_on)
      snprintf(what, sizeof(what), "the root counts %llu documents of %llu", (unsigned long long)rec[0].nsamples,
               (unsigned long long)grown_on);
    for (size_t i = 0; i < n && !what[0]; ++i) {
      if (rec[i].feature < 0) continue;
      const int64_t l = rec[i].left, r = rec[i].right;
      if (l < 0 || r < 0 || (size_t)l >= n || (size_t)r >= n)
        snprintf(what, sizeof(what), "node %zu points outside the tree", i);
      else if (rec[l].nsamples + rec[r].nsamples != rec[i].nsamples)
        snprintf(what, sizeof(what), "node %zu counts %llu documents, its children %llu + %llu", i,
                 (unsigned long long)rec[i].nsamples, (unsigned long long)rec[l].nsamples,
                 (unsigned long long)rec[r].nsamples);
    }
    if (what[0]) {
      c->err = std::string("the tree's records do not add up (") + what +
               "): device memory lost stores (profiles/r06_hunt.md)";
      return QR_ERR_HIP;
    }
  }
  if (nodes_out) {
    for (size_t i = 0; i < n; ++i) rec[i].tag = 0;  // (qr_node_t's padding leaves as zeros)
    memcpy(nodes_out, rec.data(), n * sizeof(qr_node_t));
  }
  if (nnodes_out) *nnodes_out = n;
  return QR_OK;
}

int qr_tree_end(qr_ctx *c, int newton, qr_node_t *nodes_out, size_t *nnodes_out) {
  if (!c || !c->tree_open) return QR_ERR_STATE;
  int rc = qr_k_tree_finish(c, newton);
  if (rc) return rc;
  c->tree_open = false;
  if (c->dmode) {
    // (batched growth ended before its last control call was looked at: settled later)
    if (c->dbatch && c->dbatch_unsettled) c->dbatch_pending = true;
    return QR_OK;  // all-reduce the leaf buffer, then qr_tree_leaves_finish
  }
  c->tree_valid = true;
  if ((rc = snapshot_nodes(c))) return rc;
  if (nodes_out || nnodes_out) return qr_tree_nodes(c, nodes_out, nnodes_out);
  return QR_OK;
}

int qr_tree_leaves_finish(qr_ctx *c, int newton, qr_node_t *nodes_out, size_t *nnodes_out) {
  if (!c) return QR_ERR_ARG;
  if (!c->dmode || !c->d_xleaf || c->tree_open)
    QR_FAIL(c, QR_ERR_STATE, "qr_tree_leaves_finish follows qr_tree_end on a document-sharded context");
  int rc = qr_k_tree_leaves_global(c, newton);
  if (rc) return rc;
  if (c->dbatch_redo) {  // a carried-on tree: the score update enqueued behind its first end left at once
    c->dbatch_redo = false;
    if (c->spec_scores_enqueued && (rc = qr_k_scores_update(c, c->spec_shrinkage, true))) return rc;
    c->spec_scores_enqueued = false;
  }
  c->tree_valid = true;
  if ((rc = snapshot_nodes(c))) return rc;
  if (nodes_out || nnodes_out) return qr_tree_nodes(c, nodes_out, nnodes_out);
  return QR_OK;
}

int qr_tree_fit(qr_ctx *c, size_t nleaves, uint64_t minls, int newton,
                qr_node_t *nodes_out, size_t *nnodes_out) {
  if (!c) return QR_ERR_ARG;
  if (c->world > 1 || c->dmode)
    QR_FAIL(c, QR_ERR_STATE,
            "sharded contexts must drive qr_tree_begin/decide/apply/end "
            "with the collectives in between");
  {
    int src = tree_settle_keep(c);  // (deferred scalars ride in the root scan launch, or are flushed there)
    if (src) return src;
    // (a score update nobody consumed -- no lambda pass between two trees: the new tree
    // overwrites the leaf bytes and values it needs)
    if ((src = qr_k_scores_flush(c))) return src;
  }
  // up to QR_BATCH splits per step (k_decide_batch); per-node feature subsets are keyed by
  // the node's final index, which a split applied ahead of its turn does not know yet
  // (wide-bin contexts take the batched path too while their rows are short enough for the
  // one-launch scan and the provisional children's histogram slots stay small: round 3)
  const bool wide_ok = !c->wide || (qr_k_wide_batch_ok(c) && c->d_tree2 != nullptr &&
                                    (4 * nleaves + 1) * c->wcells * 12 <= ((size_t)4 << 30));
  if (!c->mf_k && !c->no_batch && wide_ok && nleaves >= 2 && 4 * nleaves + 1 <= QR_MAXNODES) {
    if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
    int rc = ensure_hist_slots(c, 4 * nleaves + 1);
    if (rc) return rc;
    size_t depth = 1;
    while (((size_t)1 << (depth - 1)) < QR_BATCH) ++depth;
    if ((rc = ensure_level_buffers(c, depth))) return rc;
    c->cur_nleaves = nleaves;
    c->leaf_cap = nleaves;
    c->cur_maxnodes = 2 * nleaves + 1;
    c->tree_open = true;
    c->tree_valid = false;
    if ((rc = qr_k_tree_fit_batch(c, nleaves, minls))) return rc;
    // (the records say how many steps the tree took and whether the enqueued ones sufficed)
    c->spec_pending = true;
    c->spec_scores_enqueued = false;
    c->spec_newton = newton;
    return qr_tree_end(c, newton, nodes_out, nnodes_out);
  }
  // pre-sorted lists: the split search of a node when the loop pops it (k_exact.hip qr_k_exact_fit)
  // (round 5: under --subsample too -- the root lists are cut down to the sample per tree)
  if (c->xmode && !c->x_eager && nleaves >= 2 && 2 * nleaves + 1 <= QR_MAXNODES) {
    if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
    c->cur_nleaves = nleaves;
    c->leaf_cap = nleaves;
    c->cur_maxnodes = 2 * nleaves + 1;
    c->tree_open = true;
    c->tree_valid = false;
    c->dbatch = false;
    int xrc = qr_k_exact_fit(c, nleaves, minls);
    if (xrc) return xrc;
    c->spec_pending = true;   // (the last control call says whether the enqueued steps sufficed)
    c->spec_exact = true;
    c->spec_scores_enqueued = false;
    c->spec_newton = newton;
    return qr_tree_end(c, newton, nodes_out, nnodes_out);
  }
  int rc = qr_tree_begin(c, nleaves, minls);
  if (rc) return rc;
  for (size_t s = 0; s + 1 < nleaves; ++s) {
    if ((rc = qr_k_tree_decide(c))) return rc;
    if ((rc = qr_k_tree_apply(c))) return rc;
  }
  if ((rc = qr_k_tree_decide(c))) return rc;
  return qr_tree_end(c, newton, nodes_out, nnodes_out);
}

int qr_oblivious_fit(qr_ctx *c, size_t depth, uint64_t minls, int newton,
                     qr_node_t *nodes_out, size_t *nnodes_out) {
  if (!c) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  { const int src_ = tree_settle_keep(c); if (src_) return src_; }
  { const int src_ = qr_k_scores_flush(c); if (src_) return src_; }
  if (c->world > 1 || c->dmode)
    QR_FAIL(c, QR_ERR_STATE,
            "sharded contexts grow oblivious trees phase by phase (qr_obl_begin / propose / [mark] / "
            "apply with the collectives in between)");
  if (depth < 1 || ((size_t)1 << (depth + 1)) - 1 > QR_MAXNODES)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "tree depth must be in [1, 9]");
  int rc = ensure_hist_slots(c, ((size_t)1 << (depth + 1)) - 1);
  if (rc) return rc;
  if ((rc = ensure_level_buffers(c, depth))) return rc;
  c->tree_valid = false;
  c->tree_open = true;
  c->leaf_cap = (size_t)1 << depth;
  c->cur_maxnodes = ((size_t)1 << (depth + 1)) - 1;
  if ((rc = qr_k_oblivious_fit(c, depth, minls))) return rc;
  return qr_tree_end(c, newton, nodes_out, nnodes_out);
}

// ---- document-sharded leaf-wise trees with up to QR_BATCH splits per step -----------------
// (what qr_tree_fit does on one GPU, cut at the all-reduces: include/qr_hip.h)
int qr_tree_batch_supported(qr_ctx *c, size_t nleaves) {
  if (!c || !c->binned || !c->dmode || c->wide || c->mf_k || c->no_batch) return 0;
  return nleaves >= 2 && 4 * nleaves + 1 <= QR_MAXNODES ? 1 : 0;
}

int qr_tree_batch_begin(qr_ctx *c, size_t nleaves, uint64_t minls, size_t *steps_out) {
  if (!c) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  if (!qr_tree_batch_supported(c, nleaves))
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "qr_tree_batch_begin: a document-sharded context with u8 bins, every feature "
                                   "at every node and 2 <= nleaves <= 255 (otherwise qr_tree_begin / decide / apply)");
  int rc = tree_settle(c);
  if (rc) return rc;
  if ((rc = ensure_hist_slots(c, 4 * nleaves + 1))) return rc;
  size_t depth = 1;
  while (((size_t)1 << (depth - 1)) < QR_BATCH) ++depth;
  if ((rc = ensure_level_buffers(c, depth))) return rc;
  const size_t need = (size_t)QR_BATCH * c->flocal * 512 + (size_t)QR_BATCH * 2 * c->world;
  if (need > c->xb_len || 2 * nleaves * (size_t)c->world > c->xleaf_cap) QR_CHECK(c, hipStreamSynchronize(c->stream));
  if (need > c->xb_len) {
    dfree(c->d_xb);
    QR_CHECK(c, dalloc(&c->d_xb, need));
    QR_CHECK(c, hipMemset(c->d_xb, 0, need * 8));
    c->xb_len = need;
  }
  if (2 * nleaves * (size_t)c->world > c->xleaf_cap) {
    dfree(c->d_xleaf);
    c->xleaf_cap = 2 * nleaves * (size_t)c->world;
    QR_CHECK(c, dalloc(&c->d_xleaf, c->xleaf_cap));
    QR_CHECK(c, hipMemset(c->d_xleaf, 0, c->xleaf_cap * 8));
  }
  c->cur_nleaves = nleaves;
  c->leaf_cap = nleaves;
  c->cur_maxnodes = 2 * nleaves + 1;
  c->tree_open = true;
  c->tree_valid = false;
  c->dbatch = true;
  c->dbatch_unsettled = c->dbatch_pending = c->dbatch_redo = false;
  // the guess: as many steps as the last tree needed (+ QR_STEPS_PLUS); every rank grows the
  // same trees, so every rank guesses the same number
  size_t steps = nleaves - 1;
  if (c->steps_force >= 0)
    steps = std::min<size_t>(steps, (size_t)std::max<long>(c->steps_force, 1));
  else if (c->steps_hint)
    steps = std::min(steps, c->steps_hint);
  if (steps < 1) steps = 1;
  if (steps_out) *steps_out = steps;
  return qr_k_dbatch_root_hist(c, nleaves, minls);
}

int qr_tree_batch_root(qr_ctx *c) {
  if (!c || !c->tree_open || !c->dbatch) return QR_ERR_STATE;
  return qr_k_dbatch_root_decide(c, c->cur_nleaves, c->cur_minls);
}

int qr_tree_batch_apply(qr_ctx *c) {
  if (!c || !c->tree_open || !c->dbatch) return QR_ERR_STATE;
  return qr_k_dbatch_apply(c, c->cur_nleaves);
}

int qr_tree_batch_decide(qr_ctx *c, int last) {
  if (!c || !c->tree_open || !c->dbatch) return QR_ERR_STATE;
  if (last) c->dbatch_unsettled = true;
  return qr_k_dbatch_decide(c, c->cur_nleaves, c->cur_minls, last ? 1 : 0);
}

int qr_tree_batch_exchange(qr_ctx *c, void **cells, size_t *cells_i64) {
  if (!c) return QR_ERR_ARG;
  if (!c->dmode || !c->d_xb) QR_FAIL(c, QR_ERR_STATE, "qr_tree_batch_exchange follows qr_tree_batch_begin on a document-sharded context");
  if (cells) *cells = c->d_xb;
  if (cells_i64) *cells_i64 = c->xb_len;
  return QR_OK;
}

int qr_tree_batch_settle(qr_ctx *c, int *incomplete, size_t *steps_used) {
  if (!c || !incomplete) return QR_ERR_ARG;
  if (!c->dbatch || !c->dbatch_unsettled) {  // nothing to look at: the last tree is settled
    *incomplete = 0;
    if (steps_used) *steps_used = 0;
    return QR_OK;
  }
  int64_t w = 0;
  const int rc = wait_early(c, &w);
  if (rc) return rc;
  c->dbatch_unsettled = false;
  *incomplete = (int)(w & 1);
  const size_t used = (size_t)((w >> 1) & 0x7fff);
  if (steps_used) *steps_used = used;
  if (!(w & 1)) {
    ++c->spec_trees;
    c->steps_hint = used;
    if (c->steps_hint < 1) c->steps_hint = 1;
    if (c->dbatch_pending) c->spec_scores_enqueued = false;  // (the tree was whole: so is its score update)
    c->dbatch_pending = false;
  } else {
    ++c->spec_misses;
    if (c->dbatch_pending) {
      // the tree was ended behind the guess: its leaf kernels and score update left at once.
      // Open again; the caller carries it on and ends it again (qr_tree_end -> all-reduce ->
      // qr_tree_leaves_finish, which repeats the score update if one was enqueued)
      c->dbatch_pending = false;
      c->dbatch_redo = true;
      c->tree_open = true;
      c->tree_valid = false;
    }
  }
  return QR_OK;
}

// ---- feature-sharded oblivious trees, phase by phase ------------------------------------
int qr_obl_begin(qr_ctx *c, size_t depth, uint64_t minls) {
  if (!c) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  if (c->wide) QR_FAIL(c, QR_ERR_UNSUPPORTED, "level-wise growth on sharded contexts uses u8 bins (--num-thresholds in [1, 255])");
  if (depth < 1 || ((size_t)1 << (depth + 1)) - 1 > QR_MAXNODES)
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "tree depth must be in [1, 9]");
  int rc = tree_settle(c);
  if (rc) return rc;
  if ((rc = ensure_hist_slots(c, ((size_t)1 << (depth + 1)) - 1))) return rc;
  if ((rc = ensure_level_buffers(c, depth))) return rc;
  c->tree_valid = false;
  c->tree_open = true;
  c->leaf_cap = (size_t)1 << depth;
  c->cur_depth = depth;
  c->cur_maxnodes = ((size_t)1 << (depth + 1)) - 1;
  if (c->dmode) {
    // document-sharded: the level's exchange buffer ([node][feature][slot][sum, count] for the
    // widest level that builds histograms) and the leaves' (as qr_tree_begin)
    const size_t nodes = depth >= 2 ? (size_t)1 << (depth - 2) : 1;
    const size_t need = nodes * (size_t)c->flocal * 256 * 2;
    const size_t leaves = (size_t)1 << depth;
    if (need > c->xlevel_cap || 2 * leaves * (size_t)c->world > c->xleaf_cap) QR_CHECK(c, hipStreamSynchronize(c->stream));
    if (need > c->xlevel_cap) {
      dfree(c->d_xlevel);
      QR_CHECK(c, dalloc(&c->d_xlevel, need));
      QR_CHECK(c, hipMemset(c->d_xlevel, 0, need * 8));
      c->xlevel_cap = need;
    }
    if (2 * leaves * (size_t)c->world > c->xleaf_cap) {
      dfree(c->d_xleaf);
      c->xleaf_cap = 2 * leaves * (size_t)c->world;
      QR_CHECK(c, dalloc(&c->d_xleaf, c->xleaf_cap));
      QR_CHECK(c, hipMemset(c->d_xleaf, 0, c->xleaf_cap * 8));
    }
    c->cur_nleaves = leaves;
  }
  return qr_k_obl_begin(c, depth, minls);
}
int qr_obl_propose(qr_ctx *c, size_t level) {
  if (!c || !c->tree_open || level >= c->cur_depth) return QR_ERR_STATE;
  return qr_k_obl_propose(c, (int)level);
}
int qr_obl_mark(qr_ctx *c, size_t level) {
  if (!c || !c->tree_open || level >= c->cur_depth) return QR_ERR_STATE;
  if (c->dmode) QR_FAIL(c, QR_ERR_STATE, "document-sharded ranks partition their own documents: no mask to exchange");
  return qr_k_obl_mark(c, (int)level);
}
int qr_obl_apply(qr_ctx *c, size_t level) {
  if (!c || !c->tree_open || level >= c->cur_depth) return QR_ERR_STATE;
  return qr_k_obl_apply(c, (int)level, level + 1 == c->cur_depth);
}
int qr_obl_level_exchange(qr_ctx *c, size_t level, void **cells, size_t *cells_i64) {
  if (!c) return QR_ERR_ARG;
  if (!c->dmode || !c->d_xlevel) QR_FAIL(c, QR_ERR_STATE, "qr_obl_level_exchange follows qr_obl_begin on a document-sharded context");
  if (cells) *cells = c->d_xlevel;
  if (cells_i64) *cells_i64 = ((size_t)1 << level) * (size_t)c->flocal * 256 * 2;
  return QR_OK;
}
int qr_obl_exchange_buffers(qr_ctx *c, void **recs_local, void **recs_all, size_t *rec_bytes_per_rank,
                            void **mask, size_t *mask_bytes) {
  if (!c) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  if (recs_local) *recs_local = c->d_recs_local;
  if (recs_all) *recs_all = c->d_recs_all;
  if (rec_bytes_per_rank) *rec_bytes_per_rank = 2 * sizeof(qr_split_t);
  if (mask) *mask = c->d_mask;
  if (mask_bytes) *mask_bytes = (c->mask_words + QR_MAXLEVEL) * 4;
  return QR_OK;
}

int qr_scores_update(qr_ctx *c, double shrinkage) {
  if (!c) return QR_ERR_ARG;
  if (!c->tree_valid) QR_FAIL(c, QR_ERR_STATE, "no fitted tree");
  if (c->spec_pending || c->dbatch_pending) {  // (if the tree turns out incomplete, settling it repeats this update)
    c->spec_scores_enqueued = true;
    c->spec_shrinkage = shrinkage;
  }
  return qr_k_scores_update(c, shrinkage);
}

int qr_exchange_buffers(qr_ctx *c, void **recs_local, void **recs_all,
                        size_t *rec_bytes_per_rank, void **mask, size_t *mask_bytes) {
  if (!c) return QR_ERR_ARG;
  if (!c->binned) QR_FAIL(c, QR_ERR_STATE, "bins not built");
  if (recs_local) *recs_local = c->d_recs_local;
  if (recs_all) *recs_all = c->d_recs_all;
  if (rec_bytes_per_rank) *rec_bytes_per_rank = 2 * sizeof(qr_split_t);
  if (mask) *mask = c->d_mask;
  if (mask_bytes) *mask_bytes = (c->mask_words + (c->wide ? 1 : 0)) * 4;  // (wide: + the threshold word)
  return QR_OK;
}

int qr_doc_exchange_buffers(qr_ctx *c, void **hist, size_t *hist_i64, void **scal,
                            size_t *scal_i64, void **leaf, size_t *leaf_i64) {
  if (!c) return QR_ERR_ARG;
  if (!c->binned || !c->dmode) QR_FAIL(c, QR_ERR_STATE, "not a binned document-sharded context");
  if (hist) *hist = c->d_xh;
  if (hist_i64) *hist_i64 = c->xh_len;
  if (scal) *scal = c->d_xscal;
  if (scal_i64) *scal_i64 = 4 * (size_t)c->world;
  if (leaf) *leaf = c->d_xleaf;  // valid after qr_tree_begin
  if (leaf_i64) *leaf_i64 = 2 * c->cur_nleaves * (size_t)c->world;
  return QR_OK;
}

// ---------------------------------------------------------------------------
static int read_tree(qr_ctx *c, std::vector<char> &buf) {
  if (!c->tree_valid) QR_FAIL(c, QR_ERR_STATE, "no fitted tree");
  { const int src_ = tree_settle(c); if (src_) return src_; }
  buf.resize(sizeof(QrTreeState));
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(buf.data(), c->d_tree, sizeof(QrTreeState), hipMemcpyDeviceToHost));
  return QR_OK;
}

int qr_node_hist_read(qr_ctx *c, int node, double *sum_out, uint64_t *count_out) {
  if (!c) return QR_ERR_ARG;
  std::vector<char> buf;
  int rc = read_tree(c, buf);
  if (rc) return rc;
  const QrTreeState &ts = *reinterpret_cast<QrTreeState *>(buf.data());
  if (node < 0 || node >= ts.nnodes) QR_FAIL(c, QR_ERR_ARG, "node out of range");
  if (c->wide) QR_FAIL(c, QR_ERR_STATE, "this context has more than 255 thresholds per feature: qr_node_hist_read_ragged");
  QrScalars s;
  QR_CHECK(c, hipMemcpy(&s, c->d_scalars, sizeof(s), hipMemcpyDeviceToHost));
  const size_t n = (size_t)c->flocal * 256;
  std::vector<long long> hs(n);
  std::vector<uint32_t> hc(n);
  const size_t slot = (size_t)ts.nodes[node].hslot;
  QR_CHECK(c, hipMemcpy(hs.data(), c->d_hsum + slot * n, n * 8, hipMemcpyDeviceToHost));
  QR_CHECK(c, hipMemcpy(hc.data(), c->d_hcnt + slot * n, n * 4, hipMemcpyDeviceToHost));
  if (sum_out) memset(sum_out, 0, c->F * 256 * 8);
  if (count_out) memset(count_out, 0, c->F * 256 * 8);
  for (int lf = 0; lf < c->flocal; ++lf) {
    const size_t gf = (size_t)c->h_lf2gf[lf];
    for (int t = 0; t < 256; ++t) {
      if (sum_out) sum_out[gf * 256 + t] = (double)hs[(size_t)lf * 256 + t] * s.inv_scale;
      if (count_out) count_out[gf * 256 + t] = hc[(size_t)lf * 256 + t];
    }
  }
  return QR_OK;
}

int qr_node_hist_read_ragged(qr_ctx *c, int node, double *sum_out, uint64_t *count_out) {
  if (!c) return QR_ERR_ARG;
  std::vector<char> buf;
  int rc = read_tree(c, buf);
  if (rc) return rc;
  const QrTreeState &ts = *reinterpret_cast<QrTreeState *>(buf.data());
  if (node < 0 || node >= ts.nnodes) QR_FAIL(c, QR_ERR_ARG, "node out of range");
  QrScalars s;
  QR_CHECK(c, hipMemcpy(&s, c->d_scalars, sizeof(s), hipMemcpyDeviceToHost));
  const size_t slot = (size_t)ts.nodes[node].hslot;
  if (c->xmode && (!c->sub_k || !c->x_eager))
    QR_FAIL(c, QR_ERR_UNSUPPORTED, "this context grows its trees on pre-sorted lists (rows of more than "
                                   "16384 slots): there are no node histograms to read");
  if (c->wide) {
    const size_t n = c->wcells;
    std::vector<long long> hs(n);
    std::vector<uint32_t> hc(n);
    QR_CHECK(c, hipMemcpy(hs.data(), c->d_hsum + slot * n, n * 8, hipMemcpyDeviceToHost));
    QR_CHECK(c, hipMemcpy(hc.data(), c->d_hcnt + slot * n, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      if (sum_out) sum_out[i] = (double)hs[i] * s.inv_scale;
      if (count_out) count_out[i] = hc[i];
    }
    return QR_OK;
  }
  std::vector<double> ps(c->F * 256);
  std::vector<uint64_t> pc(c->F * 256);
  if ((rc = qr_node_hist_read(c, node, ps.data(), pc.data()))) return rc;
  size_t o = 0;
  for (size_t f = 0; f < c->F; ++f)
    for (uint32_t t = 0; t < c->h_thr_size[f]; ++t, ++o) {
      if (sum_out) sum_out[o] = ps[f * 256 + t];
      if (count_out) count_out[o] = pc[f * 256 + t];
    }
  return QR_OK;
}

int qr_node_samples_read(qr_ctx *c, int node, uint32_t *ids_out, size_t *n_out) {
  if (!c) return QR_ERR_ARG;
  std::vector<char> buf;
  int rc = read_tree(c, buf);
  if (rc) return rc;
  const QrTreeState &ts = *reinterpret_cast<QrTreeState *>(buf.data());
  if (node < 0 || node >= ts.nnodes) QR_FAIL(c, QR_ERR_ARG, "node out of range");
  const QrNode &nd = ts.nodes[node];
  const size_t n = nd.end - nd.begin;
  if (n_out) *n_out = n;
  if (!ids_out) return QR_OK;
  if (nd.buf == 2) {
    for (size_t i = 0; i < n; ++i) ids_out[i] = (uint32_t)(nd.begin + i);
  } else {
    QR_CHECK(c, hipMemcpy(ids_out, c->d_order[nd.buf] + nd.begin, n * 4, hipMemcpyDeviceToHost));
  }
  return QR_OK;
}

int qr_tree_split_log(qr_ctx *c, qr_split_t *out, size_t *n_out) {
  if (!c) return QR_ERR_ARG;
  std::vector<char> buf;
  int rc = read_tree(c, buf);
  if (rc) return rc;
  const QrTreeState &ts = *reinterpret_cast<QrTreeState *>(buf.data());
  if (n_out) *n_out = (size_t)ts.nsplits;
  if (out) memcpy(out, ts.split_log, ts.nsplits * sizeof(qr_split_t));
  return QR_OK;
}

int qr_metric_per_query(qr_ctx *c, double *out) {
  if (!c || !out || !c->d_qmetric) return QR_ERR_ARG;
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(out, c->d_qmetric, c->Q * 8, hipMemcpyDeviceToHost));
  return QR_OK;
}

int qr_ranks_read(qr_ctx *c, uint32_t *out) {
  if (!c || !out || !c->d_ranks) return QR_ERR_ARG;
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  QR_CHECK(c, hipMemcpy(out, c->d_ranks, c->N * 4, hipMemcpyDeviceToHost));
  return QR_OK;
}

// ---------------------------------------------------------------------------
// Compact binned form of the model for k_score_bin: per-feature sorted distinct
// thresholds, internal nodes as {feature, threshold index, children}, leaves apart.
static int build_binned_model(qr_ctx *c, const qr_node_t *nodes, size_t ntrees, size_t max_nodes,
                              const double *weights) {
  c->sb_ready = false;
  c->p4_ready = false;
  int maxf = -1;
  for (size_t i = 0; i < ntrees * max_nodes; ++i) maxf = std::max(maxf, nodes[i].feature);
  if (maxf < 0) return QR_OK;  // only leaves: the generic kernel handles it
  const size_t F = (size_t)maxf + 1;
  std::vector<std::vector<float>> thr(F);
  // reachable nodes only
  std::vector<std::vector<int>> order(ntrees);
  size_t NI = 1, NL = 1;
  for (size_t t = 0; t < ntrees; ++t) {
    const qr_node_t *n = nodes + t * max_nodes;
    std::vector<int> stack = {0};
    size_t ni = 0, nl = 0;
    while (!stack.empty()) {
      const int i = stack.back();
      stack.pop_back();
      if (i < 0 || (size_t)i >= max_nodes) return QR_OK;  // malformed: keep the generic kernel
      order[t].push_back(i);
      if (n[i].feature >= 0) {
        ++ni;
        thr[n[i].feature].push_back(n[i].threshold);
        stack.push_back(n[i].right);
        stack.push_back(n[i].left);
      } else
        ++nl;
    }
    NI = std::max(NI, ni);
    NL = std::max(NL, nl);
  }
  if (NI > 0x7fff || NL > 0x7fff) return QR_OK;
  size_t tmax = 1;
  for (auto &v : thr) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());  // -0.0 == 0.0: same comparisons
    tmax = std::max(tmax, v.size());
  }
  if (tmax > 65535) return QR_OK;
  // Self-looping leaves (k_score_bin<.., SELF>): the node array holds NI internal entries and
  // NL leaf entries per tree, children are plain indices into it, and an internal node names
  // the byte offset of its feature's bin row -- whenever that offset fits 16 bits.
  const size_t bt_size = tmax <= 255 ? 1 : 2;
  const bool self = (F - 1) * 64 * bt_size <= 0xffff && NI + NL <= 0xffff;
  const size_t NN = self ? NI + NL : NI;
  std::vector<uint16_t> cn(ntrees * NN * 4, 0), root(ntrees, 0);
  std::vector<double> leaves(ntrees * NL, 0.0);
  for (size_t t = 0; t < ntrees; ++t) {
    const qr_node_t *n = nodes + t * max_nodes;
    std::vector<int> idx(max_nodes, -1);
    size_t ni = 0, nl = 0;
    for (int i : order[t])
      idx[i] = n[i].feature >= 0 ? (int)ni++ : (self ? (int)(NI + nl++) : (int)(0x8000 | nl++));
    for (int i : order[t]) {
      if (n[i].feature >= 0) {
        uint16_t *o = &cn[(t * NN + idx[i]) * 4];
        const auto &v = thr[n[i].feature];
        o[0] = self ? (uint16_t)((size_t)n[i].feature * 64 * bt_size) : (uint16_t)n[i].feature;
        o[1] = (uint16_t)(std::lower_bound(v.begin(), v.end(), n[i].threshold) - v.begin());
        o[2] = (uint16_t)idx[n[i].left];
        o[3] = (uint16_t)idx[n[i].right];
      } else {
        const size_t li = self ? (size_t)idx[i] - NI : (size_t)(idx[i] & 0x7fff);
        leaves[t * NL + li] = n[i].value;
        if (self) {
          uint16_t *o = &cn[(t * NN + idx[i]) * 4];
          o[0] = 0;
          o[1] = 0xffff;  // any bin <= 0xffff: the walk stays here
          o[2] = o[3] = (uint16_t)idx[i];
        }
      }
    }
    root[t] = (uint16_t)idx[0];
  }
  std::vector<float> tt(F * tmax, 0.0f);
  std::vector<uint32_t> tc(F, 0);
  for (size_t f = 0; f < F; ++f) {
    tc[f] = (uint32_t)thr[f].size();
    std::copy(thr[f].begin(), thr[f].end(), tt.begin() + f * tmax);
  }
  if (c->d_sb_nodes) (void)hipFree(c->d_sb_nodes);
  c->d_sb_nodes = nullptr;
  dfree(c->d_sb_leaves); dfree(c->d_sb_root); dfree(c->d_sb_thr); dfree(c->d_sb_thr_cnt);
  QR_CHECK(c, hipMalloc(&c->d_sb_nodes, cn.size() * 2));
  QR_CHECK(c, dalloc(&c->d_sb_leaves, leaves.size()));
  QR_CHECK(c, dalloc(&c->d_sb_root, root.size()));
  QR_CHECK(c, dalloc(&c->d_sb_thr, tt.size()));
  QR_CHECK(c, dalloc(&c->d_sb_thr_cnt, tc.size()));
  QR_CHECK(c, hipMemcpy(c->d_sb_nodes, cn.data(), cn.size() * 2, hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_sb_leaves, leaves.data(), leaves.size() * 8, hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_sb_root, root.data(), root.size() * 2, hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_sb_thr, tt.data(), tt.size() * 4, hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_sb_thr_cnt, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
  c->sb_F = F;
  c->sb_NI = NI;
  c->sb_NL = NL;
  c->sb_tmax = tmax;
  c->sb_u8 = tmax <= 255;
  c->sb_self = self;
  c->sb_ready = true;
  // ---- the 4-byte records of k_score_p4, when the model allows them
  c->p4_ready = false;
  dfree(c->d_p4_batches); dfree(c->d_p4_depth);
  size_t NN4 = 0;
  for (size_t t = 0; t < ntrees; ++t) NN4 = std::max(NN4, order[t].size());
  if (tmax <= 255 && ((F - 1) / 4) * 256 + 3 <= 0xffff && NN4 <= 255) {
    const size_t NNP = NN4 <= 128 ? 128 : 256, T16 = (ntrees + 15) / 16 * 16;
    std::vector<uint32_t> w(T16 * NNP, 0u);  // unused entries and padding trees: {0, 0, node 0}
    std::vector<double> lv(T16 * NNP, 0.0);
    std::vector<uint8_t> dep(T16 / 8, 0);
    std::vector<int> pos(max_nodes), seq, dseq;
    for (size_t t = 0; t < ntrees; ++t) {
      const qr_node_t *n = nodes + t * max_nodes;
      // level order, siblings adjacent: the right child sits at the left one's position + 1 and
      // the nodes of a level are consecutive dwords (one bank each)
      seq.assign(1, 0);
      dseq.assign(1, 0);
      int maxd = 0;
      pos[0] = 0;
      for (size_t h = 0; h < seq.size(); ++h) {
        const int i = seq[h], d = dseq[h];
        if (n[i].feature >= 0) {
          maxd = std::max(maxd, d + 1);
          pos[n[i].left] = (int)seq.size();
          seq.push_back(n[i].left);
          dseq.push_back(d + 1);
          pos[n[i].right] = (int)seq.size();
          seq.push_back(n[i].right);
          dseq.push_back(d + 1);
        }
      }
      if (maxd > 255) return QR_OK;  // (cannot happen with <= 255 nodes)
      dep[t / 8] = std::max(dep[t / 8], (uint8_t)maxd);
      for (int i : seq) {
        const size_t at = t * NNP + (size_t)pos[i];
        if (n[i].feature >= 0) {
          const auto &v = thr[n[i].feature];
          const uint32_t kb = (uint32_t)(std::lower_bound(v.begin(), v.end(), n[i].threshold) - v.begin());
          const uint32_t row = (uint32_t)(n[i].feature / 4) * 256u + (uint32_t)(n[i].feature % 4);
          w[at] = row | ((255u - kb) << 16) | ((uint32_t)pos[n[i].left] << 24);
        } else {
          w[at] = (uint32_t)pos[i] << 24;
          lv[at] = n[i].value * weights[t];  // ensemble.cc:111-118's product, once per model
        }
      }
    }
    // batch by batch: the 16 record tiles, then the 16 value tiles (k_score_p4's LDS image)
    const size_t bdw = 16 * NNP * 3;  // dwords per batch
    std::vector<uint32_t> img(T16 / 16 * bdw);
    for (size_t b = 0; b < T16 / 16; ++b) {
      std::memcpy(&img[b * bdw], &w[b * 16 * NNP], 16 * NNP * 4);
      std::memcpy(&img[b * bdw + 16 * NNP], &lv[b * 16 * NNP], 16 * NNP * 8);
    }
    QR_CHECK(c, dalloc(&c->d_p4_batches, img.size()));
    QR_CHECK(c, dalloc(&c->d_p4_depth, dep.size() + 4));  // (read as dwords)
    QR_CHECK(c, hipMemcpy(c->d_p4_batches, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    QR_CHECK(c, hipMemcpy(c->d_p4_depth, dep.data(), dep.size(), hipMemcpyHostToDevice));
    c->p4_NNP = NNP;
    c->p4_ready = true;
  }
  return QR_OK;
}

int qr_ensemble_set_depth_order(qr_ctx *c, int on) {
  if (!c) return QR_ERR_ARG;
  c->ens_depth_order = on != 0;
  return QR_OK;
}

// levels of a tree's deepest leaf (0: a single leaf); -1 if the links leave the array
static int tree_depth(const qr_node_t *n, size_t max_nodes) {
  std::vector<std::pair<int, int>> stack = {{0, 0}};
  int maxd = 0;
  size_t seen = 0;
  while (!stack.empty()) {
    const auto [i, d] = stack.back();
    stack.pop_back();
    if (i < 0 || (size_t)i >= max_nodes || ++seen > max_nodes) return -1;
    if (n[i].feature >= 0) {
      maxd = std::max(maxd, d + 1);
      stack.push_back({n[i].right, d + 1});
      stack.push_back({n[i].left, d + 1});
    }
  }
  return maxd;
}

int qr_ensemble_upload(qr_ctx *c, const qr_node_t *nodes_in, size_t ntrees,
                       size_t max_nodes, const double *weights_in) {
  if (!c || !nodes_in || !weights_in || !ntrees || !max_nodes) return QR_ERR_ARG;
  QR_CHECK(c, hipSetDevice(c->device));
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  // Depth order (qr_ensemble_set_depth_order; off by default): the trees
  // are walked -- and their f64 contributions ADDED -- in ascending order of their depth (stable),
  // so that the trees a wave walks in lockstep end together.  The sum of ensemble.cc:111-118 is
  // then taken in another order: equal to the reference's to f64 rounding (~1e-16 relative per
  // tree, against north_star's 1e-5), not bit for bit.  Partial scores keep the model's columns.
  const qr_node_t *nodes = nodes_in;
  const double *weights = weights_in;
  std::vector<qr_node_t> pn;
  std::vector<double> pw;
  c->ens_perm.clear();
  if (c->ens_depth_order) {
    std::vector<int> depth(ntrees);
    bool ok = true, differ = false;
    for (size_t t = 0; t < ntrees && ok; ++t) {
      depth[t] = tree_depth(nodes_in + t * max_nodes, max_nodes);
      ok = depth[t] >= 0;
      differ = differ || depth[t] != depth[0];
    }
    if (ok && differ) {
      std::vector<uint32_t> perm(ntrees);
      for (size_t t = 0; t < ntrees; ++t) perm[t] = (uint32_t)t;
      std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return depth[a] < depth[b]; });
      pn.resize(ntrees * max_nodes);
      pw.resize(ntrees);
      for (size_t t = 0; t < ntrees; ++t) {
        memcpy(&pn[t * max_nodes], nodes_in + (size_t)perm[t] * max_nodes, max_nodes * sizeof(qr_node_t));
        pw[t] = weights_in[perm[t]];
      }
      nodes = pn.data();
      weights = pw.data();
      c->ens_perm = perm;  // position in the walk -> the model's tree
    }
  }
  dfree(c->d_ens);
  dfree(c->d_ens_w);
  QR_CHECK(c, dalloc(&c->d_ens, ntrees * max_nodes));
  QR_CHECK(c, dalloc(&c->d_ens_w, ntrees));
  QR_CHECK(c, hipMemcpy(c->d_ens, nodes, ntrees * max_nodes * sizeof(qr_node_t), hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_ens_w, weights, ntrees * 8, hipMemcpyHostToDevice));
  c->ens_trees = ntrees;
  c->ens_maxnodes = max_nodes;
  c->ens_maxf = -1;
  for (size_t i = 0; i < ntrees * max_nodes; ++i) c->ens_maxf = std::max(c->ens_maxf, (int)nodes[i].feature);
  return build_binned_model(c, nodes, ntrees, max_nodes, weights);
}

// features with index >= sb_F are never tested by the model, so a wider matrix
// is fine for the fast path as long as the row stride is passed along
static int score_any(qr_ctx *c, const float *d_x, size_t N, size_t F, double *d_out) {
  if (!c->d_ens) QR_FAIL(c, QR_ERR_STATE, "no ensemble uploaded");
  if (c->ens_maxf >= 0 && F <= (size_t)c->ens_maxf)
    QR_FAIL(c, QR_ERR_ARG, "the model tests feature " + std::to_string(c->ens_maxf + 1) +
                               " but the matrix has only " + std::to_string(F) + " columns");
  if (c->sb_ready && F >= c->sb_F) {
    const int rc = qr_k_ensemble_score_fast(c, d_x, N, F, d_out);
    if (rc >= 0) return rc;
  }
  return qr_k_ensemble_score(c, d_x, N, F, d_out);
}

int qr_ensemble_score_device(qr_ctx *c, const void *d_x, size_t N, size_t F, void *d_out) {
  if (!c || !d_x || !d_out) return QR_ERR_ARG;
  return score_any(c, (const float *)d_x, N, F, (double *)d_out);
}

int qr_ensemble_score(qr_ctx *c, const float *x, size_t N, size_t F, double *out,
                      float *kernel_ms) {
  if (!c || !x || !out || !N || !F) return QR_ERR_ARG;
  float *d_x = nullptr;
  double *d_o = nullptr;
  // a test file narrower than the model (its largest feature id is smaller): the
  // missing columns are zeros, as for any absent SVMLight feature
  size_t Fd = F;
  if (c->ens_maxf >= 0 && F <= (size_t)c->ens_maxf) Fd = (size_t)c->ens_maxf + 1;
  QR_CHECK(c, dalloc(&d_x, N * Fd));
  QR_CHECK(c, dalloc(&d_o, N));
  if (Fd == F) {
    QR_CHECK(c, hipMemcpy(d_x, x, N * F * 4, hipMemcpyHostToDevice));
  } else {
    const std::vector<float> padded = repad_rows(x, N, F, Fd);
    QR_CHECK(c, hipMemcpy(d_x, padded.data(), N * Fd * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  QR_CHECK(c, hipEventCreate(&e0));
  QR_CHECK(c, hipEventCreate(&e1));
  QR_CHECK(c, hipEventRecord(e0, c->stream));
  int rc = score_any(c, d_x, N, Fd, d_o);
  QR_CHECK(c, hipEventRecord(e1, c->stream));
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  if (!rc) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (kernel_ms) *kernel_ms = ms;
    QR_CHECK(c, hipMemcpy(out, d_o, N * 8, hipMemcpyDeviceToHost));
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dfree(d_x);
  dfree(d_o);
  return rc;
}

int qr_ensemble_partial_scores(qr_ctx *c, const float *x, size_t N, size_t F, int ignore_weights,
                               double *out) {
  if (!c || !x || !out || !N || !F) return QR_ERR_ARG;
  if (!c->d_ens) QR_FAIL(c, QR_ERR_STATE, "no ensemble uploaded");
  float *d_x = nullptr;
  double *d_o = nullptr;
  size_t Fd = F;
  if (c->ens_maxf >= 0 && F <= (size_t)c->ens_maxf) Fd = (size_t)c->ens_maxf + 1;
  const size_t T = c->ens_trees;
  // [N][T] f64 on the device, in slabs of at most 1 GB
  const size_t slab = std::max<size_t>(1, std::min(N, ((size_t)1 << 30) / (T * 8)));
  QR_CHECK(c, dalloc(&d_x, slab * Fd));
  QR_CHECK(c, dalloc(&d_o, slab * T));
  int rc = QR_OK;
  for (size_t d0 = 0; d0 < N && !rc; d0 += slab) {
    const size_t n = std::min(slab, N - d0);
    if (Fd == F) {
      QR_CHECK(c, hipMemcpy(d_x, x + d0 * F, n * F * 4, hipMemcpyHostToDevice));
    } else {
      const std::vector<float> padded = repad_rows(x + d0 * F, n, F, Fd);
      QR_CHECK(c, hipMemcpy(d_x, padded.data(), n * Fd * 4, hipMemcpyHostToDevice));
    }
    rc = qr_k_ensemble_score(c, d_x, n, Fd, nullptr, d_o, ignore_weights);
    if (rc) break;
    QR_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->ens_perm.empty()) {
      QR_CHECK(c, hipMemcpy(out + d0 * T, d_o, n * T * 8, hipMemcpyDeviceToHost));
    } else {  // (depth order: the walk's columns back to the model's)
      std::vector<double> tmp(n * T);
      QR_CHECK(c, hipMemcpy(tmp.data(), d_o, n * T * 8, hipMemcpyDeviceToHost));
      for (size_t d = 0; d < n; ++d)
        for (size_t t = 0; t < T; ++t) out[(d0 + d) * T + c->ens_perm[t]] = tmp[d * T + t];
    }
  }
  dfree(d_x);
  dfree(d_o);
  return rc;
}

int qr_oblivious_upload(qr_ctx *c, const uint32_t *feat, const float *thr,
                        const double *leaves, const float *weights,
                        const uint32_t *depths, size_t ntrees, size_t depth) {
  if (!c || !feat || !thr || !leaves || !weights || !ntrees || !depth || depth > 12)
    return QR_ERR_ARG;
  QR_CHECK(c, hipSetDevice(c->device));
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  dfree(c->d_obl_feat); dfree(c->d_obl_thr); dfree(c->d_obl_leaves); dfree(c->d_obl_w);
  dfree(c->d_obl_depths);
  const size_t nl = (size_t)1 << depth;
  QR_CHECK(c, dalloc(&c->d_obl_feat, ntrees * depth));
  QR_CHECK(c, dalloc(&c->d_obl_thr, ntrees * depth));
  QR_CHECK(c, dalloc(&c->d_obl_leaves, ntrees * nl));
  QR_CHECK(c, dalloc(&c->d_obl_w, ntrees));
  QR_CHECK(c, hipMemcpy(c->d_obl_feat, feat, ntrees * depth * 4, hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_obl_thr, thr, ntrees * depth * 4, hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_obl_leaves, leaves, ntrees * nl * 8, hipMemcpyHostToDevice));
  QR_CHECK(c, hipMemcpy(c->d_obl_w, weights, ntrees * 4, hipMemcpyHostToDevice));
  if (depths) {
    QR_CHECK(c, dalloc(&c->d_obl_depths, ntrees));
    QR_CHECK(c, hipMemcpy(c->d_obl_depths, depths, ntrees * 4, hipMemcpyHostToDevice));
  }
  c->obl_trees = ntrees;
  c->obl_depth = depth;
  // binned form: per feature the sorted distinct thresholds the ensemble tests; per
  // (tree, level) the feature and the index of its threshold among them
  c->ob_ready = false;
  c->obs_ready = false;
  dfree(c->d_ob_fk); dfree(c->d_ob_thr); dfree(c->d_ob_thr_cnt);
  uint32_t maxf = 0;
  for (size_t i = 0; i < ntrees * depth; ++i) maxf = std::max(maxf, feat[i]);
  c->obl_maxf = maxf;
  if (maxf < 65535) {
    const size_t F = (size_t)maxf + 1;
    std::vector<std::vector<float>> tv(F);
    for (size_t i = 0; i < ntrees * depth; ++i) tv[feat[i]].push_back(thr[i]);
    size_t tmax = 1;
    for (auto &v : tv) {
      std::sort(v.begin(), v.end());
      v.erase(std::unique(v.begin(), v.end()), v.end());
      tmax = std::max(tmax, v.size());
    }
    if (tmax <= 65535) {
      std::vector<uint32_t> fk(ntrees * depth), tc(F);
      std::vector<float> tt(F * tmax, 0.0f);
      for (size_t i = 0; i < ntrees * depth; ++i) {
        const auto &v = tv[feat[i]];
        const size_t k = std::lower_bound(v.begin(), v.end(), thr[i]) - v.begin();
        fk[i] = feat[i] | (uint32_t)(k << 16);
      }
      for (size_t f = 0; f < F; ++f) {
        tc[f] = (uint32_t)tv[f].size();
        std::copy(tv[f].begin(), tv[f].end(), tt.begin() + f * tmax);
      }
      QR_CHECK(c, dalloc(&c->d_ob_fk, fk.size()));
      QR_CHECK(c, dalloc(&c->d_ob_thr, tt.size()));
      QR_CHECK(c, dalloc(&c->d_ob_thr_cnt, tc.size()));
      QR_CHECK(c, hipMemcpy(c->d_ob_fk, fk.data(), fk.size() * 4, hipMemcpyHostToDevice));
      QR_CHECK(c, hipMemcpy(c->d_ob_thr, tt.data(), tt.size() * 4, hipMemcpyHostToDevice));
      QR_CHECK(c, hipMemcpy(c->d_ob_thr_cnt, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
      c->ob_F = F;
      c->ob_tmax = tmax;
      c->ob_u8 = tmax <= 255;
      c->ob_ready = true;
      // ---- k_obl_score_s: the level tests of a tree as sixteen dwords for the scalar unit
      c->obs_ready = false;
      dfree(c->d_obs_trees); dfree(c->d_obs_leaves);
      if (tmax <= 255 && depth <= 8 && (F - 1) * 64 <= 0xffff) {
        // a batch of leaf values next to eight (else four) 64-document blocks in half of a CU's
        // LDS; a thread carries four 16-byte pieces of the next batch (k_obl_score_s)
        const size_t doc_bytes = (F * 64 + 15) & ~(size_t)15, half = c->lds_cu / 2;  // (two workgroups per CU)
        size_t tb = 0, nw = 0;
        for (size_t w : {(size_t)8, (size_t)4}) {
          if (w * doc_bytes + 4 * nl * 8 > half) continue;
          const size_t cap = std::min(half - w * doc_bytes, 4 * w * 64 * 16);
          tb = std::min<size_t>(32, (cap / (nl * 8)) & ~(size_t)3);
          if (tb >= 4) {
            nw = w;
            break;
          }
        }
        if (nw) {
        const size_t tpad = (ntrees + tb - 1) / tb * tb;
        // a tree shallower than `depth` gets levels that are never true (bin > 255) at its END:
        // its leaf index comes out shifted left, and its leaf values are stored at the shifted places
        std::vector<uint32_t> tr(tpad * 16);
        std::vector<double> lw(tpad * nl, 0.0);
        for (size_t t = 0; t < tpad; ++t)
          for (size_t l = 0; l < 8; ++l) {
            tr[t * 16 + l] = 0;
            tr[t * 16 + 8 + l] = 255;
          }
        for (size_t t = 0; t < ntrees; ++t) {
          const size_t m = depths ? std::min<size_t>(depths[t], depth) : depth;
          for (size_t l = 0; l < m; ++l) {
            const uint32_t v = fk[t * depth + l];
            tr[t * 16 + l] = (v & 0xffffu) * 64u;
            tr[t * 16 + 8 + l] = v >> 16;
          }
          // generate_oblivious.cc:312-324: f32 weight promoted, one f64 product per (tree, leaf)
          for (size_t i = 0; i < ((size_t)1 << m); ++i)
            lw[t * nl + (i << (depth - m))] = (double)weights[t] * leaves[t * nl + i];
        }
        QR_CHECK(c, dalloc(&c->d_obs_trees, tr.size()));
        QR_CHECK(c, dalloc(&c->d_obs_leaves, lw.size()));
        QR_CHECK(c, hipMemcpy(c->d_obs_trees, tr.data(), tr.size() * 4, hipMemcpyHostToDevice));
        QR_CHECK(c, hipMemcpy(c->d_obs_leaves, lw.data(), lw.size() * 8, hipMemcpyHostToDevice));
        c->obs_tb = tb;
        c->obs_nw = nw;
        c->obs_tpad = tpad;
        c->obs_ready = true;
        }
      }
    }
  }
  return QR_OK;
}

int qr_oblivious_score(qr_ctx *c, const float *x, size_t N, size_t F, double *out,
                       float *kernel_ms) {
  if (!c || !x || !out || !N || !F) return QR_ERR_ARG;
  if (!c->d_obl_feat) QR_FAIL(c, QR_ERR_STATE, "no oblivious ensemble uploaded");
  float *d_x = nullptr;
  double *d_o = nullptr;
  const size_t Fin = F;
  if (F <= (size_t)c->obl_maxf) F = (size_t)c->obl_maxf + 1;  // absent features are zeros
  QR_CHECK(c, dalloc(&d_x, N * F));
  QR_CHECK(c, dalloc(&d_o, N));
  if (F == Fin) {
    QR_CHECK(c, hipMemcpy(d_x, x, N * F * 4, hipMemcpyHostToDevice));
  } else {
    const std::vector<float> padded = repad_rows(x, N, Fin, F);
    QR_CHECK(c, hipMemcpy(d_x, padded.data(), N * F * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  QR_CHECK(c, hipEventCreate(&e0));
  QR_CHECK(c, hipEventCreate(&e1));
  QR_CHECK(c, hipEventRecord(e0, c->stream));
  int rc = qr_k_obl_score_fast(c, d_x, N, F, d_o);  // binned, document-parallel
  if (rc < 0) rc = qr_k_obl_score(c, d_x, N, F, d_o);  // general fallback (f32 rows in LDS)
  QR_CHECK(c, hipEventRecord(e1, c->stream));
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  if (!rc) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (kernel_ms) *kernel_ms = ms;
    QR_CHECK(c, hipMemcpy(out, d_o, N * 8, hipMemcpyDeviceToHost));
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dfree(d_x);
  dfree(d_o);
  return rc;
}

// ---------------------------------------------------------------------------
int qr_prof_enable(qr_ctx *c, int on) {
  if (!c) return QR_ERR_ARG;
  c->prof_on = (on & 1) != 0;
  c->prof_child = (on & 2) != 0;
  c->prof_lambda = (on & 4) != 0 && !c->prof_child;
  // bits 8..15: events on every k-th root launch only (0 / 1 = every launch).  A launch that
  // carries a start and a stop event costs the stream ~7.5 us (scripts/ubench/launch_chain.hip),
  // which a caller timing whole iterations around the launches may not want on each of them
  c->prof_stride = (unsigned)((on >> 8) & 0xff);
  if (c->prof_stride == 0) c->prof_stride = 1;
  c->prof_tick = 0;
  return QR_OK;
}

static int prof_drain(qr_ctx *c) {
  if (c->prof_events.empty() && c->prof_events_child.empty()) return QR_OK;
  QR_CHECK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 2; ++k) {
    auto &ev = k ? c->prof_events_child : c->prof_events;
    for (auto &p : ev) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
        (k ? c->prof_ms_child : c->prof_ms) += ms;
        (k ? c->prof_launches_child : c->prof_launches)++;
      }
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
    ev.clear();
  }
  return QR_OK;
}

int qr_prof_reset(qr_ctx *c) {
  if (!c) return QR_ERR_ARG;
  int rc = prof_drain(c);
  c->prof_ms = c->prof_ms_child = 0;
  c->prof_launches = c->prof_launches_child = 0;
  return rc;
}

int qr_prof_get(qr_ctx *c, uint64_t *launches, double *total_ms, double *alg_bytes) {
  if (!c) return QR_ERR_ARG;
  int rc = prof_drain(c);
  if (rc) return rc;
  if (launches) *launches = c->prof_launches;
  if (total_ms) *total_ms = c->prof_ms;
  // n*F_loc (uint8 bins) + n*8 (f64 gradient, once) +
  // F_loc*256*16 (result); no sample ids at the root.
  if (alg_bytes)
    *alg_bytes = (double)c->N * c->flocal + (double)c->N * 8 + (double)c->flocal * 256 * 16;
  return QR_OK;
}

int qr_prof_get_child(qr_ctx *c, uint64_t *launches, double *total_ms) {
  if (!c) return QR_ERR_ARG;
  int rc = prof_drain(c);
  if (rc) return rc;
  if (launches) *launches = c->prof_launches_child;
  if (total_ms) *total_ms = c->prof_ms_child;
  return QR_OK;
}

}  // extern "C"
