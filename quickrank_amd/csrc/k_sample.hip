// k_sample.hip -- --subsample (mart.cc:287-329): every boosting iteration fits its
// tree on a fresh uniform sample (without replacement) of k training documents.
//
// The reference shuffles an id array with a clock-seeded engine and keeps its
// first k entries.  Here the sample of iteration t is the k documents with the
// smallest keys hash(seed, t, doc), ties by ascending document: a SELECTION, not a sort.
// The keys are a pure function of the document's index, so nothing about them is ever
// stored: a three-digit radix select (11 + 11 + 10 bits) finds the k-th smallest key T and
// the number r of documents with key == T that belong to the sample, each pass a launch that
// recomputes the hashes into an LDS histogram; then a count / scan / scatter over the context's own
// documents writes the flag array (the lambda kernel "cleans" its queries with it,
// lambdamart.cc:85-102) and the sample in ascending order into the document list the root
// node is built from.  The only memory traffic is what it writes: N flag bytes and 4k bytes
// of list (rounds 2-5 sorted (key, document) pairs with hipCUB: 16 N bytes of keys and ids
// through four radix passes, then a library compaction).  Same sample as the sort gave,
// reproducible, no host round trip.
#include "qr_internal.h"

namespace {

constexpr int SEL_B0 = 11, SEL_B1 = 11, SEL_B2 = 10;  // digits of a key, from the top
constexpr int SEL_CH = 2048;                           // documents per workgroup of the count / scatter passes

// what the select has found so far; zeroed with the histograms at the start of a draw
struct QrSelect {
  uint32_t prefix;     // the digits of T fixed so far
  uint32_t remaining;  // how many of the sample's documents lie at or beyond that prefix
  uint32_t T, r;       // the k-th smallest key; documents with key == T that are taken (the first r by index)
  uint32_t eq_before;  // document shards: documents of EARLIER ranks with key == T
  uint32_t pad[3];
};

// (document-sharded ranks: the key of a document is a function of its GLOBAL index, so every
// rank selects over the same keys and finds the same T and r, and marks its own part: no exchange)
__device__ __forceinline__ uint32_t sample_key(const unsigned long long seed, const uint32_t i, const uint32_t mask) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32) & mask;  // (mask: all ones but in the tests that want equal keys)
}

// one digit of the select: the histogram of digit (key >> SHIFT) over the documents whose higher
// digits equal the prefix found so far
template <int SHIFT, int BITS>
__global__ __launch_bounds__(256) void k_sel_hist(const uint32_t N, const unsigned long long seed, const uint32_t mask,
                                                  const QrSelect *__restrict__ sel, uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[1 << BITS];
  for (int b = threadIdx.x; b < (1 << BITS); b += 256) h[b] = 0;
  __syncthreads();
  uint32_t prefix = 0;
  if constexpr (SHIFT + BITS < 32) prefix = sel->prefix;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256) {
    const uint32_t key = sample_key(seed, (uint32_t)i, mask);
    bool in = true;
    if constexpr (SHIFT + BITS < 32) in = (key >> (SHIFT + BITS)) == prefix;
    if (in) atomicAdd(&h[(key >> SHIFT) & ((1u << BITS) - 1u)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < (1 << BITS); b += 256)
    if (h[b]) atomicAdd(&hist[b], h[b]);
}

// the digit in which the cumulative count reaches the number still wanted (one workgroup)
template <int BITS, bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void k_sel_pick(QrSelect *__restrict__ sel, const uint32_t *__restrict__ hist,
                                                  const uint32_t k) {
  constexpr int PER = (1 << BITS) / 256;
  __shared__ uint32_t part[256];
  const uint32_t rem = FIRST ? k : sel->remaining;
  uint32_t mine[PER], s = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    mine[j] = hist[threadIdx.x * PER + j];
    s += mine[j];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  const uint32_t incl = part[threadIdx.x];
  uint32_t below = incl - s;
  if (below < rem && rem <= incl) {  // (exactly one thread: 1 <= rem <= the documents under the prefix)
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (below + mine[j] >= rem) {
        const uint32_t digit = threadIdx.x * PER + j;
        const uint32_t prefix = FIRST ? digit : ((sel->prefix << BITS) | digit);
        sel->prefix = prefix;
        sel->remaining = rem - below;
        if (LAST) {
          sel->T = prefix;
          sel->r = rem - below;
        }
        break;
      }
      below += mine[j];
    }
  }
}

// document shards: the documents with key == T that lie before this rank's first one
__global__ __launch_bounds__(256) void k_sel_eq_before(const uint32_t first, const unsigned long long seed,
                                                       const uint32_t mask, QrSelect *__restrict__ sel) {
  __shared__ uint32_t n;
  if (threadIdx.x == 0) n = 0;
  __syncthreads();
  const uint32_t T = sel->T;
  uint32_t mine = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < first; i += (size_t)gridDim.x * 256)
    mine += sample_key(seed, (uint32_t)i, mask) == T;
  if (mine) atomicAdd(&n, mine);
  __syncthreads();
  if (threadIdx.x == 0 && n) atomicAdd(&sel->eq_before, n);
}

// (documents with a smaller key, documents with key == T) packed into one word: low / high half
__device__ __forceinline__ unsigned long long sel_block_scan(unsigned long long *buf, const unsigned long long v,
                                                             const int n) {
  buf[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < n; off <<= 1) {
    const unsigned long long t = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : 0ull;
    __syncthreads();
    buf[threadIdx.x] += t;
    __syncthreads();
  }
  return buf[threadIdx.x];  // inclusive
}

// the context's own documents [first, first + Nloc): per workgroup of SEL_CH documents, how many
// have a key below T and how many equal to it
__global__ __launch_bounds__(256) void k_sel_count(const uint32_t Nloc, const uint32_t first,
                                                   const unsigned long long seed, const uint32_t mask,
                                                   const QrSelect *__restrict__ sel,
                                                   unsigned long long *__restrict__ wgcnt) {
  __shared__ unsigned long long buf[256];
  const uint32_t T = sel->T;
  const uint32_t base = blockIdx.x * SEL_CH + threadIdx.x * 8;
  unsigned long long mine = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (base + j < Nloc) {
      const uint32_t key = sample_key(seed, first + base + j, mask);
      mine += (unsigned long long)(key < T) + ((unsigned long long)(key == T) << 32);
    }
  }
  const unsigned long long incl = sel_block_scan(buf, mine, 256);
  if (threadIdx.x == 255) wgcnt[blockIdx.x] = incl;
}

// exclusive prefix of the workgroups' counts (one workgroup), and the size of the rank's part of the sample
__global__ __launch_bounds__(1024) void k_sel_scan(const uint32_t nwg, const unsigned long long *__restrict__ wgcnt,
                                                   unsigned long long *__restrict__ wgoff,
                                                   const QrSelect *__restrict__ sel, uint32_t *__restrict__ count_out) {
  __shared__ unsigned long long buf[1024];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nwg; b0 += 1024) {
    const unsigned long long v = b0 + threadIdx.x < nwg ? wgcnt[b0 + threadIdx.x] : 0ull;
    const unsigned long long incl = sel_block_scan(buf, v, 1024);
    if (b0 + threadIdx.x < nwg) wgoff[b0 + threadIdx.x] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t L = (uint32_t)carry, E = (uint32_t)(carry >> 32), r = sel->r, eb = sel->eq_before;
    *count_out = L + (min(eb + E, r) - min(eb, r));
  }
}

// flags and the ascending list: a document belongs to the sample if its key is below T, or equal
// to T and fewer than r such documents precede it (earlier ranks' included)
__global__ __launch_bounds__(256) void k_sel_scatter(const uint32_t Nloc, const uint32_t first,
                                                     const unsigned long long seed, const uint32_t mask,
                                                     const QrSelect *__restrict__ sel,
                                                     const unsigned long long *__restrict__ wgoff,
                                                     uint8_t *__restrict__ present, uint32_t *__restrict__ list) {
  __shared__ unsigned long long buf[256];
  const uint32_t T = sel->T, r = sel->r, eb = sel->eq_before;
  const uint32_t base = blockIdx.x * SEL_CH + threadIdx.x * 8;
  uint32_t less = 0, eq = 0;  // bit j: document base + j
  unsigned long long mine = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (base + j < Nloc) {
      const uint32_t key = sample_key(seed, first + base + j, mask);
      less |= (uint32_t)(key < T) << j;
      eq |= (uint32_t)(key == T) << j;
    }
  }
  mine = (unsigned long long)__popc(less) + ((unsigned long long)__popc(eq) << 32);
  const unsigned long long off = wgoff[blockIdx.x] + sel_block_scan(buf, mine, 256) - mine;
  uint32_t L = (uint32_t)off, E = (uint32_t)(off >> 32);
  const uint32_t taken0 = min(eb, r);
  unsigned long long flags = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool lt = (less >> j) & 1u, e = (eq >> j) & 1u;
    const bool in = lt || (e && eb + E < r);
    if (in) {
      list[L + (min(eb + E, r) - taken0)] = base + j;
      flags |= 1ull << (8 * j);
    }
    L += lt;
    E += e;
  }
  if (base + 8 <= Nloc) {
    *reinterpret_cast<unsigned long long *>(present + base) = flags;  // (hipMalloc'd array, base a multiple of 8)
  } else {
    for (int j = 0; j < 8 && base + j < Nloc; ++j) present[base + j] = (uint8_t)((flags >> (8 * j)) & 1u);
  }
}

// per-slice (sum of squares, sum) of the pseudo-responses over the sample: the
// root node's statistics (rtnode_histogram.cc:199-203 runs over the sample ids)
__global__ __launch_bounds__(256) void k_sample_sums(const uint32_t *__restrict__ list,
                                                     const uint32_t k,
                                                     const double *__restrict__ lambda,
                                                     double *__restrict__ ssq) {
  __shared__ double a[256], b[256];
  const uint32_t base = blockIdx.x * QR_SLICE;
  double sq = 0.0, sm = 0.0;
  for (uint32_t j = base + threadIdx.x; j < base + QR_SLICE && j < k; j += 256) {
    const double l = lambda[list[j]];
    sq += l * l;
    sm += l;
  }
  a[threadIdx.x] = sq;
  b[threadIdx.x] = sm;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      a[threadIdx.x] += a[threadIdx.x + off];
      b[threadIdx.x] += b[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ssq[2 * blockIdx.x] = a[0];
    ssq[2 * blockIdx.x + 1] = b[0];
  }
}

}  // namespace

// device words of a draw's working set: three histograms, the select's state, the workgroups' counts and offsets
size_t qr_k_sample_work_bytes(size_t Nloc) {
  const size_t nwg = (Nloc + SEL_CH - 1) / SEL_CH;
  return ((size_t)(1 << SEL_B0) + (1 << SEL_B1) + (1 << SEL_B2)) * 4 + sizeof(QrSelect) + 2 * (nwg ? nwg : 1) * 8;
}

// draws the sample of the next iteration: c->d_present, and the ascending list in
// c->d_order[0][0 .. sub_n)
int qr_k_sample_draw(qr_ctx *c) {
  // N: the documents the sample is drawn from (all ranks' on a document-sharded context,
  // whose own documents are [sub_first, sub_first + Nloc) of them)
  const uint32_t Nloc = (uint32_t)c->N;
  const uint32_t N = c->dmode ? (uint32_t)c->Nglobal : Nloc;
  const uint32_t first = c->dmode ? (uint32_t)c->sub_first : 0u;
  c->sub_iter += 0x9E3779B97F4A7C15ull;
  const unsigned long long seed = (unsigned long long)(c->sub_seed + c->sub_iter);
  const uint32_t mask = c->sub_key_mask, k = (uint32_t)c->sub_k;
  uint32_t *h0 = reinterpret_cast<uint32_t *>(c->d_sample_work), *h1 = h0 + (1 << SEL_B0), *h2 = h1 + (1 << SEL_B1);
  QrSelect *sel = reinterpret_cast<QrSelect *>(h2 + (1 << SEL_B2));
  const uint32_t nwg = (Nloc + SEL_CH - 1) / SEL_CH;
  unsigned long long *wgcnt = reinterpret_cast<unsigned long long *>(sel + 1), *wgoff = wgcnt + (nwg ? nwg : 1);
  QR_CHECK(c, hipMemsetAsync(h0, 0, (size_t)((char *)(sel + 1) - (char *)h0), c->stream));
  const unsigned hgrid = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, ((size_t)N + 4095) / 4096));
  hipLaunchKernelGGL((k_sel_hist<SEL_B1 + SEL_B2, SEL_B0>), dim3(hgrid), dim3(256), 0, c->stream, N, seed, mask, sel, h0);
  hipLaunchKernelGGL((k_sel_pick<SEL_B0, true, false>), dim3(1), dim3(256), 0, c->stream, sel, h0, k);
  hipLaunchKernelGGL((k_sel_hist<SEL_B2, SEL_B1>), dim3(hgrid), dim3(256), 0, c->stream, N, seed, mask, sel, h1);
  hipLaunchKernelGGL((k_sel_pick<SEL_B1, false, false>), dim3(1), dim3(256), 0, c->stream, sel, h1, k);
  hipLaunchKernelGGL((k_sel_hist<0, SEL_B2>), dim3(hgrid), dim3(256), 0, c->stream, N, seed, mask, sel, h2);
  hipLaunchKernelGGL((k_sel_pick<SEL_B2, false, true>), dim3(1), dim3(256), 0, c->stream, sel, h2, k);
  if (first)
    hipLaunchKernelGGL(k_sel_eq_before, dim3((unsigned)std::min<size_t>(1024, ((size_t)first + 4095) / 4096)), dim3(256), 0,
                       c->stream, first, seed, mask, sel);
  if (nwg)
    hipLaunchKernelGGL(k_sel_count, dim3(nwg), dim3(256), 0, c->stream, Nloc, first, seed, mask, sel, wgcnt);
  hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(1024), 0, c->stream, nwg, wgcnt, wgoff, sel, c->d_sample_count);
  if (nwg)
    hipLaunchKernelGGL(k_sel_scatter, dim3(nwg), dim3(256), 0, c->stream, Nloc, first, seed, mask, sel, wgoff,
                       c->d_present, c->d_order[0]);
  QR_CHECK(c, hipGetLastError());
  c->sub_n = c->sub_k;
  if (c->dmode) {  // how many of the sample's documents are this rank's: the host sizes launches by it
    uint32_t cnt = 0;
    QR_D2H(c, &cnt, c->d_sample_count, 4);
    c->sub_n = cnt;
  }
  return QR_OK;
}

int qr_k_sample_sums(qr_ctx *c) {
  const unsigned grid = (unsigned)std::max<size_t>(1, (c->sub_n + QR_SLICE - 1) / QR_SLICE);
  hipLaunchKernelGGL(k_sample_sums, dim3(grid), dim3(256), 0, c->stream, c->d_order[0],
                     (uint32_t)c->sub_n, c->d_lambda, c->d_ssq);
  QR_CHECK(c, hipGetLastError());
  return QR_OK;
}
