"""Seeded synthetic learning-to-rank data (MSLR-shaped), shared by tests and bench.

Features i.i.d. U[0,1) f32 (optionally some columns quantised to few integer
levels, a constant column and a duplicated column -- SURVEY.md section 8c's
adversarial cases); labels in {0..4} driven by the first four features.
"""
import numpy as np


def make_queries(nq, docs_per_query, seed=0, ragged=False):
    rng = np.random.default_rng(seed + 1000)
    if ragged:
        sizes = rng.integers(1, 2 * docs_per_query, size=nq)
    else:
        sizes = np.full(nq, docs_per_query)
    qoff = np.zeros(nq + 1, np.uint64)
    qoff[1:] = np.cumsum(sizes)
    return qoff


def make_dataset(nq=40, docs_per_query=30, F=16, seed=0, ragged=False,
                 adversarial=False):
    qoff = make_queries(nq, docs_per_query, seed, ragged)
    N = int(qoff[-1])
    rng = np.random.default_rng(seed)
    x = rng.random((N, F), dtype=np.float32)
    if adversarial and F >= 8:
        x[:, 4] = np.floor(x[:, 4] * 7)          # <= 7 integer levels
        x[:, 5] = 0.5                            # constant column
        x[:, 6] = x[:, 2]                        # duplicate column
        x[:, 7] = np.floor(x[:, 7] * 300) / 300  # > 255 uniques but discrete
    s = x[:, :4].sum(axis=1, dtype=np.float64)
    labels = np.minimum(4, np.floor(1.25 * s)).astype(np.float32)
    if adversarial:
        # a few all-zero-label queries
        for q in range(0, nq, 7):
            labels[int(qoff[q]):int(qoff[q + 1])] = 0
    return x, labels, qoff


def make_mslr_like(nq=6000, mean_q=120, F=136, sparse_cols=40, seed=7):
    """Stand-in for MSLR-WEB10K fold 1 (BASELINE.json configs[0]; the files are not
    in the image): ~nq * mean_q documents in ragged queries (log-normal sizes, mean
    ~mean_q, clipped to [1, 1200]); the last `sparse_cols` columns are sparse count
    features (90 % zeros, else one of 32 integer levels: the "uniques <= nthresholds"
    threshold branch with one very hot bin per column), the others real-valued
    U[0,1); labels 0..4 with MSLR's skew P = .52/.32/.13/.02/.01, driven by four
    real columns, two count columns and noise."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.round(rng.lognormal(np.log(mean_q) - 0.18, 0.6, nq)), 1, 1200).astype(np.int64)
    qoff = np.zeros(nq + 1, np.uint64)
    qoff[1:] = np.cumsum(sizes)
    N = int(qoff[-1])
    x = rng.random((N, F), dtype=np.float32)
    if sparse_cols:
        c0 = F - sparse_cols
        lv = np.floor(x[:, c0:] * 32).astype(np.float32) + 1
        lv[rng.random((N, sparse_cols), dtype=np.float32) < 0.9] = 0
        x[:, c0:] = lv
    rel = (0.5 * x[:, 0] + 0.4 * x[:, 1] + 0.3 * x[:, 2] + 0.2 * x[:, 3]).astype(np.float64)
    if sparse_cols:
        rel += 0.15 * (x[:, F - 1] > 0) + 0.1 * (x[:, F - 2] > 0)
    rel += 0.3 * rng.standard_normal(N)
    cuts = np.quantile(rel, [0.52, 0.84, 0.97, 0.99])
    labels = np.searchsorted(cuts, rel).astype(np.float32)
    return x, labels, qoff
