"""Seeded synthetic learning-to-rank data (MSLR-shaped), shared by tests and bench.

Features i.i.d. U[0,1) f32 (optionally some columns quantised to few integer
levels, a constant column and a duplicated column -- SURVEY.md section 8c's
adversarial cases); labels in {0..4} driven by the first four features.
"""
import numpy as np


def make_queries(nq, docs_per_query, seed=0, ragged=False, sizes=None):
    rng = np.random.default_rng(seed + 1000)
    if sizes is not None:
        sizes = np.asarray(sizes)
        nq = len(sizes)
    elif ragged:
        sizes = rng.integers(1, 2 * docs_per_query, size=nq)
    else:
        sizes = np.full(nq, docs_per_query)
    qoff = np.zeros(nq + 1, np.uint64)
    qoff[1:] = np.cumsum(sizes)
    return qoff


def make_dataset(nq=40, docs_per_query=30, F=16, seed=0, ragged=False,
                 adversarial=False, sizes=None):
    qoff = make_queries(nq, docs_per_query, seed, ragged, sizes)
    nq = len(qoff) - 1
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
    """Stand-in for MSLR-WEB10K fold 1 (BASELINE.json configs[0]): the generator lives in
    bench.py (`synth_mslr`), which reports the same set as `mslr_shaped`."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synth_mslr
    return synth_mslr(nq=nq, mean_q=mean_q, F=F, sparse_cols=sparse_cols, seed=seed)
