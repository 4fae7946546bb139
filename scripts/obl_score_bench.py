"""Oblivious-ensemble scoring throughput (generate_oblivious.cc semantics), config 4 shape."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from quickrank_amd._capi import Context
T, D, N, F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 6, 1000000, 136
rng = np.random.default_rng(3)
feat = rng.integers(0, F, (T, D)).astype(np.uint32)
thr = rng.random((T, D), dtype=np.float32)
leaves = rng.standard_normal((T, 1 << D))
w = np.full(T, 0.1, np.float32)
x = rng.random((N, F), dtype=np.float32)
c = Context(0)
c.upload_oblivious(feat, thr, leaves, w)
s, ms = c.score_oblivious(x)
s, ms = c.score_oblivious(x)
print(f"oblivious {T} trees depth {D} x {N} docs x {F} features: {ms:.2f} ms, {N / ms * 1e3:.3e} docs/s, "
      f"{N * T * D / ms * 1e3:.3e} level tests/s")
# reference semantics on a sample
idx = np.zeros((2000, T), np.int64)
for l in range(D):
    idx |= (x[:2000][:, feat[:, l]] > thr[:, l]).astype(np.int64) << (D - 1 - l)
want = np.zeros(2000)
for t in range(T):
    want = want + np.float64(w[t]) * leaves[t, idx[:, t]]
print("sample bit-exact:", bool(np.array_equal(s[:2000], want)))
