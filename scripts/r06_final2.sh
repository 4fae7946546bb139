#!/bin/bash
# Round 6, the last GPU call: the new tests first (select-based --subsample, equal keys over document
# shards, the moved bin-map rebuild), then the full suite and smoke on the library as committed, then what a
# draw of the sample costs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06z2
mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_docshard.py tests/test_gpu_bins_verify.py -m gpu -q -k "subsample or bins_verify or verify" > $O/pytest_new.txt 2>&1
tail -4 $O/pytest_new.txt
timeout 560 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt
timeout 100 python - > $O/sample_draw_cost.txt 2>&1 <<'PY'
import time, numpy as np
import quickrank_amd as qr
for N in (1_000_000, 8_000_000):
    rng = np.random.default_rng(0)
    F = 8
    x = rng.random((N, F), dtype=np.float32)
    labels = rng.integers(0, 5, N).astype(np.float32)
    qoff = np.arange(0, N + 1, 100, dtype=np.uint64)
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    c.build_bins(16)
    def cost(reps=40):
        c.compute_residuals(); c.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            c.compute_residuals()
        c.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    off = cost()
    c.set_subsample(0.5, seed=1)
    on = cost()
    print(f"N={N}: qr_residual_compute {off:.1f} us without a sample, {on:.1f} us with the draw of k = N/2 -> the draw (select + flags + list + sample sums) ~{on - off:.1f} us", flush=True)
    c.close()
PY
cat $O/sample_draw_cost.txt | tail -3
