"""CPU baseline beside the wide path's number (TEST TOOL): the oracle -- the C +
OpenMP restatement of the reference's loop -- trained with the reference's DEFAULT
`--num-thresholds 0` (every distinct value a threshold, mart.cc:155-158) on the same MSLR-shaped
stand-in `scripts/wide_bench.py` times the device on, all host cores, a bounded number of
iterations.  Run on the GPU box so that both numbers come from one machine:
    python tests/tools/wide_cpu_baseline.py [iterations=3] [nthresholds=0]
(the oracle lives under oracle/: only tests/ may call it, so this is not part of scripts/)."""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    nthr = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import oracle
    from datagen import make_mslr_like
    oracle.build(ref=False)
    x, labels, qoff = make_mslr_like()
    cores = oracle.available_cores()
    t0 = time.perf_counter()
    m = oracle.train(x, labels, qoff, algo="LAMBDAMART", ntrees=iters, shrinkage=0.1, nthresholds=nthr,
                     nleaves=10, minls=1, esr=0, threads=cores)
    wall = time.perf_counter() - t0
    it = m["iter_seconds"]
    sec = float(np.mean(it[1:])) if len(it) > 1 else float(it[0])
    out = {"workload": f"MSLR-shaped stand-in, {len(labels)} docs x {x.shape[1]} features, LambdaMART 10 leaves, "
                       f"--num-thresholds {nthr}", "kind": "port", "cores": cores,
           "slots_max": int(m["thr_size"].max()), "slots_total": int(m["thr_size"].sum()),
           "ms_per_iteration": sec * 1e3, "iterations": iters, "init_s": wall - float(np.sum(it)),
           "ndcg10_after": float(m["train_metric"][-1])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
