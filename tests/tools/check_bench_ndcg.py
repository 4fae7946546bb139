"""TEST TOOL (CPU, ~25 s): the NDCG@10 a bench.py line reports after its warmup + steps lambda
passes against the oracle trained on the same synthetic set -- an independent check that the
timed region did the whole job at the full size.
    python tests/tools/check_bench_ndcg.py profiles/r02_j_bench.json"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from bench import synth
oracle.build(ref=False)
d = json.load(open(sys.argv[1]))
passes = d["warmup"] + d["steps"]          # the last pass ranks the scores after passes - 1 trees
x, labels, qoff = synth(10000, 100, 136)
om = oracle.train(x, labels, qoff, algo="LAMBDAMART", ntrees=passes, shrinkage=0.1, nthresholds=255,
                  nleaves=10, minls=1, esr=0)
want = float(np.asarray(om["train_metric"])[passes - 2])
got = d["config"]["ndcg10_last"]
print(f"bench ndcg10_last {got!r}  oracle after {passes - 1} trees {want!r}  relative difference {abs(got - want) / want:.2e}")
sys.exit(0 if abs(got - want) <= 1e-12 * want else 1)
