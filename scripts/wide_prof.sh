#!/bin/bash
# kernel breakdown of the wide path (1024 thresholds) on the MSLR-shaped stand-in
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/wp.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import torch; torch.cuda.init()
from datagen import make_mslr_like
from quickrank_amd._capi import Context
x, l, q = make_mslr_like()
c = Context(0); c.upload(x, l, q); c.build_bins(int(sys.argv[1])); c.reset_scores()
for it in range(6):
    c.compute_lambdas("NDCG", 10); c.fit_tree(10, 1, True, read=False); c.update_scores(0.1); c.metric_last(); c.tree_nodes()
c.synchronize()
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/wide_prof -o w --output-format csv -- python /tmp/wp.py ${1:-1024} > /dev/null 2>&1
python scripts/prof_summary.py gpurun_out/wide_prof w | head -30
