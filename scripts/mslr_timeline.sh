#!/bin/bash
# kernel timeline (start, duration, gap to the previous end) of one boosting iteration on the
# MSLR-shaped stand-in (QR_TL_BENCH=1: the bench workload), 255 thresholds:
#   bash scripts/mslr_timeline.sh [iteration]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/mt.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import torch; torch.cuda.init()
from quickrank_amd._capi import Context
if os.environ.get("QR_TL_BENCH"):  # the bench workload instead (1M x 136, 10000 queries of 100)
    from bench import synth
    x, l, q = synth(int(os.environ.get("QR_TL_QUERIES", "10000")), 100, 136)
else:
    from datagen import make_mslr_like
    x, l, q = make_mslr_like()
c = Context(0); c.upload(x, l, q); c.build_bins(255); c.reset_scores()
for it in range(int(os.environ.get("QR_TL_ITERS", "40"))):
    c.compute_lambdas("NDCG", 10)
    if it: c.tree_nodes()  # (the previous tree's records, read under this iteration's lambda pass: bench.py's order)
    c.fit_tree(10, 1, True, read=False); c.update_scores(0.1); c.metric_last()
c.synchronize()
PY
rm -rf gpurun_out/mslr_tl
rocprofv3 --kernel-trace -d gpurun_out/mslr_tl -o t --output-format csv -- python /tmp/mt.py > /dev/null 2>&1
python - "${1:-30}" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/mslr_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# iterations start at the k_lambda launches that follow a k_prep-less gap: split on the first k_lambda after a non-lambda kernel
its, prev = [], ""
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if "k_lambda" in n and "k_lambda" not in prev: its.append(i)
    prev = n
k = int(sys.argv[1]); a, b = its[k], its[k + 1]
t0 = int(rows[a]["Start_Timestamp"]); last = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - last) / 1e3:6.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:70]}")
    last = max(last, e)
print(f"iteration {k}: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
PY
