#!/bin/bash
# Round 6, VERDICT r5 item 2: what bounds k_hist_batch.  bash scripts/r06_child.sh [ab|pmc|all]
#   ab   pipeline depth of the child launches (QR_HIST_SETS_CHILD 3 = product, 4, 5, 6; libraries built
#        beforehand: QR_HIP_LIB=.../libqr_sN.so QR_HIP_EXTRA_FLAGS=-DQR_HIST_SETS_CHILD=N python -m quickrank_amd.build)
#   pmc  the counter passes of scripts/pmc_child.sh at 8M and 1M
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
W=${1:-all}
O=gpurun_out/r06_child
mkdir -p $O
if [ "$W" = ab ] || [ "$W" = all ]; then
  for L in hip s4 s5 s6 pk pk4; do
    [ -f quickrank_amd/lib/libqr_$L.so ] || continue
    QR_HIP_LIB=$PWD/quickrank_amd/lib/libqr_$L.so timeout 600 python bench.py --no-cpu-baseline --no-scoring --huge-blocks 0 \
      --steps 40 --warmup 5 > $O/ab_$L.json 2> $O/ab_$L.err
  done
  python - <<'PY' > $O/ab_table.md
import json, glob, os
print("| library | 1M ms/step | root us (1M) | root frac | child us/tree (1M) | child frac (1M) | strong_8M ms | 8M root frac | MSLR-shaped ms |")
print("|---|---|---|---|---|---|---|---|---|")
for L in ("hip", "s4", "s5", "s6", "pk", "pk4"):
    p = f"gpurun_out/r06_child/ab_{L}.json"
    if not os.path.exists(p): continue
    try:
        j = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print(f"| {L} | failed: {e} |"); continue
    c = j.get("roofline_child_hist", {})
    rf = j.get("roofline", {})
    print(f"| {L} | {j['ms_per_step']:.4f} | {rf.get('avg_launch_us')} | {rf.get('frac')} | {c.get('us_per_tree')} | {c.get('frac')} | "
          f"{j.get('strong_8M', {}).get('ms_per_step')} | {(j.get('strong_8M', {}).get('roofline') or {}).get('frac')} | {(j.get('mslr_shaped') or {}).get('ms_per_step')} |")
PY
  cat $O/ab_table.md
fi
if [ "$W" = gather ] || [ "$W" = all ]; then
  # the memory side alone: the child launches' requests without the histogram (scripts/ubench/gather_lines.hip)
  timeout 300 scripts/ubench/gather_lines 8000000 > $O/gather_8M.md 2>&1
  timeout 300 scripts/ubench/gather_lines 1000000 > $O/gather_1M.md 2>&1
  head -70 $O/gather_8M.md
fi
if [ "$W" = pmc ] || [ "$W" = all ]; then
  bash scripts/pmc_child.sh r06pmc8 80000 > $O/pmc8.out 2>&1
  bash scripts/pmc_child.sh r06pmc1 10000 > $O/pmc1.out 2>&1
  tail -40 gpurun_out/r06pmc8/table.md gpurun_out/r06pmc1/table.md
fi
