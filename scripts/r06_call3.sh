#!/bin/bash
# Round 6, third GPU call (every step under its own timeout).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06_repro
mkdir -p $O gpurun_out/r06_child
T0=$(date +%s)
stamp() { echo "$1: $(( $(date +%s) - T0 )) s" >> gpurun_out/r06_call3_times.txt; }
timeout 300 python tests/tools/repro_first_tree.py 1 2 --procs 8 --iters 150 > $O/s1cfg2_p8.txt 2>&1; stamp repro_s1cfg2
timeout 300 python tests/tools/repro_first_tree.py 1 2 --procs 4 --iters 60 --jitter > $O/s1cfg2_jitter.txt 2>&1; stamp repro_s1cfg2_jitter
timeout 300 python tests/tools/repro_first_tree.py 0 248 --procs 4 --iters 60 --jitter > $O/s0cfg248_jitter.txt 2>&1; stamp repro_s0cfg248_jitter
OMP_NUM_THREADS=8 timeout 500 python tests/tools/abort_hunt.py 4 --jitter --no-torch --parallel 2 > $O/sweep_jitter.txt 2>&1; stamp sweep_jitter
tail -4 $O/s1cfg2_p8.txt $O/s1cfg2_jitter.txt $O/s0cfg248_jitter.txt $O/sweep_jitter.txt
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r06_pytest_b.txt 2>&1; stamp pytest
tail -12 gpurun_out/r06_pytest_b.txt
B="python bench.py --no-cpu-baseline --no-scoring --huge-blocks 0 --steps 40 --warmup 5"
timeout 300 $B > gpurun_out/r06_child/blocks_equal.json 2> /dev/null; stamp bench_equal
QR_BLOCKS_WIDE=1 timeout 300 $B > gpurun_out/r06_child/blocks_wide.json 2> /dev/null; stamp bench_wide
python - <<'PY'
import json
for L in ("equal", "wide"):
    try:
        j = json.loads(open(f"gpurun_out/r06_child/blocks_{L}.json").read().strip().splitlines()[-1])
        c, rf, s8 = j.get("roofline_child_hist", {}), j.get("roofline", {}), j.get("strong_8M", {})
        print(L, "1M ms", j["ms_per_step"], "root us", rf.get("avg_launch_us"), "frac", rf.get("frac"), "child us/tree", c.get("us_per_tree"),
              "child frac", c.get("frac"), "8M ms", s8.get("ms_per_step"), "8M root frac", (s8.get("roofline") or {}).get("frac"),
              "mslr", (j.get("mslr_shaped") or {}).get("ms_per_step"))
    except Exception as e:
        print(L, "failed", e)
PY
PMC_TIMEOUT=200 timeout 1100 bash scripts/pmc_child.sh r06pmc1 10000 > gpurun_out/r06_child/pmc1.out 2>&1; stamp pmc1
tail -14 gpurun_out/r06pmc1/table.md
cat gpurun_out/r06_call3_times.txt
