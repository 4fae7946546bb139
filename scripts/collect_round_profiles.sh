#!/bin/bash
# Everything the round's numbers come from, on ONE box with the committed code (GPU box; run through
# gpurun): the full bench line, the kernel trace of the headline workload and of the 8M set, the
# PMC passes of the histogram launches (FETCH_SIZE and WRITE_SIZE in separate runs, as
# MI355X_MICROARCH.md prescribes), the MSLR-shaped stand-in.  Results under gpurun_out/<tag>/;
# the summaries are copied into profiles/ by hand (scripts/prof_summary.py, pmc_summary.py).
#   bash scripts/collect_round_profiles.sh r03z
set -x
TAG=${1:-r03z}
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
B="python bench.py --no-extras --no-cpu-baseline --no-scoring"
rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- $B --steps 20 --warmup 5 > $O/bench_traced.json 2> $O/prof.err
rocprofv3 --kernel-trace --stats -d $O/prof8m -o bench --output-format csv -- $B --queries 80000 --steps 12 --warmup 3 > $O/bench8m_traced.json 2> $O/prof8m.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- $B --steps 3 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- $B --steps 3 --warmup 1 > /dev/null 2> $O/pmc_write.err
python scripts/prof_summary.py $O/prof bench > $O/summary.md 2>&1
python scripts/prof_summary.py $O/prof8m bench > $O/summary8m.md 2>&1
python scripts/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch.txt 2>&1
python scripts/pmc_summary.py $O/pmc_write > $O/pmc_write.txt 2>&1
bash scripts/wide_prof.sh 1024 > $O/wide_prof_1024.txt 2>&1
python scripts/wide_bench.py > $O/wide_bench.txt 2>&1
ls -la $O
