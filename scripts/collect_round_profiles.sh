#!/bin/bash
# Everything the round's numbers come from, on ONE box with the committed code (GPU box; run through
# gpurun): the full bench line, the kernel trace of the headline workload and of the 8M set, the
# PMC passes of the histogram launches (FETCH_SIZE and WRITE_SIZE in separate runs, as
# MI355X_MICROARCH.md prescribes) and of the lambda pass (SQ instruction counters), the wide-path
# timings (--num-thresholds 1024 / 4096 / 0) with the kernel trace of the pre-sorted path.
# Results under gpurun_out/<tag>/; the summaries are copied into profiles/ by hand.
#   bash scripts/collect_round_profiles.sh r04z
set -x
TAG=${1:-r04z}
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
B="python bench.py --no-extras --no-cpu-baseline --no-scoring"
rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- $B --steps 20 --warmup 5 > $O/bench_traced.json 2> $O/prof.err
rocprofv3 --kernel-trace --stats -d $O/prof8m -o bench --output-format csv -- $B --queries 80000 --steps 12 --warmup 3 > $O/bench8m_traced.json 2> $O/prof8m.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- $B --steps 6 --warmup 2 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- $B --steps 6 --warmup 2 > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O/pmc_sq -o pmc --output-format csv -- $B --steps 20 --warmup 5 > /dev/null 2> $O/pmc_sq.err
python scripts/prof_summary.py $O/prof bench > $O/summary.md 2>&1
python scripts/prof_summary.py $O/prof8m bench > $O/summary8m.md 2>&1
python scripts/child_classes.py $O/prof > $O/classes1m.txt 2>&1
python scripts/child_classes.py $O/prof8m > $O/classes8m.txt 2>&1
python scripts/doc_batch_bench.py 1 60 2>&1 | grep "per iteration" > $O/doc_protocol.txt
python scripts/pmc_tables.py hist $O/pmc_fetch $O/pmc_write $O/pmc_hist.json > /dev/null 2> $O/pmc_hist.err
python scripts/pmc_tables.py lambda $O/pmc_sq $O/lambda_pmc.json > /dev/null 2> $O/pmc_lambda.err
WB_ITERS=13 WB_NTHR=255,1024,4096,0 python scripts/wide_bench.py > $O/wide_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_wide -o w --output-format csv -- env WB_NTHR=0 python scripts/wide_bench.py > /dev/null 2> $O/prof_wide.err
python scripts/prof_summary.py $O/prof_wide w 2>&1 | grep -v rocprim | cut -c1-130 > $O/wide_summary.md
cp $O/prof/bench_kernel_stats.csv $O/bench_kernel_stats.csv
cp $O/prof8m/bench_kernel_stats.csv $O/8M_kernel_stats.csv
rm -rf $O/prof $O/prof8m $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/prof_wide
ls -la $O
