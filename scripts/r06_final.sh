#!/bin/bash
# Round 6, last GPU call: the full suite, the full bench line, kernel traces (1M, 8M, scoring at config 5's
# full size), the histogram launches' FETCH / WRITE passes and the lambda pass's SQ counters, the scoring
# kernels' SQ / LDS counters -- everything under its own timeout.  Summaries are copied into profiles/ by hand.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06z
mkdir -p $O
T0=$(date +%s); stamp() { echo "$1: $(( $(date +%s) - T0 )) s" >> $O/times.txt; }
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; stamp pytest
tail -5 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; stamp smoke
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; stamp bench
B="python bench.py --no-extras --no-cpu-baseline --no-scoring"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- $B --steps 20 --warmup 5 > $O/bench_traced.json 2> $O/prof.err; stamp trace1M
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof8m -o bench --output-format csv -- $B --queries 80000 --steps 12 --warmup 3 > $O/bench8m_traced.json 2> $O/prof8m.err; stamp trace8M
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- $B --steps 6 --warmup 2 > /dev/null 2> $O/pmc_fetch.err; stamp pmc_fetch
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- $B --steps 6 --warmup 2 > /dev/null 2> $O/pmc_write.err; stamp pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O/pmc_sq -o pmc --output-format csv -- $B --steps 20 --warmup 5 > /dev/null 2> $O/pmc_sq.err; stamp pmc_sq
python scripts/prof_summary.py $O/prof bench > $O/summary.md 2>&1
python scripts/prof_summary.py $O/prof8m bench > $O/summary8m.md 2>&1
python scripts/child_classes.py $O/prof > $O/classes1m.txt 2>&1
python scripts/child_classes.py $O/prof8m > $O/classes8m.txt 2>&1
python scripts/pmc_tables.py hist $O/pmc_fetch $O/pmc_write $O/pmc_hist.json > /dev/null 2> $O/pmc_hist.err
python scripts/pmc_tables.py lambda $O/pmc_sq $O/lambda_pmc.json > /dev/null 2> $O/pmc_lambda.err
cp $O/prof/bench_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
cp $O/prof8m/bench_kernel_stats.csv $O/8M_kernel_stats.csv 2>/dev/null
# scoring (Metric 2): config 5 at full size, balanced and leaf-wise shaped; trace + counters at 1M documents x 1000 trees
timeout 400 python scripts/score_fullsize.py > $O/score_full.txt 2>&1; stamp score_full
timeout 400 python scripts/score_fullsize.py --ragged > $O/score_full_ragged.txt 2>&1; stamp score_full_ragged
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_score -o s --output-format csv -- python scripts/score_bench.py --trees 1000 --docs 1000000 > $O/score_traced.txt 2> $O/prof_score.err; stamp score_trace
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $O/pmc_score1 -o pmc --output-format csv -- python scripts/score_bench.py --trees 1000 --docs 1000000 > /dev/null 2> $O/pmc_score1.err; stamp score_pmc1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD -d $O/pmc_score2 -o pmc --output-format csv -- python scripts/score_bench.py --trees 1000 --docs 1000000 > /dev/null 2> $O/pmc_score2.err; stamp score_pmc2
python scripts/prof_summary.py $O/prof_score s 2>&1 | cut -c1-160 > $O/score_summary.md
cp $O/prof_score/s_kernel_stats.csv $O/score_kernel_stats.csv 2>/dev/null
python - <<'PY' > $O/score_pmc.md 2>&1
import csv, glob, collections
O = "gpurun_out/r06z"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_score1", "pmc_score2"):
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            k = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0].replace("void ", ""))
            per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for (did, name), cs in per.items():
            if name.startswith("k_score") or name.startswith("k_doc_bins"):
                for c, v in cs.items():
                    agg[name][c].append(v)
names = sorted(agg)
cols = sorted({c for n in names for c in agg[n]})
print("| kernel | launches | " + " | ".join(cols) + " |")
print("|---|---|" + "---|" * len(cols))
for n in names:
    k = max(len(v) for v in agg[n].values())
    print(f"| `{n[:60]}` | {k} | " + " | ".join(f"{sum(agg[n][c]) / len(agg[n][c]):.4g}" if agg[n][c] else "" for c in cols) + " |")
PY
rm -rf $O/prof $O/prof8m $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/prof_score $O/pmc_score1 $O/pmc_score2
cat $O/times.txt; tail -3 $O/pytest_gpu.txt; head -c 600 $O/bench.json
