set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r03v
for m in obl mslr; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r03v/prof_$m -o p --output-format csv -- python scripts/prof_modes.py $m > /dev/null 2> gpurun_out/r03v/prof_$m.err
  python scripts/prof_summary.py gpurun_out/r03v/prof_$m p > gpurun_out/r03v/summary_$m.md 2>&1
done
