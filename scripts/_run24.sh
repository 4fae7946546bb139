set -x
cd /root/repo
mkdir -p gpurun_out/r03x
for s in 11 12 13; do
  python tests/tools/fuzz_parity.py 300 $s 2>&1 | tail -2 >> gpurun_out/r03x/fuzz.txt
done
python tests/tools/long_query_check.py 2>&1 | tail -5 > gpurun_out/r03x/long_query.txt
python tests/tools/batch_vs_single_check.py 2>&1 | tail -5 > gpurun_out/r03x/batch_vs_single.txt
python tests/tools/big_leaves_check.py 2>&1 | tail -5 > gpurun_out/r03x/big_leaves.txt
