set -x
cd /root/repo
mkdir -p gpurun_out/r03z3
python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -15 > gpurun_out/r03z3/wide_tests.txt
python -m pytest tests/test_gpu_fullsize.py -x -q -k "1024 or wide" 2>&1 | tail -5 >> gpurun_out/r03z3/wide_tests.txt
python scripts/wide_bench.py > gpurun_out/r03z3/wide_bench.txt 2>&1
bash scripts/wide_prof.sh 1024 > gpurun_out/r03z3/wide_prof_1024.txt 2>&1
