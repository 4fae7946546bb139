set -x
cd /root/repo
mkdir -p gpurun_out/r03k
python -m pytest tests/test_gpu_transport.py -x -q 2>&1 | tail -3 > gpurun_out/r03k/transport.txt
bash scripts/wide_prof.sh 1024 > gpurun_out/r03k/wide_prof_1024.txt 2>&1
python scripts/wide_bench.py > gpurun_out/r03k/wide_bench.txt 2>&1
