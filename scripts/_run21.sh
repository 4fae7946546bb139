set -x
cd /root/repo
mkdir -p gpurun_out/r03u
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_wide.py tests/test_gpu_cli.py -x -q 2>&1 | tail -15 > gpurun_out/r03u/tests.txt
