set -x
cd /root/repo
mkdir -p gpurun_out/r03n
python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -5 > gpurun_out/r03n/wide_tests.txt
python scripts/wide_bench.py > gpurun_out/r03n/wide_bench_8192.txt 2>&1
QR_HIP_LIB=/root/repo/quickrank_amd/lib/libqr_w4096.so python scripts/wide_bench.py > gpurun_out/r03n/wide_bench_4096.txt 2>&1
