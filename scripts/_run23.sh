set -x
cd /root/repo
mkdir -p gpurun_out/r03w
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_wide.py -x -q -k "obl or Obl or oblivious or config4" 2>&1 | tail -4 > gpurun_out/r03w/tests.txt
python scripts/obl_bench.py 6 > gpurun_out/r03w/obl.txt 2>&1
python scripts/obl_bench.py 6 >> gpurun_out/r03w/obl.txt 2>&1
python scripts/obl_bench.py 4 >> gpurun_out/r03w/obl.txt 2>&1
