set -x
cd /root/repo
mkdir -p gpurun_out/r03t
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03t/tests.log 2>&1
tail -3 gpurun_out/r03t/tests.log
