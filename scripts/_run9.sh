set -x
cd /root/repo
mkdir -p gpurun_out/r03j
python -m pytest tests/test_gpu_transport.py -x -q 2>&1 | tail -15 > gpurun_out/r03j/transport.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03j/tests.log 2>&1
tail -3 gpurun_out/r03j/tests.log
