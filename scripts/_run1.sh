set -x
mkdir -p gpurun_out/r03a
cd /root/repo
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03a/tests.log 2>&1
tail -3 gpurun_out/r03a/tests.log
python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
cat gpurun_out/r03a/bench.json | head -c 600
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r03a/prof -o bench -- python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 5 > gpurun_out/r03a/bench_traced.json 2> gpurun_out/r03a/prof.err
find gpurun_out/r03a/prof -name "*.csv" | head
