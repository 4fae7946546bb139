set -x
cd /root/repo
mkdir -p gpurun_out/r03f
( time python bench.py > gpurun_out/r03f/bench_full.json 2> gpurun_out/r03f/bench_full.err ) 2> gpurun_out/r03f/bench_full.time
rocprofv3 --kernel-trace --stats -d gpurun_out/r03f/prof8m -o bench --output-format csv -- python bench.py --queries 80000 --no-extras --no-cpu-baseline --no-scoring --steps 12 --warmup 3 > gpurun_out/r03f/bench8m_traced.json 2> gpurun_out/r03f/prof8m.err
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03f/tests.log 2>&1
tail -3 gpurun_out/r03f/tests.log
