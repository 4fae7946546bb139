set -x
cd /root/repo
mkdir -p gpurun_out/r03h
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 > gpurun_out/r03h/parity.txt
python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 > gpurun_out/r03h/bench.json 2> gpurun_out/r03h/bench.err
python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 > gpurun_out/r03h/bench2.json 2> gpurun_out/r03h/bench2.err
python bench.py --queries 80000 --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 3 > gpurun_out/r03h/bench8m.json 2> gpurun_out/r03h/bench8m.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r03h/prof -o bench --output-format csv -- python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 5 > gpurun_out/r03h/bench_traced.json 2> gpurun_out/r03h/prof.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r03h/prof8m -o bench --output-format csv -- python bench.py --queries 80000 --no-extras --no-cpu-baseline --no-scoring --steps 12 --warmup 3 > gpurun_out/r03h/bench8m_traced.json 2> gpurun_out/r03h/prof8m.err
