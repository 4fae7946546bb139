set -x
cd /root/repo
mkdir -p gpurun_out/r03d
hipcc --offload-arch=gfx950 -O3 -w -o scripts/ubench/launch_chain scripts/ubench/launch_chain.hip && ./scripts/ubench/launch_chain > gpurun_out/r03d/launch_chain.txt 2>&1
python scripts/step_timing.py 2 > gpurun_out/r03d/step_timing.txt 2>&1
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 > gpurun_out/r03d/parity.txt
python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 > gpurun_out/r03d/bench.json 2> gpurun_out/r03d/bench.err
python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 > gpurun_out/r03d/bench2.json 2> gpurun_out/r03d/bench2.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r03d/prof -o bench --output-format csv -- python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 5 > gpurun_out/r03d/bench_traced.json 2> gpurun_out/r03d/prof.err
