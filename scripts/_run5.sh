set -x
cd /root/repo
mkdir -p gpurun_out/r03e
hipcc --offload-arch=gfx950 -O3 -w -o scripts/ubench/noop_probe scripts/ubench/noop_probe.hip && ./scripts/ubench/noop_probe > gpurun_out/r03e/noop_probe.txt 2>&1
python scripts/step_timing.py 2 > gpurun_out/r03e/step_timing.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -x -q 2>&1 | tail -3 > gpurun_out/r03e/parity.txt
python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 > gpurun_out/r03e/bench.json 2> gpurun_out/r03e/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r03e/prof -o bench --output-format csv -- python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 5 > gpurun_out/r03e/bench_traced.json 2> gpurun_out/r03e/prof.err
