set -x
cd /root/repo
mkdir -p gpurun_out/r03b
hipcc --offload-arch=gfx950 -O3 -w -o scripts/ubench/launch_chain scripts/ubench/launch_chain.hip && ./scripts/ubench/launch_chain > gpurun_out/r03b/launch_chain.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w -o scripts/ubench/lds_atomic scripts/ubench/lds_atomic.hip && ./scripts/ubench/lds_atomic > gpurun_out/r03b/lds_atomic.txt 2>&1
python scripts/step_timing.py 3 > gpurun_out/r03b/step_timing.txt 2>&1
QR_BENCH_NO_EVENTS=1 python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 > gpurun_out/r03b/bench_noev.json 2> gpurun_out/r03b/bench_noev.err
python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 > gpurun_out/r03b/bench_ev.json 2> gpurun_out/r03b/bench_ev.err
QR_BENCH_NO_EVENTS=1 python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 > gpurun_out/r03b/bench_noev2.json 2> gpurun_out/r03b/bench_noev2.err
