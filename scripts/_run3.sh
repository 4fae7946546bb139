set -x
cd /root/repo
mkdir -p gpurun_out/r03c
hipcc --offload-arch=gfx950 -O3 -w -o scripts/ubench/icache scripts/ubench/icache.hip && ./scripts/ubench/icache > gpurun_out/r03c/icache.txt 2>&1
QR_TIMING_LIB=libqr_steptiming_noinl.so python scripts/step_timing.py 2 > gpurun_out/r03c/step_timing_noinl.txt 2>&1
