set -x
cd /root/repo
mkdir -p gpurun_out/r03q
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 > gpurun_out/r03q/parity.txt
for v in "1 1" "0 1" "1 0" "0 1" "1 1" "0 2"; do
  set -- $v
  QR_STEPS_PLUS=$1 QR_CONT_STEPS=$2 python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 120 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plus $1 cont $2: 1M', d['ms_per_step'])" >> gpurun_out/r03q/ab.txt
done
QR_STEPS_PLUS=0 python bench.py --queries 80000 --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plus 0 8M', d['ms_per_step'])" >> gpurun_out/r03q/ab.txt
python bench.py --queries 80000 --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plus 1 8M', d['ms_per_step'])" >> gpurun_out/r03q/ab.txt
cat gpurun_out/r03q/ab.txt
