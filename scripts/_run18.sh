set -x
cd /root/repo
mkdir -p gpurun_out/r03r
QR_PAD_MIN_DOCS=1000 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 > gpurun_out/r03r/parity_padded.txt
for v in 3000000 999999999 3000000 999999999; do
  QR_PAD_MIN_DOCS=$v python bench.py --queries 80000 --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad_min $v 8M', d['ms_per_step'])" >> gpurun_out/r03r/ab.txt
done
for v in 1000 999999999 1000 999999999; do
  QR_PAD_MIN_DOCS=$v python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad_min $v 1M', d['ms_per_step'], d['roofline']['avg_launch_us'])" >> gpurun_out/r03r/ab.txt
done
cat gpurun_out/r03r/ab.txt
