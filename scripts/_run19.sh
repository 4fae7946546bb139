set -x
cd /root/repo
mkdir -p gpurun_out/r03s
for v in base ns4 ns5 base ns4 ns5; do
  if [ $v = base ]; then L=""; else L="QR_HIP_LIB=/root/repo/quickrank_amd/lib/libqr_$v.so"; fi
  env $L python bench.py --queries 80000 --no-extras --no-cpu-baseline --no-scoring --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 8M', d['ms_per_step'])" >> gpurun_out/r03s/ab.txt
done
for v in base ns4 ns5 base ns4 ns5; do
  if [ $v = base ]; then L=""; else L="QR_HIP_LIB=/root/repo/quickrank_amd/lib/libqr_$v.so"; fi
  env $L python bench.py --no-extras --no-cpu-baseline --no-scoring --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 1M', d['ms_per_step'])" >> gpurun_out/r03s/ab.txt
done
cat gpurun_out/r03s/ab.txt
