set -x
cd /root/repo
mkdir -p gpurun_out/r03z4
for nb in 0 1 0 1; do
  if [ $nb = 1 ]; then export QR_NO_BATCH=1; else unset QR_NO_BATCH; fi
  echo "QR_NO_BATCH=$nb" >> gpurun_out/r03z4/wide_ab.txt
  WB_ITERS=43 WB_NTHR=1024,4096 python scripts/wide_bench.py 2>&1 | grep nthresholds >> gpurun_out/r03z4/wide_ab.txt
done
cat gpurun_out/r03z4/wide_ab.txt
