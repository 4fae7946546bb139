set -x
cd /root/repo
mkdir -p gpurun_out/r03p
python scripts/lambda_ablation.py > gpurun_out/r03p/lambda_ablation.txt 2>&1
