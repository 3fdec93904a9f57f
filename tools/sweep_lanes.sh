set -e
for L in 1 2 4 8 16 64; do
  timeout -k 10 120 python -u bench.py --lanes $L --steps 1000 --warmup 100 --no-cpu >> gpurun_out/sweep1.log 2>&1
done
