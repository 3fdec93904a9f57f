set -e
for t in 256 1024 4096 16384; do
  timeout -k 10 100 tools/kexp_t$t 65536 10 1 > gpurun_out/tile_${t}_a.log 2>&1
  timeout -k 10 100 tools/kexp_t$t 1048576 3 1 > gpurun_out/tile_${t}_b.log 2>&1
done
