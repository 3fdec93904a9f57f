# Everything the committed profiles of the current library come from (results under gpurun_out/r3_*): the bench line of the driver's
# command and of the default command, rocprofv3 kernel stats of the headline's launches (twenty batches per launch), of
# one launch per step and of the 2^20-agent launch, and the HBM traffic counter passes (one rocprofv3 --pmc run per
# counter group) of the headline's launch.
set -e
: > gpurun_out/r3_rows_sweep.txt
timeout -k 10 500 python -u bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_steps20.json 2> gpurun_out/r3_bench_steps20.err
bash tools/prof_kernel.sh r3_stats > gpurun_out/r3_stats.txt 2>&1
bash tools/prof_kernel.sh r3_stats_single --fuse 1 --steps 2000 --warmup 200 > gpurun_out/r3_stats_single.txt 2>&1
bash tools/prof_pmc.sh r3_pmc_fsize "FETCH_SIZE" > gpurun_out/r3_pmc_fsize.txt 2>&1
bash tools/prof_pmc.sh r3_pmc_write "WRITE_SIZE TCC_EA0_WRREQ_sum" > gpurun_out/r3_pmc_write.txt 2>&1
bash tools/prof_pmc.sh r3_pmc_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" > gpurun_out/r3_pmc_sq.txt 2>&1
python3 tools/traffic_json.py gpurun_out/r3_pmc_fsize gpurun_out/r3_pmc_write gpurun_out/r3_traffic.json 1310720 k_group_batches "bench.py --steps 100 --warmup 20 --ramp-ms 5 --headline-only" > gpurun_out/r3_traffic.txt 2>&1
bash tools/prof_kernel.sh r3_large_stats --large-only --steps 300 > gpurun_out/r3_large_stats.txt 2>&1
bash tools/prof_pmc.sh r3_large_fsize "FETCH_SIZE" --large-only --steps 5 > gpurun_out/r3_large_fsize.txt 2>&1
bash tools/prof_pmc.sh r3_large_write "WRITE_SIZE" --large-only --steps 5 > gpurun_out/r3_large_write.txt 2>&1
python3 tools/traffic_json.py gpurun_out/r3_large_fsize gpurun_out/r3_large_write gpurun_out/r3_traffic_large.json 1048576 "k_group<" "bench.py --large-only --steps 5" > gpurun_out/r3_traffic_large.txt 2>&1
timeout -k 10 500 python -u bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
# fp64: kernel stats and HBM counters of the fused launch (twenty 65536 x 16 batches per sccbf_solve_batches_f64 launch)
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_f64_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_batches.py --dtype f64 --ks 20 --reps 200 > $GRAFT_REPO_ROOT/gpurun_out/r3_f64_stats.log 2>&1 )
cp $(find gpurun_out/r3_f64_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r3_f64_kernel_stats.csv
: > gpurun_out/r3_f64_pmc.txt
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  rm -rf gpurun_out/r3_f64_pmc.d
  ( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 160 rocprofv3 --pmc $CTRS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_f64_pmc.d -- python3 $GRAFT_REPO_ROOT/tools/bench_batches.py --dtype f64 --ks 20 --reps 20 > $GRAFT_REPO_ROOT/gpurun_out/r3_f64_pmc.log 2>&1 )
  python3 tools/pmc_summary.py gpurun_out/r3_f64_pmc.d >> gpurun_out/r3_f64_pmc.txt
done
rm -rf gpurun_out/r3_f64_pmc.d
# long obstacle lists: launch times of every model from per-agent lists and from tiles, then the counters of the three kernel families
for L in agents tiles; do
  echo "== layout $L" >> gpurun_out/r3_rows_sweep.txt
  timeout -k 10 500 python -u tools/bench_rows.py --layout $L --rows 16,64,128,129,200,256,257,512,1024 2>&1 | grep "^M" >> gpurun_out/r3_rows_sweep.txt
done
# the same sweep in double precision (bicycle and unicycle models)
: > gpurun_out/r3_rows_sweep_f64.txt
for L in agents tiles; do
  echo "== layout $L" >> gpurun_out/r3_rows_sweep_f64.txt
  timeout -k 10 500 python -u tools/bench_rows.py --dtype f64 --layout $L --rows 16,64,128,129,200,257,512,1024 --models dbm,dum 2>&1 | grep "^M" >> gpurun_out/r3_rows_sweep_f64.txt
done
LAYOUT=agents bash tools/prof_rows_pmc.sh 128 257 1024
LAYOUT=tiles bash tools/prof_rows_pmc.sh 128 257 1024
# long lists with ten independent batches per launch (sccbf_solve_batches_f32) against one launch per batch
: > gpurun_out/r3_rows_fused.txt
for R in 64 128 257 512; do for L in tiles agents; do
  echo "== $R rows, layout $L" >> gpurun_out/r3_rows_fused.txt
  timeout -k 10 300 python -u tools/bench_batches.py --ks 1,10 --ring 10 --reps 30 --rows $R --layout $L 2>&1 | grep "batches per launch" >> gpurun_out/r3_rows_fused.txt
done; done
