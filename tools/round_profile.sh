# Everything the committed profiles come from: GPU tests, the default bench line, the rocprofv3
# kernel stats of the same command (launches of independent steps overlapping in three graph branches), of the
# same steps one at a time (--branches 1) and launched eagerly, and of the 2^20-agent launch, and the HBM traffic counter passes
# (one rocprofv3 --pmc run per counter group).  Results under gpurun_out/r2_*.
set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2_gpu_tests.log 2>&1
timeout -k 10 400 python -u bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
bash tools/prof_kernel.sh r2_stats > gpurun_out/r2_stats.txt 2>&1
bash tools/prof_kernel.sh r2_stats_serial --branches 1 > gpurun_out/r2_stats_serial.txt 2>&1
bash tools/prof_kernel.sh r2_stats_eager --no-graph --steps 500 --warmup 50 > gpurun_out/r2_stats_eager.txt 2>&1
bash tools/prof_pmc.sh r2_pmc_fsize "FETCH_SIZE" > gpurun_out/r2_pmc_fsize.txt 2>&1
bash tools/prof_pmc.sh r2_pmc_write "WRITE_SIZE TCC_EA0_WRREQ_sum" > gpurun_out/r2_pmc_write.txt 2>&1
bash tools/prof_pmc.sh r2_pmc_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" > gpurun_out/r2_pmc_sq.txt 2>&1
# the 2^20-agent launch (non-temporal slot loads): 300 walks over its ring of 3, warm-up as long (the clocks settle after some 10 ms)
bash tools/prof_kernel.sh r2_large_stats --large-only --steps 300 > gpurun_out/r2_large_stats.txt 2>&1
bash tools/prof_pmc.sh r2_large_fsize "FETCH_SIZE" --large-only --steps 5 > gpurun_out/r2_large_fsize.txt 2>&1
bash tools/prof_pmc.sh r2_large_write "WRITE_SIZE" --large-only --steps 5 > gpurun_out/r2_large_write.txt 2>&1
bash tools/prof_pmc.sh r2_large_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" --large-only --steps 5 > gpurun_out/r2_large_sq.txt 2>&1
# the closed loop around the filter: course tracking with the obstacles out of reach, then in the C4 field
bash tools/prof_closed_loop.sh --far > gpurun_out/r2_closed_loop_far.txt 2>&1
cp $(find gpurun_out/prof_cl -name "*kernel_stats.csv" | head -1) gpurun_out/r2_closed_loop_far_kernel_stats.csv
rm -rf gpurun_out/prof_cl
bash tools/prof_closed_loop.sh > gpurun_out/r2_closed_loop_c4.txt 2>&1
cp $(find gpurun_out/prof_cl -name "*kernel_stats.csv" | head -1) gpurun_out/r2_closed_loop_c4_kernel_stats.csv
