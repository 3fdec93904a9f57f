# Everything the committed profiles come from: GPU tests, the default bench line, the rocprofv3
# kernel stats of the same workload, and the two HBM traffic counter passes.
set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
timeout -k 10 300 python -u bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
bash tools/prof_kernel.sh prof_stats > gpurun_out/prof_stats.txt 2>&1
bash tools/prof_pmc.sh pmc_fetch "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" > gpurun_out/pmc_fetch.txt 2>&1
bash tools/prof_pmc.sh pmc_fsize "FETCH_SIZE" > gpurun_out/pmc_fsize.txt 2>&1
bash tools/prof_pmc.sh pmc_write "WRITE_SIZE TCC_EA0_WRREQ_sum" > gpurun_out/pmc_write.txt 2>&1
bash tools/prof_pmc.sh pmc_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" > gpurun_out/pmc_sq.txt 2>&1
