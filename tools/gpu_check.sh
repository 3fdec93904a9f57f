# GPU tests, then the experiment harness (timing + VALU counters); everything under gpurun_out/
set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
timeout -k 10 120 tools/kexp 65536 10 1 > gpurun_out/kexp_a.log 2>&1
timeout -k 10 120 tools/kexp 1048576 3 1 > gpurun_out/kexp_b.log 2>&1
bash tools/prof_kexp_pmc.sh kexp_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" > gpurun_out/kexp_sq.txt 2>&1
