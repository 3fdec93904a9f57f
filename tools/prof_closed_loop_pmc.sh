set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_cl
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_closed_loop.py "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT
