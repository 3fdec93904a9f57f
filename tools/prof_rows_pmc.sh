# Counters of the long-list kernels: for M in "$@" (default 128 257 1024) three rocprofv3 --pmc passes over
# tools/bench_rows.py (DBM, 65536 agents): HBM reads, HBM writes, instruction mix / LDS / occupancy.
# LAYOUT=tiles|agents|planes (default tiles) picks the obstacle layout.
# Output: gpurun_out/r3_rows_pmc_<layout>_<M>.txt
set -e
ROWS="${@:-128 257 1024}"
LAYOUT="${LAYOUT:-tiles}"
cd /tmp && export TMPDIR=/tmp
for M in $ROWS; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r3_rows_pmc_${LAYOUT}_$M
  : > $OUT.txt
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
    rm -rf $OUT.d
    timeout -s KILL 150 rocprofv3 --pmc $CTRS --output-format csv -d $OUT.d -- python3 $GRAFT_REPO_ROOT/tools/bench_rows.py --rows $M --models dbm --layout $LAYOUT > $OUT.log 2>&1 || { echo "pass failed: $CTRS" >> $OUT.txt; tail -3 $OUT.log >> $OUT.txt; continue; }
    grep "^M " $OUT.log >> $OUT.txt || true
    python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT.d >> $OUT.txt
  done
  rm -rf $OUT.d
done
