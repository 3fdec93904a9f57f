# SQ instruction counters of the filter kernel for several lanes-per-QP choices
set -e
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
for L in 1 2 4; do
  bash tools/prof_pmc.sh sq_l$L "$C" --lanes $L > gpurun_out/sq_l$L.txt 2>&1
done
