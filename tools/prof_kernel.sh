# rocprofv3 kernel trace + stats of the bench's own launches (twenty batches per sccbf_solve_batches_f32 launch by
# default; add --fuse 1 for one sccbf_solve_f32 launch per step, --large-only for the 2^20-agent launch)
# usage: prof_kernel.sh <outname> [bench args]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4000 --warmup 400 --headline-only "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/$(basename $OUT)_kernel_stats.csv
head -6 $f
