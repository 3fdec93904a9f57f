# rocprofv3 kernel trace + stats of the default bench (eager launches so every launch is a row)
# usage: prof_kernel.sh <outname> [bench args]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --no-cpu --no-large --no-fp64 --no-graph "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/$(basename $OUT)_kernel_stats.csv
head -6 $f
