# rocprofv3 kernel trace + stats of the default bench command (hipGraph replays as in the bench line: kernels
# launched from a graph are 0.5-0.7 us shorter than eager ones; add --no-graph for eager launches)
# usage: prof_kernel.sh <outname> [bench args]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 500 --no-cpu --no-large --no-fp64 "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/$(basename $OUT)_kernel_stats.csv
head -6 $f
