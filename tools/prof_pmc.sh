# one rocprofv3 --pmc pass over a short bench run; usage: prof_pmc.sh <outname> "<counters>" [bench args]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
CTRS="$2"
shift 2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 160 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --ramp-ms 5 --headline-only "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT
