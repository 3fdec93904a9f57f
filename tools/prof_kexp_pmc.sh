# PMC pass over the experiment harness; usage: prof_kexp_pmc.sh <outname> "<counters>"
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $OUT -- $GRAFT_REPO_ROOT/tools/kexp 65536 10 1 > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT
