# Per-kernel SQ counters of any command of this repo, on the GPU box (counters only: no trace flags beside --pmc):
#   bash tools/pmc_kernels.sh <out.txt> <kernel substring> "<COUNTERS, at most ~8 per pass>" python3 <script> [args]
# A second/third pass with other counters is another call of this script.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; export TMPDIR=/tmp
OUT=$1; FLT=$2; CS=$3; shift 3
PROG=$1; SCRIPT=$R/$2; shift 2
cd /tmp; rm -rf $O/pmc_x
rocprofv3 --pmc $CS -d $O/pmc_x -o p -- $PROG $SCRIPT "$@" > $O/pmc_x.out 2> $O/pmc_x.err || { tail -5 $O/pmc_x.err; exit 1; }
for c in $CS; do python3 $R/tools/pmc_dump.py $O/pmc_x $c "$FLT"; done >> $O/$OUT
rm -rf $O/pmc_x; cat $O/$OUT
