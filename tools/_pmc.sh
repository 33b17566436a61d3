R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
CS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
rm -rf $O/pmc_x; rocprofv3 --pmc $CS -d $O/pmc_x -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_x.json 2> $O/pmc_x.err || { tail -5 $O/pmc_x.err; exit 1; }
for c in $CS; do python3 $R/tools/pmc_dump.py $O/pmc_x $c "${1:-wgrad}"; done > $O/r03_pmc_wgrad.txt
rm -rf $O/pmc_x; cat $O/r03_pmc_wgrad.txt
