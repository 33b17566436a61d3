#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun): default bench line, rocprofv3 kernel statistics and PMC passes of the SAME
# tile choices (FACENET_TUNE_CACHE), summaries under gpurun_out/ (copied to profiles/ afterwards).
#   gpurun --timeout 1200 -- 'bash tools/profile_r03.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
export FACENET_TUNE_CACHE=$O/r03_tile_cache.json
# keep the committed tile choices when there are some (RETUNE=1 times the variants afresh): tuning is noisy at the 1 % level
if [ -z "$RETUNE" ] && [ -f $R/profiles/r03_tile_cache.json ]; then cp $R/profiles/r03_tile_cache.json $FACENET_TUNE_CACHE;
elif [ -z "$RETUNE" ] && [ -f $R/profiles/r02_tile_cache.json ]; then cp $R/profiles/r02_tile_cache.json $FACENET_TUNE_CACHE; fi
cd /tmp
echo "[1] default bench (100 steps, CPU legs)"; python3 $R/bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err || exit 1
tail -c 400 $O/r03_bench_default.json; echo
echo "[2] kernel trace"; rocprofv3 --kernel-trace --stats -d $O/prof_r03 -o st -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_bench_under_rocprof.json 2> $O/prof_r03.err || exit 1
python3 $R/tools/rocpd_stats.py $(find $O/prof_r03 -name '*_results.db' | head -1) $O/r03_kernel_stats_bench_steps20.csv $O/r03_kernel_stats_summary.md || exit 1
for c in FETCH_SIZE WRITE_SIZE MfmaUtil VALUBusy OccupancyPercent; do
  echo "[3] pmc $c"; rocprofv3 --pmc $c -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
done
python3 $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/r03_pmc_hbm_traffic.json $O/r03_bench_default.json || exit 1
for c in MfmaUtil VALUBusy OccupancyPercent; do python3 $R/tools/pmc_dump.py $O/pmc_$c $c > $O/r03_pmc_$c.txt; done
rm -rf $O/prof_r03 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MfmaUtil $O/pmc_VALUBusy $O/pmc_OccupancyPercent
echo done
