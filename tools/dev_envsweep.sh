#!/bin/bash
# Developer aid: does a HIP runtime knob change the per-launch gap of the replayed step graph?  (run through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  env "$@" timeout -k 10 200 python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/env.json 2> $R/gpurun_out/env.err || { echo "fail $*"; tail -3 $R/gpurun_out/env.err; return; }
  python -c "
import json
d=json.load(open('$R/gpurun_out/env.json')); print('$*', d['ms_per_step'], d['ms_per_step_train_only'])"
}
run X=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run ROC_USE_FGS_KERNARG=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
