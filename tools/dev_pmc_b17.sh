#!/bin/bash
# Developer aid: PMC counters of the fused Block17 kernels (one counter group per pass; run through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU" "MfmaUtil" "MemUnitStalled" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp -d $O/pmcb17_$tag -o p -- python3 $R/tools/dev_block17.py 180 > /dev/null 2>&1
  for c in $grp; do python3 $R/tools/pmc_dump.py $O/pmcb17_$tag $c block17; done
  rm -rf $O/pmcb17_$tag
done
