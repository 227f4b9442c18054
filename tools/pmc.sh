#!/bin/bash
# PMC passes over tools/pmc_frame.py (one process per pass; SQ block: at most 8 counters per pass, TCC: FETCH_SIZE and WRITE_SIZE
# never together — `rocprofv3` aborts with "error code 38: Request exceeds the capabilities of the hardware to collect" otherwise).
# usage: tools/pmc.sh TAG   -> gpurun_out/pmc_TAG_*/ and gpurun_out/pmc_TAG_summary.json
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/tools/pmc_frame.py > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
done <<LIST
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_LDS
GRBM_GUI_ACTIVE
LIST
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_* > $R/gpurun_out/pmc_${TAG}_summary.json
