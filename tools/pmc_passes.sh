#!/bin/bash
# usage: tools/pmc_passes.sh <outdir under gpurun_out> <script.py> -- runs the SQ counter passes one at a time
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; SCRIPT=$R/$2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 $SCRIPT > $OUT.p$i.log 2>&1 && echo "pass $i ok"
done
