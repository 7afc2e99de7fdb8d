#!/bin/bash
# Instruction and cycle counters of one Othello 256-channel convolution layer at 16384 samples (run through gpurun
# from the repo root): three --pmc passes, kernel trace only, summarised into gpurun_out/oth_pmc.csv.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0; DIRS=""
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
 i=$((i+1)); d=$OUT/oth_pmc_$i
 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/tools/probe_othello_conv.py 16384 > $OUT/oth_pmc_$i.log 2>&1
 DIRS="$DIRS $d"
done
cd $R && python3 tools/pmc_summary.py $OUT/oth_pmc.csv $DIRS --only k_oth && rm -rf $DIRS
