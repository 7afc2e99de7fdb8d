#!/bin/bash
# Instruction and cycle counters of the evaluator kernels on synthetic activations (32768 leaves), one GPU
# (run through gpurun from the repo root): three --pmc passes per kernel, kernel trace only, summarised into
# gpurun_out/<tag>_pmc_evaluator.csv.  The SQ_* cycle counters tally quad-cycles (MI355X_MICROARCH.md).
set -e -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
DIRS=""
for k in ${KERNELS:-conv stem_folded attn heads}; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    d=$OUT/${TAG}_pmc_ev_${k}_$i
    echo "== $k pass $i"; date
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/tools/probe_nn.py $k 0 5 > $OUT/${TAG}_pmc_ev_${k}_$i.log 2>&1
    DIRS="$DIRS $d"
  done
done
cd $R
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_evaluator.csv $DIRS --only k_
rm -rf $DIRS
