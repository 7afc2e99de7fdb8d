#!/bin/bash
# Profiles of a round on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench, FETCH_SIZE / WRITE_SIZE passes (separate, kernel trace only),
#   stall counters of the tree kernels on the tree-only probe.  Outputs under gpurun_out/<tag>_*.
set -e -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
echo "== FETCH_SIZE"; date
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 6 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
echo "== WRITE_SIZE"; date
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py --steps 6 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmc_write.err
echo "== stall counters (tree-only probe)"; date
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_stall -- python3 $R/tools/probe_select.py 8192 6 12 > $OUT/${TAG}_pmc_stall.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_insts -- python3 $R/tools/probe_select.py 8192 6 12 > $OUT/${TAG}_pmc_insts.log 2>&1
echo "== done"; date
cd $R
# (the selection kernel's corrected HBM bytes per launch go to profiles/traffic_select.json under the default bench's key)
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_fetch_write_by_kernel.csv $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write --only k_ \
    --traffic "k_select8x4|games=8192|n_playout=200|K=4|streams=1|evaluator=cnn|lead_in=12" --kernel k_select8x4
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_tree_kernels_stall.csv $OUT/${TAG}_pmc_stall $OUT/${TAG}_pmc_insts --only k_
find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_bench_kernel_stats.csv
# the raw per-dispatch files are large: keep the summaries only
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_stall $OUT/${TAG}_pmc_insts
find $OUT/${TAG}_trace -name "*kernel_trace.csv" -size +20M -delete
