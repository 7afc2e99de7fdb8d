#!/bin/bash
# The bench lines of a round, one GPU (run through gpurun from the repo root): headline, three streams, tree-only,
# BASELINE config 2's per-GPU share, config 5's two shapes with and without the table, config 3/4 (Othello).
TAG=${1:-r03}
OUT=gpurun_out
run() { name=$1; shift; echo "== $name: $*"; "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err || { echo "FAILED $name"; tail -5 $OUT/${TAG}_$name.err; }; tail -c 400 $OUT/${TAG}_$name.json; echo; }
run bench_final python bench.py
run bench_streams3 python bench.py --no-cpu-baseline --no-tree-only --streams 3
run dropin python tools/measure_dropin.py 8192 4
run bench_config2_1gpu python bench.py --no-cpu-baseline --no-tree-only --n-playout 800 --steps 20
run bench_config5_16384_table python bench.py --no-cpu-baseline --no-tree-only --games 16384 --table 20 --steps 40
run othello_c3_native python tools/bench_othello.py 4096 400 6 2
EVALUATOR=hash run othello_c3_tree_only python tools/bench_othello.py 4096 400 10 4
