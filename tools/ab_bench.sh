mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-tree-only --steps 30 > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', d['value'], d['backprop_kernel_avg_us'], d['roofline']['avg_launch_us'])"; }
