#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "overlap or reproducible or batching or options or tiny_pipeline_parity" > $O/r3g_parity_small.log 2>&1; tail -6 $O/r3g_parity_small.log
for i in 1 2; do
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$O/r3g_bench.err | tail -1 > $O/r3g_bench.json
python -c "import sys,json; d=json.load(open('$O/r3g_bench.json')); print(round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['frac'],4), d['roofline']['step_kernel_ms'], round(d['roofline']['launches_per_denoise_step']))" || tail -5 $O/r3g_bench.err
done
