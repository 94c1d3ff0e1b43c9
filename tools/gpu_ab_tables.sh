#!/bin/bash
# Same-box A/B of tuning tables (or library builds): tools/gpu_ab_tables.sh TAG "ENV=... [ENV2=...]" "ENV=..." ... -- every arm runs the headline loop
# (bench.py --steps 3, no optional legs) twice, arms interleaved, and prints images/s and loop ms per denoising step.
TAG=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
  i=0
  for arm in "$@"; do
    i=$((i+1))
    env $arm timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp16-leg --no-pmc 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('arm$i rep$rep [$arm]', round(d['value'],4), 'img/s  loop', round(d['loop_ms_per_denoise_step'],3), 'ms/step  parts', d['call_parts_ms'])" | tee -a gpurun_out/$TAG.log
  done
done
