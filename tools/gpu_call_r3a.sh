#!/bin/bash
# round 3, call A: parity table (all legs), new kernel checks, bench A/B of the fp32 residual stream, GEMM pitch probe
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
date +%s > $O/r3a_t0
timeout 1500 python tools/gpu_parity_table.py > $O/r3a_parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/r3a_parity.log
cp $O/fullsize_parity.json $O/r3a_fullsize_parity.json 2>/dev/null
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "stream_f32 or layernorm or linear_3072" > $O/r3a_kchecks.log 2>&1; tail -3 $O/r3a_kchecks.log
for tag in f32 16bit f32 16bit; do
  extra=""; [ $tag = 16bit ] && extra="--residual-16bit"
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra 2>$O/r3a_bench_$tag.err | tail -1 > $O/r3a_bench_$tag.json
  python -c "import sys,json; d=json.load(open('$O/r3a_bench_$tag.json')); print('$tag', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['frac'],4), d['roofline']['step_kernel_ms'])" || tail -5 $O/r3a_bench_$tag.err
done
timeout 300 python tools/gpu_r3_probe.py > $O/r3a_probe.log 2>&1; tail -30 $O/r3a_probe.log
echo "elapsed $(( $(date +%s) - $(cat $O/r3a_t0) )) s"
