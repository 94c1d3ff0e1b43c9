#!/bin/bash
# round 3, call I: LayerNorm fold with per-row final statistics from the producer's last-arriving tile: kernel checks (all GEMM / conv checks
# as regression), cost probe, engine parity with the option on, bench A/B off / on
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "ln_fold or gemm or conv or linear or stream_f32 or geglu" > $O/r3i_kchecks.log 2>&1; tail -25 $O/r3i_kchecks.log | cut -c1-200
timeout 300 python tools/gpu_r3_lnprobe.py 2>&1 | tee $O/r3i_lnprobe.log | cut -c1-200
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "options" > $O/r3i_options.log 2>&1; tail -5 $O/r3i_options.log | cut -c1-250
for tag in off on off2 on2; do
  extra=""; case $tag in on*) extra="--fuse-ln";; esac
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra 2>$O/r3i_bench_$tag.err | tail -1 > $O/r3i_bench_$tag.json
  python -c "import sys,json; d=json.load(open('$O/r3i_bench_$tag.json')); print('$tag', round(d['value'],4), round(d['ms_per_step'],1), d['roofline']['step_kernel_ms'], d['roofline']['launches_per_denoise_step'])" || tail -5 $O/r3i_bench_$tag.err
done
