#!/bin/bash
# round 3, call D: LayerNorm fold -- kernel checks, model parity (tiny/mid + full size), bench A/B
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "ln_fold or stream_f32 or linear_ or geglu or vt_" > $O/r3d_kchecks.log 2>&1; tail -15 $O/r3d_kchecks.log
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r3d_parity_small.log 2>&1; tail -5 $O/r3d_parity_small.log
timeout 900 python tools/gpu_parity_table.py cfg2_unets cfg2_b2_2steps cfg2_30steps > $O/r3d_parity.log 2>&1; grep -E "hip_bf16 |hip_f16 |ref_fp16" $O/r3d_parity.log | cut -c1-150
for tag in fuse nofuse fuse nofuse; do
  extra=""; [ $tag = nofuse ] && extra="--no-fuse-ln"
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra 2>$O/r3d_bench_$tag.err | tail -1 > $O/r3d_bench_$tag.json
  python -c "import sys,json; d=json.load(open('$O/r3d_bench_$tag.json')); print('$tag', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['frac'],4), d['roofline']['step_kernel_ms'], round(d['roofline']['launches_per_denoise_step']))" || tail -5 $O/r3d_bench_$tag.err
done
