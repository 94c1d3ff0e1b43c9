#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "ln_fold" > $O/r3e_kchecks.log 2>&1; tail -3 $O/r3e_kchecks.log
for tag in fuse nofuse fuse nofuse; do
  extra=""; [ $tag = nofuse ] && extra="--no-fuse-ln"
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra 2>$O/r3e_bench_$tag.err | tail -1 > $O/r3e_bench_$tag.json
  python -c "import sys,json; d=json.load(open('$O/r3e_bench_$tag.json')); print('$tag', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['frac'],4), d['roofline']['step_kernel_ms'], round(d['roofline']['launches_per_denoise_step']))" || tail -5 $O/r3e_bench_$tag.err
done
