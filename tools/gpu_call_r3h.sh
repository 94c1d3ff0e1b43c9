#!/bin/bash
# round 3, call H: fp8 attention (probe layout, quant, kernel, engine), RCCL C ABI at world 1, a13 test, multi-GPU script at world 1
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "f8 or probe" > $O/r3h_kchecks.log 2>&1; tail -25 $O/r3h_kchecks.log | cut -c1-200
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "rccl or embeddings or fp8 or two_ranks or bench_on" > $O/r3h_misc.log 2>&1; tail -12 $O/r3h_misc.log | cut -c1-250
timeout 300 python tools/multi_gpu_check.py $O/r3h_multigpu_world1.json 2>&1 | tail -2 | cut -c1-300
for tag in f16 f16_fp8; do
  extra="--dtype f16"; [ $tag = f16_fp8 ] && extra="--dtype f16 --attn-fp8"
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $extra 2>$O/r3h_bench_$tag.err | tail -1 > $O/r3h_bench_$tag.json
  python -c "import sys,json; d=json.load(open('$O/r3h_bench_$tag.json')); print('$tag', round(d['value'],4), round(d['ms_per_step'],1), d['roofline']['step_kernel_ms'], d['roofline'].get('attn_fwd'))" || tail -5 $O/r3h_bench_$tag.err
done
