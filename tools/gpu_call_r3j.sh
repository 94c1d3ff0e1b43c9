#!/bin/bash
# round 3, call J: ping-pong attention with packed softmax arithmetic and the lazy running max (bf16): all attention checks, launch timing and
# whole-pipeline A/B against the previous build of attention.hip (idm-vton_amd/libidmvton_hip_prevattn.so, built from HEAD~)
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "attn" > $O/r3j_kchecks.log 2>&1; tail -25 $O/r3j_kchecks.log | cut -c1-200
for tag in prev new prev new; do
  lib=$R/idm-vton_amd/libidmvton_hip.so; [ $tag = prev ] && lib=$R/idm-vton_amd/libidmvton_hip_prevattn.so
  echo "== $tag"; IDMVTON_HIP_LIB=$lib timeout 120 python tools/gpu_quick_attn.py 2>&1 | grep -v amdgpu.ids
done | tee $O/r3j_quick_attn.log
for tag in prev new prev new; do
  lib=$R/idm-vton_amd/libidmvton_hip.so; [ $tag = prev ] && lib=$R/idm-vton_amd/libidmvton_hip_prevattn.so
  IDMVTON_HIP_LIB=$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r3j_bench_$tag.json
  python -c "import sys,json; d=json.load(open('$O/r3j_bench_$tag.json')); print('$tag', round(d['value'],4), round(d['ms_per_step'],1), d['roofline']['step_kernel_ms'], d['roofline']['attn_fwd'])"
done | tee $O/r3j_bench_ab.log
