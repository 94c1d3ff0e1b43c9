#!/bin/bash
# round 3, call K: the softmax forms of the ping-pong attention kernel side by side (tune bits 28-29), against the previous build
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "attn_self" > $O/r3k_kchecks.log 2>&1; tail -5 $O/r3k_kchecks.log | cut -c1-200
for tag in prev new prev new; do
  lib=$R/idm-vton_amd/libidmvton_hip.so; [ $tag = prev ] && lib=$R/idm-vton_amd/libidmvton_hip_prevattn.so
  echo "== $tag"; IDMVTON_HIP_LIB=$lib timeout 120 python tools/gpu_quick_attn.py 2>&1 | grep -v amdgpu.ids
done | tee $O/r3k_quick_attn.log
