#!/bin/bash
# round 4, call I: fused cross-attention with K fragments prefetched under the main loop; LayerNorm rows-per-wave A/B
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "xattn or layernorm" > $O/r4i_kchecks.log 2>&1; tail -6 $O/r4i_kchecks.log | cut -c1-300
timeout 600 python tools/gpu_r4_xattn.py 2>&1 | grep -v amdgpu.ids | tee $O/r4i_xattn_probe.log
for r in 1 2 1 2; do IDMVTON_LN_RPW=$r timeout 300 python tools/gpu_r4_ln.py 2>&1 | grep -v amdgpu.ids; done | tee $O/r4i_ln_probe.log
