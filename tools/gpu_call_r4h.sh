#!/bin/bash
# round 4, call H: cross-attention fused into its query projection: kernel checks, timing against the two-launch form, model parity
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "xattn or attn_cross" > $O/r4h_kchecks.log 2>&1; tail -12 $O/r4h_kchecks.log | cut -c1-300
timeout 600 python tools/gpu_r4_xattn.py 2>&1 | grep -v amdgpu.ids | tee $O/r4h_xattn_probe.log
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r4h_parity.log 2>&1; tail -12 $O/r4h_parity.log | cut -c1-400
