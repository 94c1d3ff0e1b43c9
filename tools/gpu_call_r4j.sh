#!/bin/bash
# round 4, call J: the 256x192 hand-scheduled tile + fused cross-attention fix: kernel checks, GEMM probe
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "h192 or h5f or xattn or layernorm" > $O/r4j_kchecks.log 2>&1; tail -6 $O/r4j_kchecks.log | cut -c1-300
timeout 600 python tools/gpu_r4_gemm.py --quick 2>&1 | grep -v amdgpu.ids | tee $O/r4j_gemm_probe.log
