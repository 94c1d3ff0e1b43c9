#!/bin/bash
# round 4, call D: V^T epilogue through LDS (every tile that has a transposed part) -- kernel checks of every vt / QKV / ln-fold case and the
# GEMM probe (fused QKV with V^T vs the same GEMM without)
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "vt or qkv or linear" > $O/r4d_kchecks.log 2>&1; tail -8 $O/r4d_kchecks.log | cut -c1-250
timeout 600 python tools/gpu_r4_gemm.py --quick 2>&1 | grep -v amdgpu.ids | grep -A9 "qkv\|plain 3072" | tee $O/r4d_gemm_probe.log
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r4d_parity.log 2>&1; tail -8 $O/r4d_parity.log | cut -c1-300
