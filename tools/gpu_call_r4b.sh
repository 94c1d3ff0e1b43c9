#!/bin/bash
# round 4, call B: the 8-wave hand-scheduled loop's placement forms (DMA split, barrier position, static priority) + its V^T part
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "h5f" > $O/r4b_kchecks.log 2>&1; tail -5 $O/r4b_kchecks.log | cut -c1-200
timeout 600 python tools/gpu_r4_gemm.py --quick 2>&1 | grep -v amdgpu.ids | tee $O/r4b_gemm_probe.log
