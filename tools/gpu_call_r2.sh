#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "attn_small or quickgelu" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_clip_gpu.py tests/test_dropin_gpu.py tests/test_boundary_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40
