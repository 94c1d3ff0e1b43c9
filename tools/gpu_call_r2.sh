#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/gpu_r2_probe.py gemm > $O/r2i_probe_gemm.log 2>&1; echo "gemm probe rc=$?"; grep -v "_warm \|xwarm_wpf" $O/r2i_probe_gemm.log | tail -70
timeout 600 python -m pytest tests/test_dropin_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
