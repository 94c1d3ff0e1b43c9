#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "attn" 2>&1 | tail -4
timeout 600 python tools/gpu_r2_probe.py attn > $O/r2m_probe_attn.log 2>&1; echo "attn probe rc=$?"; grep -A9 "tryon_L1\|tryon_L2\|garm_L1" $O/r2m_probe_attn.log | head -50
