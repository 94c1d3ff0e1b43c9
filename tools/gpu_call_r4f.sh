#!/bin/bash
# round 4, call F: V^T epilogue forms per tile under three timing regimes; odd-size parity tests; ups-to-odd-grid kernel checks
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/gpu_r4_vt.py 2>&1 | grep -v amdgpu.ids | tee $O/r4f_vt_probe.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "ups or conv" > $O/r4f_kchecks.log 2>&1; tail -5 $O/r4f_kchecks.log | cut -c1-250
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "sizes or tiny_pipeline_parity" > $O/r4f_sizes.log 2>&1; tail -25 $O/r4f_sizes.log | cut -c1-400
