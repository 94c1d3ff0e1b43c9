#!/bin/bash
# round 4, call N: in-launch 2-way split-K of the 128x256 tile: kernel checks (fp32 reference, repeatability under load, counters back to zero),
# timing against the other tiles on ff2 3072x1280x5120 / the 1280^2 projections
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "splitk" > $O/r4n_kchecks.log 2>&1; tail -5 $O/r4n_kchecks.log | cut -c1-300
timeout 600 python tools/gpu_r4_gemm.py --quick 2>&1 | grep -v amdgpu.ids | grep -A10 "^ff2\|^proj\|^plain" | tee $O/r4n_gemm_probe.log
