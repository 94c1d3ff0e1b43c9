#!/bin/bash
# round 4, call C: split-precision VAE decode: kernel checks, tiny pipeline parity, full-size decode vs the fp32 oracle + timing
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "split or groupnorm or elementwise" > $O/r4c_kchecks.log 2>&1; tail -15 $O/r4c_kchecks.log | cut -c1-250
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "tiny_pipeline_parity" > $O/r4c_parity.log 2>&1; tail -15 $O/r4c_parity.log | cut -c1-300
timeout 900 python tools/gpu_r4_vae.py 2>&1 | grep -v amdgpu.ids | tee $O/r4c_vae.log
