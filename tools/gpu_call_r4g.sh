#!/bin/bash
# round 4, call G: sizes the reference accepts (odd latent levels, token counts that are not multiples of 16 / 64), softmax n_valid checks
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "softmax or split or vt_" > $O/r4g_kchecks.log 2>&1; tail -5 $O/r4g_kchecks.log | cut -c1-250
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "sizes" > $O/r4g_sizes.log 2>&1; tail -25 $O/r4g_sizes.log | cut -c1-400
