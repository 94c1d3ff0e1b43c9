#!/bin/bash
# round 2, GPU call F: full-size parity against the oracle (config 2 / config 4, bf16 + fp16), tiny / mid pipeline parity
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
free -g | head -2; nproc
timeout 1500 python -m pytest tests/test_fullsize_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider --durations=12 > $O/r2f_fullsize.log 2>&1; echo "fullsize rc=$?"; tail -40 $O/r2f_fullsize.log
cat $O/fullsize_parity.json
