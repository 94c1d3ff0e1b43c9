#!/bin/bash
# Round 4, call W: persistent tiles with cross-tile prefetch in the hand-scheduled Linear loop: kernel checks (incl. the 5-workgroup walk forms),
# then the probe against the compiler tile and hipBLASLt (bit-compare with the ring tile on the multi-round shapes).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/r4w_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r4w_build.log; exit 1; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "walk or h5f or h192 or ln_fold or geglu" --tb=short 2>&1 | tail -12 | cut -c1-250
timeout 900 python tools/gpu_r4_gemm.py 2>&1 | tee $O/r4w_gemm_probe.log | grep -v "r128x\|p128x" | cut -c1-200
