#!/bin/bash
# Round 4, call R: 256x256 as 4 waves of 128x128 fed by buffer loads + ds_write (tile_hint variant 5, form 2): kernel checks, then the
# GEMM probe against the 8-wave hand-scheduled forms, the compiler tiles and hipBLASLt.
# (ran at commit 5efd548: the form it measures was removed afterwards and lives in that commit)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/r4r_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r4r_build.log; exit 1; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "h5g4" --tb=short 2>&1 | tail -12 | cut -c1-250
timeout 900 python tools/gpu_r4_gemm.py --quick 2>&1 | tee $O/r4r_gemm_probe.log | tail -60 | cut -c1-260
