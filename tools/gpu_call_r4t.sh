#!/bin/bash
# Round 4, call T: the 8-wave hand-scheduled loop with its operands staged through registers (buffer_load_b128 -> ds_write_b128) instead of
# LDS-DMA (variant 5 form 3): kernel checks, then the probe next to the DMA forms and hipBLASLt.
# (ran at commit 7b6465b: the measurement forms were removed afterwards and live in that commit)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/r4t_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r4t_build.log; exit 1; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "h5v or h192v" --tb=short 2>&1 | tail -12 | cut -c1-250
timeout 900 python tools/gpu_r4_gemm.py 2>&1 | tee $O/r4t_gemm_probe.log | grep -v "r128x\|p128x\|h256f0" | cut -c1-200
