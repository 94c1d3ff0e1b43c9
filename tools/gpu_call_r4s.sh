#!/bin/bash
# Round 4, call S: what bounds the hand-scheduled 256x256 loop -- operand delivery or what happens inside the CU?  The same kernel with every
# k-tile re-reading tile 0 (L2-hot) and with no DMA in the loop at all, next to the real one and hipBLASLt.
# (ran at commit 7b6465b: the measurement forms were removed afterwards and live in that commit)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/r4s_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r4s_build.log; exit 1; }
IDMVTON_GEMM_DIAG=1 timeout 900 python tools/gpu_r4_gemm.py 2>&1 | tee $O/r4s_gemm_diag.log | grep -v "r128x\|p128x\|h192" | cut -c1-200
