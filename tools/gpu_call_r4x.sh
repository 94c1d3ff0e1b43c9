#!/bin/bash
# Round 4, call X: same-box A/B through bench.py of the library before (libidmvton_hip_prev.so) and after the persistent-tile walk in gemm_lin.hip
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/r4x_build.log 2>&1 || { echo BUILD FAILED; exit 1; }
for tag in cur prev cur prev cur prev; do
  lib=$PWD/idm-vton_amd/libidmvton_hip.so; [ $tag != cur ] && lib=$PWD/idm-vton_amd/libidmvton_hip_$tag.so
  IDMVTON_HIP_LIB=$lib timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp16-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],4), round(d['ms_per_step'],1), round(d['loop_ms_per_denoise_step'],3))" | tee -a $O/r4x_ab_persistent.log
done
