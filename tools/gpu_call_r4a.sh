#!/bin/bash
# round 4, call A: the hand-scheduled 256x256 Linear loop (gemm_lin.hip, variants 4 / 5): kernel checks, A/B against the compiler-scheduled
# tiles and hipBLASLt on the real shapes, and this box's baseline bench line
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "h4f or h5f" > $O/r4a_kchecks.log 2>&1; tail -5 $O/r4a_kchecks.log | cut -c1-200
timeout 600 python tools/gpu_r4_gemm.py --quick 2>&1 | grep -v amdgpu.ids | tee $O/r4a_gemm_probe.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r4a_bench.json 2> $O/r4a_bench.err; tail -c 1500 $O/r4a_bench.json
