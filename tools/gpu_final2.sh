#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 170 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-300
timeout 40 python tools/gpu_quick_gemm.py 2>&1 | grep -v amdgpu.ids | tee $O/quick_gemm.log
timeout 100 python bench.py > $O/bench_clean.json 2> $O/bench_clean.err; echo "bench rc=$?"; cut -c1-900 $O/bench_clean.json
