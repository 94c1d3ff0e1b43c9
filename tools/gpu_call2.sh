#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
timeout 300 python tools/gpu_debug.py ac 2>&1 | grep -v amdgpu.ids | tee $O/debug3.log | cut -c1-400
timeout 600 python tools/gpu_tune.py > $O/tune.log 2>&1; echo "tune rc=$?"; tail -3 $O/tune.log
timeout 900 python tools/gpu_ab.py --modes serial_graph,overlap_graph,overlap_eager > $O/ab.log 2>&1; echo "ab rc=$?"; grep '^{' $O/ab.log
