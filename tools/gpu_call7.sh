#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=line -p no:cacheprovider -k "ring_ or linear or geglu or conv or vt_" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_new.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kp0 -o kp -- python $R/tools/gpu_kprobe.py 5 > $O/kp0.log 2>&1; echo "kp0 rc=$?"
cd $R
db=$(find $O/kp0 -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db --by-grid --match gemm > $O/kp0_table.txt
cat $O/kp0_table.txt | cut -c1-160
find $O/kp0 -size +5M -delete 2>/dev/null
