#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=line -p no:cacheprovider -k "r256x256 or groupnorm" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_new.log
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/kp1 -o kp -- python $R/tools/gpu_kprobe.py 3 > $O/kp1.log 2>&1; echo "kp1 rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/kp2 -o kp -- python $R/tools/gpu_kprobe.py 3 > $O/kp2.log 2>&1; echo "kp2 rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --kernel-trace --output-format csv -d $O/kp3 -o kp -- python $R/tools/gpu_kprobe.py 3 > $O/kp3.log 2>&1; echo "kp3 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kp0 -o kp -- python $R/tools/gpu_kprobe.py 5 > $O/kp0.log 2>&1; echo "kp0 rc=$?"
cd $R
python tools/pmc_by_kernel.py $O/kp1 --match _kernel > $O/kp1_table.txt 2>&1
python tools/pmc_by_kernel.py $O/kp2 $O/kp3 --match _kernel > $O/kp23_table.txt 2>&1
db=$(find $O/kp0 -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db --by-grid --match _kernel > $O/kp0_table.txt
cat $O/kp0_table.txt | cut -c1-160; cat $O/kp1_table.txt | cut -c1-240; cat $O/kp23_table.txt | cut -c1-200
find $O/kp0 $O/kp1 $O/kp2 $O/kp3 -size +5M -delete 2>/dev/null
