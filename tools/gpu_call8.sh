#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
cd /tmp
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/kp4 -o kp -- python $R/tools/gpu_kprobe.py 3 > $O/kp4.log 2>&1; echo "kp4 rc=$?"
timeout 200 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/kp5 -o kp -- python $R/tools/gpu_kprobe.py 3 > $O/kp5.log 2>&1; echo "kp5 rc=$?"
cd $R
python tools/pmc_by_kernel.py $O/kp4 --match gemm_conv > $O/kp4_table.txt 2>&1; cat $O/kp4_table.txt | cut -c1-250
python tools/pmc_by_kernel.py $O/kp5 --match gemm_conv > $O/kp5_table.txt 2>&1; cat $O/kp5_table.txt | cut -c1-250
tail -3 $O/kp4.log $O/kp5.log
find $O/kp4 $O/kp5 -size +5M -delete 2>/dev/null
