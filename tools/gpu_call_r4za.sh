#!/bin/bash
# Round 4, last call: PMC counters of the hand-scheduled persistent Linear kernel next to the compiler tiles and hipBLASLt on 8192^3 and the
# 3072x10240x1280 shape (two rocprofv3 --pmc passes, counters only).
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/r4za_build.log 2>&1 || { echo BUILD FAILED; exit 1; }
cd /tmp
rm -rf $O/pmc_r4lin_a $O/pmc_r4lin_b
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_r4lin_a -o run -- python $R/tools/gpu_gemm_pmc.py 2 > $O/r4za_a.log 2>&1; echo "pass a rc=$?"
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU -d $O/pmc_r4lin_b -o run -- python $R/tools/gpu_gemm_pmc.py 2 > $O/r4za_b.log 2>&1; echo "pass b rc=$?"
cd $R
python tools/pmc_db_by_kernel.py $O/pmc_r4lin_a $O/pmc_r4lin_b | tee $O/r4za_pmc_gemm_lin_vs_hipblaslt.txt | cut -c1-260
find $O/pmc_r4lin_a $O/pmc_r4lin_b -name "*.db" -size +20M -delete 2>/dev/null
