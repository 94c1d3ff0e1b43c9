#!/bin/bash
# Round 4, call P: the rocprofv3 legs of gpu_final.sh and the PMC traffic passes again, this time without bench.py's fp16 companion
# leg inside the profiled process (call O's "last call" / PMC window was the fp16 engine's).
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=r04_final
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/${TAG}_build2.log 2>&1 || { echo BUILD FAILED; exit 1; }
cd /tmp
for mode in default serial; do
  extra=""; [ $mode = serial ] && extra="--no-overlap"
  rm -rf $O/prof_$mode
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fp16-leg $extra > $O/${TAG}_bench_prof_$mode.json 2> $O/${TAG}_prof_$mode.err; echo "prof $mode rc=$?"
  db=$(find $O/prof_$mode -name "*.db" | head -1)
  if [ -n "$db" ]; then
    python $R/tools/rocpd_summary.py $db > $O/${TAG}_prof_${mode}_kernel_stats.txt
    python $R/tools/rocpd_summary.py $db --by-grid > $O/${TAG}_prof_${mode}_kernel_stats_by_grid.txt
    python $R/tools/rocpd_summary.py $db --timeline | tee $O/${TAG}_prof_${mode}_timeline.txt
  fi
  rm -rf $O/prof_$mode
done
cd $R
bash tools/gpu_pmc_traffic.sh "round 4 kernels (gemm_conv + gemm_lin + gemm_xattn), bf16 engine only"
