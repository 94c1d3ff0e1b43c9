#!/bin/bash
# Round-end evidence (tools/gpu_final.sh TAG, e.g. r06_final): build, full GPU test suite, smoke(), the bench line as the driver runs it (with its own PMC traffic leg; the summary is
# also written to profiles-ready JSON), other configurations, rocprofv3 kernel traces of the timed hipGraph loop (two-stream and serial).
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r06_final}
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/${TAG}_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/${TAG}_build.log; exit 1; }
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; tail -16 $O/${TAG}_pytest_gpu.log | cut -c1-200
cp $O/fullsize_parity.json $O/${TAG}_fullsize_parity.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/${TAG}_smoke.log | cut -c1-400
t0=$(date +%s)
timeout 900 python bench.py --steps 10 --warmup 3 --pmc-out $O/${TAG}_pmc_traffic.json > $O/${TAG}_bench_clean.json 2> $O/${TAG}_bench_clean.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"; cut -c1-1200 $O/${TAG}_bench_clean.json
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-pmc --no-fp16-leg > $O/${TAG}_bench_2.json 2>/dev/null; cut -c1-160 $O/${TAG}_bench_2.json
timeout 600 python bench.py --height 1536 --width 1024 --denoise-steps 50 --batch 1 --steps 2 --no-cpu-baseline --no-roofline --no-pmc > $O/${TAG}_bench_cfg4.json 2> $O/${TAG}_bench_cfg4.err; echo "bench cfg4 rc=$?"; cut -c1-160 $O/${TAG}_bench_cfg4.json
timeout 600 python bench.py --batch 4 --steps 2 --no-cpu-baseline --no-roofline --no-pmc > $O/${TAG}_bench_b4.json 2> $O/${TAG}_bench_b4.err; echo "bench B=4 rc=$?"; cut -c1-160 $O/${TAG}_bench_b4.json
cd /tmp
for mode in default serial; do
  extra=""; [ $mode = serial ] && extra="--no-overlap"
  rm -rf $O/prof_$mode
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-pmc --no-fp16-leg $extra > $O/${TAG}_bench_prof_$mode.json 2> $O/${TAG}_prof_$mode.err; echo "prof $mode rc=$?"
  db=$(find $O/prof_$mode -name "*.db" | head -1)
  if [ -n "$db" ]; then
    python $R/tools/rocpd_summary.py $db > $O/${TAG}_prof_${mode}_kernel_stats.txt
    python $R/tools/rocpd_summary.py $db --by-grid > $O/${TAG}_prof_${mode}_kernel_stats_by_grid.txt
    python $R/tools/rocpd_summary.py $db --timeline | tee $O/${TAG}_prof_${mode}_timeline.txt
  fi
  rm -rf $O/prof_$mode
done
