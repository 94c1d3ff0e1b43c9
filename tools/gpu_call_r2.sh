#!/bin/bash
# round 2, GPU call H: the whole -m gpu suite (timed), smoke, clean bench, rocprofv3 kernel traces (default + serial)
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
( time timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 ) > $O/r2h_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $O/r2h_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r2h_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/r2h_smoke.log | cut -c1-400
timeout 900 python bench.py > $O/r2h_bench_clean.json 2> $O/r2h_bench_clean.err; echo "bench rc=$?"; cut -c1-2500 $O/r2h_bench_clean.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_default -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/r2h_bench_prof_default.json 2> $O/prof_default.err; echo "prof default rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap > $O/r2h_bench_prof_serial.json 2> $O/prof_serial.err; echo "prof serial rc=$?"
cd $R
for d in prof_default prof_serial; do
  db=$(find $O/$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/r2h_${d}_kernel_stats.txt && python tools/rocpd_summary.py $db --by-grid > $O/r2h_${d}_kernel_stats_by_grid.txt
done
find $O/prof_default $O/prof_serial -name "*.db" -size +20M -delete 2>/dev/null
head -14 $O/r2h_prof_serial_kernel_stats.txt | cut -c1-170
