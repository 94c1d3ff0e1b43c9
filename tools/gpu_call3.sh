#!/bin/bash
# Measurement call: GPU tests, the standard bench line (clean), rocprofv3 kernel traces (default command and serial mode),
# PMC passes (FETCH_SIZE / WRITE_SIZE separately, kernel-trace only) on a short serial run.
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_clean.json 2> $O/bench_clean.err; echo "bench rc=$?"; cat $O/bench_clean.json | cut -c1-1500
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_default -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_prof_default.json 2> $O/prof_default.err; echo "prof default rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap > $O/bench_prof_serial.json 2> $O/prof_serial.err; echo "prof serial rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-graph --no-overlap --no-cpu-baseline --no-roofline > $O/pmc_$c.json 2> $O/pmc_$c.err; echo "pmc $c rc=$?"
done
cd $R
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json 2> $O/pmc_summary.err; head -c 600 $O/pmc_traffic.json
for d in prof_default prof_serial; do
  db=$(find $O/$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/${d}_kernel_stats.txt && python tools/rocpd_summary.py $db --by-grid > $O/${d}_kernel_stats_by_grid.txt
  find $O/$d -name "*stats*.csv" | head -3
done
# keep the merge small: drop raw traces, keep summaries
find $O/prof_default $O/prof_serial -name "*.db" -size +20M -delete 2>/dev/null
find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*.csv" -size +20M -delete 2>/dev/null
du -sh $O | tail -1
