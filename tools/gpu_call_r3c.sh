#!/bin/bash
# round 3, call C: rocprofv3 kernel trace of a TIMED hipGraph call (no roofline replica in the trace): timeline idle / overlap + per-kernel stats
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
TAG=${1:-r3c}
cd /tmp
for mode in default serial; do
  extra=""; [ $mode = serial ] && extra="--no-overlap"
  rm -rf $O/prof_$mode
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $extra > $O/${TAG}_bench_prof_$mode.json 2> $O/${TAG}_prof_$mode.err; echo "prof $mode rc=$?"
  db=$(find $O/prof_$mode -name "*.db" | head -1)
  if [ -n "$db" ]; then
    python $R/tools/rocpd_summary.py $db > $O/${TAG}_prof_${mode}_kernel_stats.txt
    python $R/tools/rocpd_summary.py $db --by-grid > $O/${TAG}_prof_${mode}_kernel_stats_by_grid.txt
    python $R/tools/rocpd_summary.py $db --timeline > $O/${TAG}_prof_${mode}_timeline.txt
    cat $O/${TAG}_prof_${mode}_timeline.txt
  fi
  rm -rf $O/prof_$mode
done
cd $R
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench.json 2>$O/${TAG}_bench.err; cut -c1-400 $O/${TAG}_bench.json
