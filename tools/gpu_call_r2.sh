#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/prof_tl_default -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_tl_default.json 2> $O/prof_tl_default.err; echo "prof default rc=$?"
timeout 600 rocprofv3 --kernel-trace -d $O/prof_tl_serial -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-overlap > $O/bench_tl_serial.json 2> $O/prof_tl_serial.err; echo "prof serial rc=$?"
cd $R
for d in prof_tl_default prof_tl_serial; do
  db=$(find $O/$d -name "*.db" | head -1)
  python tools/rocpd_summary.py $db --timeline | tee $O/${d}_timeline.txt
done
find $O/prof_tl_default $O/prof_tl_serial -name "*.db" -size +20M -delete 2>/dev/null
