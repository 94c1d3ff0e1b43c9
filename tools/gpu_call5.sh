#!/bin/bash
# tuner (incl. 256x256 tile) -> table in place -> PMC traffic passes (serial eager, 2 denoise steps) -> clean bench line
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 600 python tools/gpu_tune.py > $O/tune.log 2>&1; echo "tune rc=$?"; tail -3 $O/tune.log
[ -s $O/tune_gfx950.json ] && cp $O/tune_gfx950.json idm-vton_amd/tune_gfx950.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-graph --no-overlap --no-cpu-baseline --no-roofline > $O/pmc_$c.json 2> $O/pmc_$c.err; echo "pmc $c rc=$?"
done
cd $R
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json 2> $O/pmc_summary.err
python -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print({k: d.get(k) for k in ('gemm_conv_bytes_per_launch','gemm_conv_launches_counted')}); print({k:(round(v.get('hbm_bytes_per_launch',0)/1e6,2), v.get('FETCH_SIZE',{}).get('dispatches')) for k,v in d['denoise_step'].items()})"
mkdir -p profiles; cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py > $O/bench_clean.json 2> $O/bench_clean.err; echo "bench rc=$?"; cut -c1-2500 $O/bench_clean.json
find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*.csv" -size +8M -delete 2>/dev/null
