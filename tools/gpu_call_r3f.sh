#!/bin/bash
# round 3, call F: fast erf + batched VAE encode: checks, parity, bench, timeline with the outside-the-loop breakdown
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "geglu or gelu or elementwise" > $O/r3f_kchecks.log 2>&1; tail -3 $O/r3f_kchecks.log
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_boundary_gpu.py tests/test_clip_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/r3f_parity_small.log 2>&1; tail -5 $O/r3f_parity_small.log
for i in 1 2; do
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>$O/r3f_bench.err | tail -1 > $O/r3f_bench.json
python -c "import sys,json; d=json.load(open('$O/r3f_bench.json')); print(round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['frac'],4), d['roofline']['step_kernel_ms'], round(d['roofline']['launches_per_denoise_step']))" || tail -5 $O/r3f_bench.err
done
cd /tmp
rm -rf $O/prof_default
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_default -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/r3f_bench_prof.json 2> $O/r3f_prof.err
db=$(find $O/prof_default -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db --timeline | tee $O/r3f_prof_default_timeline.txt
python $R/tools/rocpd_summary.py $db --by-grid > $O/r3f_prof_default_kernel_stats_by_grid.txt
rm -rf $O/prof_default
