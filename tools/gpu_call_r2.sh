#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
rm -f $O/tune_gfx950.json
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2p_bench_oldtune.json 2> $O/r2p_bench_oldtune.err; echo "old tune rc=$? $(python -c "
import json; d=json.loads(open('$O/r2p_bench_oldtune.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
timeout 1500 python tools/gpu_tune.py > $O/r2p_tune.log 2>&1; echo "tune rc=$?"; tail -3 $O/r2p_tune.log
[ -f $O/tune_gfx950.json ] && cp $O/tune_gfx950.json $R/idm-vton_amd/tune_gfx950.json
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2p_bench_tuned.json 2> $O/r2p_bench_tuned.err; echo "tuned rc=$? $(python -c "
import json; d=json.loads(open('$O/r2p_bench_tuned.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_kernel_ms'], d['roofline']['launches_per_denoise_step'])")"
