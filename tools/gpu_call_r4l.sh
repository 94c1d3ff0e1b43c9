#!/bin/bash
# round 4, call L: tuning regime A/B on one box: table tuned with weights touched back into cache (call K's) vs weights from HBM (the loop's real state)
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cp $R/idm-vton_amd/tune_gfx950.json $O/tune_warm_weights.json
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-leg > $O/r4l_bench_warmtune.json 2> $O/r4l_bench.err; python -c "
import json; d = json.loads(open('gpurun_out/r4l_bench_warmtune.json').read().strip().splitlines()[-1]); print('warm-weights table', d['value'], d['loop_ms_per_denoise_step'], d['roofline']['step_kernel_ms'])"
timeout 1500 python tools/gpu_tune.py --out $O/tune_cold_weights.json > $O/r4l_tune.log 2>&1; tail -2 $O/r4l_tune.log
cp $O/tune_cold_weights.json $R/idm-vton_amd/tune_gfx950.json
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-leg > $O/r4l_bench_coldtune.json 2>> $O/r4l_bench.err; python -c "
import json; d = json.loads(open('gpurun_out/r4l_bench_coldtune.json').read().strip().splitlines()[-1]); print('cold-weights table', d['value'], d['loop_ms_per_denoise_step'], d['roofline']['step_kernel_ms'])"
cp $O/tune_warm_weights.json $R/idm-vton_amd/tune_gfx950.json
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-leg > $O/r4l_bench_warmtune2.json 2>> $O/r4l_bench.err; python -c "
import json; d = json.loads(open('gpurun_out/r4l_bench_warmtune2.json').read().strip().splitlines()[-1]); print('warm-weights table again', d['value'], d['loop_ms_per_denoise_step'], d['roofline']['step_kernel_ms'])"
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "ln_fold" > $O/r4l_kchecks.log 2>&1; tail -3 $O/r4l_kchecks.log | cut -c1-300
