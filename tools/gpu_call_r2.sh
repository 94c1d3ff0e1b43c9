#!/bin/bash
# round 2, GPU call E: kernel checks (attention final variants, prefetch, colscale), bench with / without prefetch, re-tune, bench
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=line -p no:cacheprovider -k "attn or vt or probe or colscale or prefetch or gpu_available" > $O/r2e_pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -8 $O/r2e_pytest_attn.log
timeout 600 python tools/gpu_r2_probe.py attn > $O/r2e_probe_attn.log 2>&1; echo "attn probe rc=$?"; grep -A7 "tryon_L1\|tryon_L2" $O/r2e_probe_attn.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2e_bench_pf.json 2> $O/r2e_bench_pf.err; echo "bench(prefetch) rc=$?"; cut -c1-1500 $O/r2e_bench_pf.json
IDMVTON_NO_PREFETCH=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2e_bench_nopf.json 2> $O/r2e_bench_nopf.err; echo "bench(no prefetch) rc=$?"; cut -c1-1500 $O/r2e_bench_nopf.json
timeout 1500 python tools/gpu_tune.py > $O/r2e_tune.log 2>&1; echo "tune rc=$?"; tail -4 $O/r2e_tune.log
cp $O/tune_gfx950.json $R/idm-vton_amd/tune_gfx950.json
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2e_bench_tuned.json 2> $O/r2e_bench_tuned.err; echo "bench(tuned) rc=$?"; cut -c1-1500 $O/r2e_bench_tuned.json
