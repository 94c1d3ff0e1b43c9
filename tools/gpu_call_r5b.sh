#!/bin/bash
# Round 5, call b: after the LayerNorm-fold prune and the second batch of tile candidates (w8 forms, 16-wave tiles, 8-wave fused cross-attention
# projection): kernel + engine parity tests, the tuner, bench with the committed table vs the new one, and bench.py's own PMC leg.
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/r5b_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r5b_build.log; exit 1; }
t0=$(date +%s)
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_kernels_hypothesis_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r5b_pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; tail -6 $O/r5b_pytest_gpu.log | cut -c1-220
t0=$(date +%s)
timeout 600 python tools/gpu_tune.py --out $O/r5b_tune_gfx950.json > $O/r5b_tune.log 2>&1; echo "tune rc=$? in $(( $(date +%s) - t0 )) s"; tail -3 $O/r5b_tune.log | cut -c1-200
for tab in committed new committed new; do
  [ $tab = new ] && export IDMVTON_TUNE_TABLE=$O/r5b_tune_gfx950.json || unset IDMVTON_TUNE_TABLE
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-leg --no-pmc > $O/r5b_bench_$tab.json.tmp 2> $O/r5b_bench_$tab.err; echo "bench $tab rc=$?"
  python - <<PY
import json
d = json.load(open("$O/r5b_bench_$tab.json.tmp"))
r = d.get("roofline", {})
print("$tab", round(d["value"], 4), "img/s loop", round(d["loop_ms_per_denoise_step"], 3), "ms/step frac", round(r.get("frac", 0), 4), "loop_mfma_frac", round(r.get("loop_mfma_frac", 0), 4), r.get("step_kernel_ms"))
PY
  cat $O/r5b_bench_$tab.json.tmp >> $O/r5b_bench_$tab.json
done
export IDMVTON_TUNE_TABLE=$O/r5b_tune_gfx950.json
t0=$(date +%s)
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --pmc-out $O/r5b_pmc_traffic.json > $O/r5b_bench_pmc.json 2> $O/r5b_bench_pmc.err; echo "bench+pmc rc=$? in $(( $(date +%s) - t0 )) s"
python - <<PY
import json
d = json.load(open("$O/r5b_bench_pmc.json"))
r = d["roofline"]
print(round(d["value"], 4), {k: d.get(k, {}).get("value") for k in ("fp16", "fp16_fp8")}, "traffic", r.get("traffic"), r.get("traffic_per_kernel"), "|", r.get("traffic_source", "")[:160])
PY
