#!/bin/bash
# Round 5, call a: the whole GPU suite on the restructured GEMM translation units (new tiles: 8-wave 128x128, 320x256, intra-workgroup split-K; new
# full-size fp16+fp8 parity leg), the tuner over the widened candidate set, bench with the committed table vs the new one on the same box, and the
# library yardstick on the M = 3072 shapes.
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/r5a_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r5a_build.log; exit 1; }
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $O/r5a_pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; tail -25 $O/r5a_pytest_gpu.log | cut -c1-220
cp $O/fullsize_parity.json $O/r5a_fullsize_parity.json 2>/dev/null
t0=$(date +%s)
timeout 600 python tools/gpu_tune.py --out $O/r5a_tune_gfx950.json > $O/r5a_tune.log 2>&1; echo "tune rc=$? in $(( $(date +%s) - t0 )) s"; tail -4 $O/r5a_tune.log | cut -c1-200
for tab in committed new committed new; do
  [ $tab = new ] && export IDMVTON_TUNE_TABLE=$O/r5a_tune_gfx950.json || unset IDMVTON_TUNE_TABLE
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-leg > $O/r5a_bench_$tab.json.tmp 2> $O/r5a_bench_$tab.err; echo "bench $tab rc=$?"
  python - <<PY
import json
d = json.load(open("$O/r5a_bench_$tab.json.tmp"))
r = d.get("roofline", {})
print("$tab", round(d["value"], 4), "img/s loop", round(d["loop_ms_per_denoise_step"], 3), "ms/step frac", round(r.get("frac", 0), 4), r.get("step_kernel_ms"))
PY
  cat $O/r5a_bench_$tab.json.tmp >> $O/r5a_bench_$tab.json
done
unset IDMVTON_TUNE_TABLE
timeout 300 python tools/gpu_library_yardstick.py --gemm-only > $O/r5a_yardstick.log 2>&1; echo "yardstick rc=$?"; grep -E "^gemm|3072|12288" $O/r5a_yardstick.log | head -12 | cut -c1-200
