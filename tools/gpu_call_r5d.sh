#!/bin/bash
# Round 5, call d: 12-wave 256x192 (also for launches with a V^T part: the fused QKV), 16-wave 320x256; V^T parts allowed on the w8 / w16 tiles.
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/r5d_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r5d_build.log; exit 1; }
t0=$(date +%s)
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "w12_256x192 or w16_320x256 or w8_128x128 or w16_256x256 or w16_128x256 or gemm_f8_out" > $O/r5d_pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; tail -4 $O/r5d_pytest_gpu.log | cut -c1-220
timeout 600 python tools/gpu_tune.py --out $O/r5d_tune_gfx950.json > $O/r5d_tune.log 2>&1; echo "tune rc=$?"; tail -2 $O/r5d_tune.log | cut -c1-200
show() { python - <<PY
import json
d = json.load(open("$1"))
r = d.get("roofline", {})
print("$2", round(d["value"], 4), "img/s loop", round(d["loop_ms_per_denoise_step"], 3), "ms/step frac", round(r.get("frac", 0), 4), r.get("step_kernel_ms"))
PY
}
for tab in installed new installed new; do
  [ $tab = new ] && export IDMVTON_TUNE_TABLE=$O/r5d_tune_gfx950.json || unset IDMVTON_TUNE_TABLE
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-leg --no-pmc > $O/r5d_bench_$tab.json.tmp 2> $O/r5d_bench_$tab.err; echo "bench $tab rc=$?"
  show $O/r5d_bench_$tab.json.tmp $tab; cat $O/r5d_bench_$tab.json.tmp >> $O/r5d_bench_$tab.json
done
