#!/bin/bash
# Round 5, call f: the 256-register ping-pong attention build with the softmax denominator accumulated by the matrix pipe (LSUM): attention kernel
# checks, the tuner (does the deep build win a signature now?), bench A/B installed table vs new table.
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/r5f_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r5f_build.log; exit 1; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attn" > $O/r5f_pytest_gpu.log 2>&1; echo "pytest attn rc=$?"; tail -3 $O/r5f_pytest_gpu.log | cut -c1-200
timeout 600 python tools/gpu_tune.py --out $O/r5f_tune_gfx950.json > $O/r5f_tune.log 2>&1; echo "tune rc=$?"; grep "^attn" $O/r5f_tune.log | cut -c1-330
show() { python - <<PY
import json
d = json.load(open("$1"))
r = d.get("roofline", {})
print("$2", round(d["value"], 4), "img/s loop", round(d["loop_ms_per_denoise_step"], 3), "ms/step", r.get("step_kernel_ms"), "attn TF", round(r["attn_fwd"]["achieved"], 1))
PY
}
for tab in installed new installed new; do
  [ $tab = new ] && export IDMVTON_TUNE_TABLE=$O/r5f_tune_gfx950.json || unset IDMVTON_TUNE_TABLE
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-leg --no-pmc > $O/r5f_bench_$tab.json.tmp 2> $O/r5f_bench_$tab.err; echo "bench $tab rc=$?"
  show $O/r5f_bench_$tab.json.tmp $tab; cat $O/r5f_bench_$tab.json.tmp >> $O/r5f_bench_$tab.json
done
