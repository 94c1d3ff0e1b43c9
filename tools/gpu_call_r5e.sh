#!/bin/bash
# Round 5, call e: side-stream weight prefetch beside TryonNet's GEMM chain (IDMVTON_PREFETCH="ahead[,blocks[,min_kib]]"): results identical
# (the execution-form parity tests with it on), then bench A/B on one box: off / ahead 1 / 2 / 3 / more or fewer workgroups.
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/r5e_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r5e_build.log; exit 1; }
IDMVTON_PREFETCH=2 timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "tiny_pipeline or two_stream or bit_reproducible or timestep_batching" > $O/r5e_pytest_gpu.log 2>&1; echo "pytest (prefetch on) rc=$?"; tail -3 $O/r5e_pytest_gpu.log | cut -c1-200
show() { python - <<PY
import json
d = json.load(open("$1"))
r = d.get("roofline", {})
print("$2", round(d["value"], 4), "img/s loop", round(d["loop_ms_per_denoise_step"], 3), "ms/step", r.get("step_kernel_ms"))
PY
}
for pf in off 2 off 1 3 2,128 2,24 4 off 2; do
  [ $pf = off ] && unset IDMVTON_PREFETCH || export IDMVTON_PREFETCH=$pf
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-leg --no-pmc > $O/r5e_bench_tmp.json 2> $O/r5e_bench_$pf.err; echo "bench prefetch=$pf rc=$?"
  show $O/r5e_bench_tmp.json "prefetch=$pf"; cat $O/r5e_bench_tmp.json >> $O/r5e_bench_pf_$pf.json
done
