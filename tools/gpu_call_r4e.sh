#!/bin/bash
# round 4, call E: DressCode / fp8 GPU test, config-4 decode (chunked split-precision attention), parity with the restored image bars,
# bench line with the new fields (old tune table) -> re-tune with the hand-scheduled tile as a candidate -> bench again on the same box
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dresscode_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/r4e_dresscode.log 2>&1; tail -12 $O/r4e_dresscode.log | cut -c1-300
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_dropin_gpu.py tests/test_boundary_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/r4e_parity.log 2>&1; tail -12 $O/r4e_parity.log | cut -c1-300
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/r4e_bench_oldtune.json 2> $O/r4e_bench_oldtune.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4e_bench_oldtune.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "call_parts_ms", "loop_ms_per_denoise_step", "fp16")}, {k: d["roofline"][k] for k in ("frac", "frac_in_loop", "step_kernel_ms")})
PY
timeout 1200 python tools/gpu_tune.py > $O/r4e_tune.log 2>&1; tail -3 $O/r4e_tune.log
cp $O/tune_gfx950.json $R/idm-vton_amd/tune_gfx950.json
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp16-leg > $O/r4e_bench_newtune.json 2> $O/r4e_bench_newtune.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4e_bench_newtune.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "call_parts_ms", "loop_ms_per_denoise_step")}, {k: d["roofline"][k] for k in ("frac", "frac_in_loop", "step_kernel_ms")})
PY
